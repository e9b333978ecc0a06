#!/usr/bin/env python3
"""Headline benchmark: closed-loop MPC scenario-steps per second.

One "step" = one fused closed-loop step (plan -> linearise/condense -> QP solve -> RK4 plant -> lap count,
plus the EKF in the output-feedback workload) for every scenario of the batch.  Default workload (c2): LMPC2,
B = 4096 perturbed skidpad initial states per GPU, horizon 50, dt 0.01, vx 8, numpy default_rng(rank) draws, state
resident in HBM.  With --gpus G every rank runs its own scenarios (weak scaling, no collective in the loop; one RCCL
all-reduce of the statistics vector at the end).  Other workloads (--workload): c3 LPV N=80 with per-scenario speeds,
c4 output-feedback OFL + EKF with measurement noise, c5 nonlinear SQP with a tyre/mass sweep.

  python bench.py --gpus N --steps K --warmup W        (N>1: under torch.distributed.run, one rank per GPU)

The timed region starts at closed-loop step 50 of every scenario (c2; 10/20/5 for c3/c4/c5), after the constrained
transient from the perturbed initial states: with --warmup W below that, the remaining steps run during set-up
(reported as config.setup_steps).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VX = 8.0
PEAK_FP64_TFLOPS = 78.6      # MI355X FP64 vector = matrix peak (datasheet); the guide's tables carry no f64 row
PEAK_HBM_GBS = 8000.0


def initial_states(B: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    py = rng.uniform(-0.3, 0.3, B); psi = rng.uniform(-0.1, 0.1, B)
    vy = rng.uniform(-0.2, 0.2, B); r = rng.uniform(-0.2, 0.2, B); dl = rng.uniform(-0.05, 0.05, B)
    return np.stack([np.full(B, 15.0), py, np.cos(psi), np.sin(psi), np.full(B, VX), vy, r, dl], axis=1)


def ipm_flops(N: int) -> float:
    """One interior-point iteration: H*U 2N^2, Newton matrix 3N^2, LDL' N^3/3, two solves 4N^2, vector updates 100N."""
    return N ** 3 / 3 + 9 * N * N + 100 * N


def algorithmic_flops(workload: str, N: int, mean_iters: float, mean_sqp: float = 0.0) -> float:
    """FP64 flops one scenario-step needs with the method that is run (symmetry and zero weights exploited).
    c2/c4 (speed constant, problem condensed once per launch and shared): planner 60(N+1); unconstrained minimiser as
    an affine map of the step's data, 3 reference components + 5 state terms: 2(3N^2 + 5N); bound check 10N; RK4 400;
    c4 adds the EKF (two 10x10x10 products, 7x7 solve with 10 right-hand sides, gain products) ~ 8000.
    c3 (condensed per scenario and step): stage model 130N; prediction columns and gradient 16N^2; Hessian, 3 weighted
    outputs, N^3 + 6N^2; LDL' N^3/3 + 2N^2.
    c5, per SQP iteration: RK4 roll-out 160N; RK4 derivative 700N; prediction columns 52N^2/2; Hessian, 5 weighted
    outputs, 5N^3/3; gradient and H*U 4N^2; LDL' N^3/3 + 2N^2.
    Every interior-point iteration adds ipm_flops(N)."""
    plan = 60 * (N + 1) + 400
    if workload in ("c2", "c4"):
        fixed = plan + 2 * (3 * N * N + 5 * N) + 10 * N + (8000 if workload == "c4" else 0)
    elif workload == "c3":
        fixed = plan + 130 * N + 16 * N * N + N ** 3 + 6 * N * N + N ** 3 / 3 + 2 * N * N
    else:
        fixed = plan + mean_sqp * (860 * N + 26 * N * N + 5 * N ** 3 / 3 + 4 * N * N + N ** 3 / 3 + 2 * N * N)
    return fixed + ipm_flops(N) * mean_iters


def algorithmic_bytes(workload: str, N: int) -> float:
    """HBM bytes one scenario-step must move when every step is its own launch: plant state 8 + lap/progress/speed 3
    + u0 1 doubles, read and written once; c3 adds the linearisation memory 6(N+1)+N, c4 the 10-state plant, estimate,
    covariance and 7 noise samples, c5 the input iterate N.  (The condensed Hessian scratch is method traffic.)"""
    extra = {"c2": 0, "c3": 6 * (N + 1) + N, "c4": 2 + 10 + 100 + 7, "c5": N}[workload]
    return (8 + 3 + 1 + extra) * 8 * 2


def cpu_baseline(workload: str, nscen: int, nsteps: int) -> dict | None:
    """The CPU restatement timed on this host, OpenMP over scenarios, on a bounded sample of the workload."""
    if workload == "c4":
        return None          # the checker composes the output-feedback loop in Python; it has no timed batch loop
    from oracle import oracle as O
    O.build()
    orc = O.Oracle("f64")
    kind, N, Tf = WORKLOADS[workload]["kind"], WORKLOADS[workload]["N"], WORKLOADS[workload]["Tf"]
    cfg = O.make_cfg(kind, N, Tf, sqp_tol=1e-30, sqp_max_iter=3) if kind == "NL" else O.make_cfg(kind, N, Tf)
    st = initial_states(nscen, 0)
    laps = np.zeros(nscen, np.int32); pp = np.zeros(nscen)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    t0 = time.perf_counter()
    orc.skidpad_run(cfg, st, laps, pp, VX, nsteps, record=False)
    dtm = time.perf_counter() - t0
    out = {"value": nscen * nsteps / dtm, "unit": "scenario-steps/s", "cores": threads, "kind": "port",
           "sample": f"{nscen} scenarios x {nsteps} closed-loop steps ({kind} controller, N={N}, vx={VX}; transient "
                     f"included), oracle/mpc_oracle.c double build, OpenMP over scenarios, {dtm:.1f} s"}
    if workload == "c2":
        # one scenario on one thread, a call per step: the latency a single controller sees on this host (the
        # reference reports the median of its per-solve run times, plot_generator.py:1160-1191)
        x = np.array([15.0, 0.1, 1.0, 0.0, VX, 0.0, 0.0, 0.0]); lap, prev = 0, 0.0          # skidpad_simulator.py:50-60
        lat = []
        for _ in range(1000):
            t0 = time.perf_counter()
            _, x, lap, prev, _, _ = orc.skidpad_step(cfg, x, lap, prev, VX)
            lat.append(time.perf_counter() - t0)
        out["single_scenario"] = {"p50_step_latency_ms": 1e3 * float(np.median(lat)), "p99_step_latency_ms":
                                  1e3 * float(np.percentile(lat, 99)), "cores": 1,
                                  "sample": "default skidpad start, 1000 closed-loop steps, one call per step"}
    return out


def measured_traffic(args):
    """HBM bytes per launch from the PMC passes kept under profiles/ (rocprofv3 cannot run inside this process); only
    for the configuration those passes were taken on, else None."""
    path = os.path.join(ROOT, "profiles", "r01e_c2_fused_hbm_traffic.json")
    if args.workload != "c2" or args.per_step_launch or args.steps != 1000 or args.batch not in (None, 4096) or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)["hbm_bytes_per_launch"]


WORKLOADS = {
    "c2": {"kind": "L", "N": 50, "Tf": 0.5, "B": 4096, "kernel": "k_mpc<50, fixed-model>",
           "text": "LMPC2 batched: 4096 perturbed skidpad initial states per GPU, fixed A,B (vx=8), N=50, dt=0.01, closed loop"},
    "c3": {"kind": "LPV", "N": 80, "Tf": 0.8, "B": 8192, "kernel": "k_mpc_any<2>",
           "text": "LPVMPC batched: 8192 scenarios per GPU, per-scenario vx~U(6,12), N=80, dt=0.01, lin_mode reference, closed loop"},
    "c4": {"kind": "OFL", "N": 50, "Tf": 0.5, "B": 8192, "kernel": "k_mpc<50, fixed-model> (MODE_OFL)",
           "text": "OFLMPC2 + EKF batched: 8192 scenarios per GPU, step reference, side force d_f~U(-12,-8), noisy "
                   "measurements, N=50, dt=0.01, output-feedback closed loop"},
    "c5": {"kind": "NL", "N": 50, "Tf": 0.5, "B": 131072, "kernel": "k_mpc_nl<50>",
           "text": "NLMPC batched: 131072 scenarios per GPU, tyre/mass sweep, N=50, dt=0.01, 3 SQP iterations per step, closed loop"},
}


def build_workload(name: str, B: int, rank: int, dev, steps_total: int):
    """Returns (run(n), controller, status tensor, iters setter)."""
    import torch
    from mpc_amd.batched import SkidpadBatch, StepOFLBatch
    w = WORKLOADS[name]
    if name == "c2":
        sim = SkidpadBatch(w["N"], w["Tf"], initial_states(B, rank), model="L", device=dev)
        return sim, sim.run
    if name == "c3":
        st = initial_states(B, 1000 + rank)
        st[:, 4] = np.random.default_rng(1 + 7919 * rank).uniform(6.0, 12.0, B)
        sim = SkidpadBatch(w["N"], w["Tf"], st, model="LPV", device=dev)
        return sim, sim.run
    if name == "c4":
        rng = np.random.default_rng(2 + 7919 * rank)
        true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))
        true0[:, 9] = rng.uniform(-12.0, -8.0, B)
        est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
        est0[:, 1] += rng.normal(0, 0.05, B); est0[:, 5] += rng.normal(0, 0.02, B)
        sim = StepOFLBatch(w["N"], w["Tf"], true0, est0, device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(3 + rank)
        sigma = torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01], dtype=torch.float64, device=dev)
        noise = torch.randn((steps_total, B, 7), dtype=torch.float64, device=dev, generator=gen) * sigma
        pos = [0]

        def run(n):
            sim.run(n, meas_noise=noise[pos[0]: pos[0] + n]); pos[0] += n
        return sim, run
    if name == "c5":
        rng = np.random.default_rng(4 + 7919 * rank)
        sim = SkidpadBatch(w["N"], w["Tf"], initial_states(B, 2000 + rank), model="NL", device=dev,
                           sqp_tol=1e-30, sqp_max_iter=3)          # fixed 3 iterations per step
        mp = np.tile(sim.ocp.model_params(), (B, 1))
        mp[:, 0] = rng.uniform(170, 230, B); mp[:, 1] = rng.uniform(280, 310, B)
        mp[:, 4] *= rng.uniform(0.6, 1.2, B); mp[:, 5] *= rng.uniform(0.6, 1.2, B)
        sim.ocp.set_model_params(mp)
        return sim, sim.run
    raise ValueError(name)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="scenarios per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-launch", action="store_true", help="c2 only: one kernel launch per closed-loop step")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = {"c2": 1000, "c3": 100, "c4": 300, "c5": 50}[args.workload]
    transient = {"c2": 50, "c3": 10, "c4": 20, "c5": 5}[args.workload]
    if args.warmup is None:
        args.warmup = transient
    # The workloads time closed-loop steps from step `transient` of every scenario on (the heavily constrained start from
    # the perturbed initial states is warm-up by definition).  A run asked for fewer warm-up steps advances the scenarios
    # through the missing ones during set-up, so the timed steps are the same steps of the same trajectories.
    settle = max(0, transient - args.warmup)
    if args.per_step_launch and args.workload != "c2":
        ap.error("--per-step-launch is implemented for the c2 workload")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    B = args.batch or wl["B"]
    N = wl["N"]
    sim, run = build_workload(args.workload, B, rank, dev, args.steps + args.warmup + settle)
    if settle:
        run(settle)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The first `warmup` steps from the perturbed initial states are the heavily constrained transient.
    barrier()
    tw0 = time.perf_counter()
    if args.per_step_launch:
        for _ in range(args.warmup):
            sim.step()
    elif args.warmup:
        run(args.warmup)
    barrier()
    t_warm = time.perf_counter() - tw0

    # Timed region: exactly `steps` closed-loop steps of every scenario.  Default: ONE launch carries each
    # scenario through all steps (scenarios are independent); --per-step-launch issues one launch per step.
    nlaunch = args.steps if args.per_step_launch else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlaunch)]
    iters_hist = torch.zeros((nlaunch, B), dtype=torch.int32, device=dev)
    barrier()
    t0 = time.perf_counter()
    if args.per_step_launch:
        for i in range(args.steps):
            sim.iters = iters_hist[i]
            ev[i][0].record()
            sim.step()
            ev[i][1].record()
    else:
        sim.iters = iters_hist[0]
        ev[0][0].record()
        run(args.steps)
        ev[0][1].record()
    barrier()
    elapsed = time.perf_counter() - t0

    st16 = sim.ocp.stats()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_constr = float((iters_hist > 0).sum().item()) if args.per_step_launch else float(st16["n_constrained"])
    stats = torch.tensor([float(iters_hist.sum().item()), n_constr, float((sim.status != 0).sum().item()),
                          float(st16["sum_sqp_iterations"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)       # RCCL: the only collective of the path
    elapsed = float(t.item())
    kernel_ms = np.array([a.elapsed_time(b) for a, b in ev])
    steps_per_launch = args.steps / nlaunch

    if rank == 0:
        total = B * world * args.steps
        mean_iters = float(stats[0].item()) / total
        mean_sqp = float(stats[3].item()) / total
        flops = algorithmic_flops(args.workload, N, mean_iters, mean_sqp)
        launch_s = float(kernel_ms.mean()) * 1e-3
        ach_tflops = flops * B * steps_per_launch / launch_s / 1e12
        out = {
            "metric": "MPC QP solves/sec (node, batched skidpad scenarios) + p50 step latency",
            "value": total / elapsed,
            "unit": "scenario-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "p50_step_latency_ms": float(np.median(kernel_ms)) / steps_per_launch,
            "p99_step_latency_ms": float(np.percentile(kernel_ms, 99)) / steps_per_launch,
            "launch_mode": "per-step" if args.per_step_launch else f"fused ({args.steps} steps per launch)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["text"], "name": args.workload, "scenarios_per_gpu": B, "horizon": N,
                       "parallelism": f"scenario-sharded x{world}",
                       "first_timed_closed_loop_step": settle + args.warmup, "setup_steps": settle},
            "solver": {"mean_ipm_iterations_per_scenario_step": mean_iters,
                       "constrained_qp_per_scenario_step": float(stats[1].item()) / total,
                       "mean_sqp_iterations_per_scenario_step": mean_sqp if args.workload == "c5" else None,
                       "status_nonzero_last_launch": int(stats[2].item()),
                       "status_histogram_rank0": {str(k): int(v) for k, v in enumerate(torch.bincount(sim.status.long()).tolist()) if v},
                       "transient_scenario_steps_per_s": (B * world * args.warmup / t_warm) if args.warmup else None},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / PEAK_FP64_TFLOPS, "traffic": measured_traffic(args),
                         "kernel": wl["kernel"], "avg_launch_ms": launch_s * 1e3,
                         "scenario_steps_per_launch": B * steps_per_launch,
                         "algorithmic_flops_per_scenario_step": flops,
                         "note": "FP64 vector/matrix peak; the kernels are latency- and issue-bound, not FP64-throughput-bound",
                         "hbm": {"achieved_GBs": algorithmic_bytes(args.workload, N) * B * steps_per_launch / launch_s / 1e9,
                                 "peak_GBs": PEAK_HBM_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = {"c2": (1024, 300), "c3": (256, 40), "c4": (0, 0), "c5": (512, 20)}[args.workload]
            out["cpu_baseline"] = cpu_baseline(args.workload, *sample)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
