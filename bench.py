#!/usr/bin/env python3
"""Headline benchmark: closed-loop MPC scenario-steps per second.

One "step" = one fused closed-loop step (plan -> linearise/condense -> QP solve -> RK4 plant -> lap count,
plus the EKF in the output-feedback workload) for every scenario of the batch.  Default workload (c2): LMPC2,
B = 4096 perturbed skidpad initial states per GPU, horizon 50, dt 0.01, vx 8, numpy default_rng(rank) draws, state
resident in HBM.  Other workloads (--workload): c3 LPV N=80 with per-scenario speeds, c4 output-feedback OFL + EKF with
measurement noise, c5 nonlinear SQP with a tyre/mass sweep.

  python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 every rank runs its own scenarios (weak scaling, no collective in the loop; one all-reduce of the
statistics vectors, sums and maxima, at the end).  Started without a launcher (WORLD_SIZE unset), the script starts
the N ranks itself as fresh child processes under torch.distributed.run, before anything touches a GPU; started under
torch.distributed.run it is one of the ranks.  --dry-run exercises that multi-rank path on the CPU (gloo, fabricated
per-rank statistics, no kernel).

The timed region starts at closed-loop step 50 of every scenario (c2; 10/20/5 for c3/c4/c5), after the constrained
transient from the perturbed initial states: with --warmup W below that, the remaining steps run during set-up
(reported as config.setup_steps).  After the timed region the script also times single-step launches (the p50 / p99
step latency of the metric) and, for c2, one 1000-step launch far from the transient (solver.steady_state_*).
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


def algorithmic_flops(workload: str, N: int, mean_iters: float, mean_sqp: float = 0.0, ncomp: int = 3, extra: float = 0.0,
                      aset_qps: float = 0.0) -> float:
    """FP64 flops one scenario-step needs with the method that is run (symmetry and zero weights exploited).
    c2/c4 (speed constant, problem condensed once per launch and shared): planner 60(N+1); unconstrained minimiser as
    an affine map of the step's data, `ncomp` weighted reference components (2 with the yaml's weights: the x weight is
    zero) + 5 state terms: 2(ncomp N^2 + 5N); bound check 10N; RK4 400; c4 adds the EKF (two 10x10x10 products, 7x7
    solve with 10 right-hand sides, gain products) ~ 8000.
    c3 (stage-wise Riccati interior point, nothing condensed), per stage of the horizon: stage model 130; free response,
    gradient scale and warm start ~60, plus ~400 on the steps that try the unconstrained minimiser first (taken as
    300 on average); per interior-point iteration ~880 (step and roll-out 75, Riccati matrix and vector recursion with
    the barrier terms 400, affine forward sweep 120, corrector backward sweep 150, corrector forward sweep 130).
    c5, per SQP iteration: RK4 roll-out 160N; RK4 derivative 700N; prediction columns 52N^2/2; Hessian, 5 weighted
    outputs, 5N^3/3; gradient and H*U 4N^2; LDL' N^3/3 + 2N^2 for the unconstrained minimiser -- not counted for the
    `aset_qps` QPs per scenario-step that the active-set solve finishes (it is tried before that factorisation whenever
    the previous QP was constrained; the few it finishes after a cold start are left out as well, which undercounts).
    Every interior-point iteration of c2/c4/c5 adds ipm_flops(N); the active-set solves add the count the kernel keeps
    for them (`extra`, per scenario-step: LDL' of the m x m system m^3/3, two solves 2m^2, the update of the inputs 2mN,
    scans ~20N per iteration; 2m^2 + 2mN per refinement pass)."""
    plan = 60 * (N + 1) + 400
    if workload in ("c2", "c4"):
        fixed = plan + 2 * (ncomp * N * N + 5 * N) + 10 * N + (8000 if workload == "c4" else 0)
    elif workload == "c3":
        return plan + N * (130 + 300 + 880 * mean_iters)
    else:
        fixed = plan + mean_sqp * (860 * N + 26 * N * N + 5 * N ** 3 / 3 + 4 * N * N) + max(0.0, mean_sqp - aset_qps) * (N ** 3 / 3 + 2 * N * N)
    return fixed + ipm_flops(N) * mean_iters + extra


def weighted_reference_components(ocp) -> int:
    """How many of the three reference components (x, y, sin psi) carry weight in the stage or terminal cost: the
    cached affine map of the fixed-model kernel has one N x N block per such component."""
    W, We = np.asarray(ocp.W), np.asarray(ocp.W_e)[:6, :6]
    return int(sum(bool(W[c, c] != 0.0 or np.any(We[c, :] != 0.0) or np.any(We[:, c] != 0.0)) for c in range(3)))


def algorithmic_bytes(workload: str, N: int) -> float:
    """HBM bytes one scenario-step must move when every step is its own launch: plant state 8 + lap/progress/speed 3
    + u0 1 doubles, read and written once; c3 adds the linearisation memory 6(N+1)+N, c4 the 10-state plant, estimate,
    covariance and 7 noise samples, c5 the input iterate N.  (The condensed Hessian scratch is method traffic.)"""
    extra = {"c2": 0, "c3": 6 * (N + 1) + N, "c4": 2 + 10 + 100 + 7, "c5": N}[workload]
    return (8 + 3 + 1 + extra) * 8 * 2


def method_bytes(workload: str, N: int, mean_iters: float) -> float | None:
    """Workspace bytes the lane-per-scenario kernel streams per scenario-step (c3; the other kernels keep their problem in
    registers and LDS): per stage and interior-point iteration its five sweeps read or write 198 doubles (step 44,
    Riccati 33, affine forward 36, corrector backward 41, corrector forward 44; the solver build without the longitudinal
    position, which the workload's weights select); set-up of a step (stage model and references written, free
    response, gradient scale, warm start or the unconstrained attempt) about 70."""
    if workload != "c3":
        return None
    return 8.0 * N * (70 + 198 * mean_iters)


def host_threads() -> int:
    """Threads the CPU baseline runs on: the cores this process may use, capped by OMP_NUM_THREADS."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(avail, cap) if cap > 0 else avail)


def cpu_baseline(workload: str, nscen: int, nsteps: int) -> dict | None:
    """The CPU restatement timed on this host, OpenMP over scenarios, on a bounded sample of the workload.  As a tuned
    CPU controller would, it takes the unconstrained minimiser (one Riccati pass) where that satisfies the bounds and
    runs its interior point only on the other steps."""
    from oracle import oracle as O
    O.build()
    orc = O.Oracle("f64")
    orc.set_unconstrained_first(True)
    kind, N, Tf = WORKLOADS[workload]["kind"], WORKLOADS[workload]["N"], WORKLOADS[workload]["Tf"]
    cfg = O.make_cfg(kind, N, Tf, sqp_tol=1e-30, sqp_max_iter=3) if kind == "NL" else O.make_cfg(kind, N, Tf)
    threads = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(threads)
    if workload == "c4":
        rng = np.random.default_rng(2)
        x = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (nscen, 1))
        x[:, 9] = rng.uniform(-12.0, -8.0, nscen)
        est = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (nscen, 1))
        est[:, 1] += rng.normal(0, 0.05, nscen); est[:, 5] += rng.normal(0, 0.02, nscen)
        P = np.tile(np.diag(O.EKF_P0_DIAG), (nscen, 1, 1))
        noise = rng.standard_normal((nsteps, nscen, 7)) * O.EKF_R_DIAG
        t0 = time.perf_counter()
        orc.ofl_run(cfg, x, est, P, 15.0, nsteps, noise=noise, record=False)
        dtm = time.perf_counter() - t0
        what = "output-feedback closed-loop steps (OFL controller + EKF, noisy measurements"
    else:
        st = initial_states(nscen, 0)
        laps = np.zeros(nscen, np.int32); pp = np.zeros(nscen)
        t0 = time.perf_counter()
        orc.skidpad_run(cfg, st, laps, pp, VX, nsteps, record=False)
        dtm = time.perf_counter() - t0
        what = f"closed-loop steps ({kind} controller, vx={VX}"
    out = {"value": nscen * nsteps / dtm, "unit": "scenario-steps/s", "cores": threads, "kind": "port",
           "sample": f"{nscen} scenarios x {nsteps} {what}, N={N}; transient included), oracle/mpc_oracle.c double "
                     f"build, unconstrained minimiser first, OpenMP over scenarios on {threads} threads, {dtm:.1f} s"}
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
    orc.set_unconstrained_first(False)
    return out


def measured_traffic(args):
    """HBM bytes per launch of the timed region from the PMC passes kept under profiles/ (rocprofv3 cannot run inside
    this process); only for a configuration those passes were taken on, else None."""
    import glob
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))       # the latest set of passes
    if args.per_step_launch or not tables:
        return None
    with open(tables[-1]) as fh:
        table = json.load(fh)
    key = f"{args.workload}:B={args.batch or WORKLOADS[args.workload]['B']}:steps={args.steps}:warmup={args.warmup}"
    return table.get(key, {}).get("hbm_bytes_per_launch")


WORKLOADS = {
    "c2": {"kind": "L", "N": 50, "Tf": 0.5, "B": 4096, "kernel": "k_mpc<50, fixed-model>",
           "text": "LMPC2 batched: 4096 perturbed skidpad initial states per GPU, fixed A,B (vx=8), N=50, dt=0.01, closed loop"},
    "c3": {"kind": "LPV", "N": 80, "Tf": 0.8, "B": 8192, "kernel": "k_mpc_lanes (stage-wise Riccati interior point, one lane per scenario)",
           "text": "LPVMPC batched: 8192 scenarios per GPU, per-scenario vx~U(6,12), N=80, dt=0.01, lin_mode reference, closed loop"},
    "c4": {"kind": "OFL", "N": 50, "Tf": 0.5, "B": 8192, "kernel": "k_mpc<50, fixed-model> (MODE_OFL)",
           "text": "OFLMPC2 + EKF batched: 8192 scenarios per GPU, step reference, side force d_f~U(-12,-8), noisy "
                   "measurements, N=50, dt=0.01, output-feedback closed loop"},
    "c5": {"kind": "NL", "N": 50, "Tf": 0.5, "B": 131072, "kernel": "k_mpc_nl<50>",
           "text": "NLMPC batched: 131072 scenarios per GPU, tyre/mass sweep, N=50, dt=0.01, 3 SQP iterations per step, closed loop"},
}


def build_workload(name: str, B: int, rank: int, dev, steps_total: int, qp_method: str = "auto"):
    """Returns (run(n), controller, status tensor, iters setter)."""
    import torch
    from mpc_amd.batched import SkidpadBatch, StepOFLBatch
    w = WORKLOADS[name]
    if name == "c2":
        sim = SkidpadBatch(w["N"], w["Tf"], initial_states(B, rank), model="L", device=dev, qp_method=qp_method)
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
        sim = StepOFLBatch(w["N"], w["Tf"], true0, est0, device=dev, qp_method=qp_method)
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


SUM_KEYS = ("scenario_steps", "ipm_iterations", "constrained_solves", "status_nonzero", "sqp_iterations", "sum_abs_u0",
            "active_set_solves", "active_set_iterations", "active_set_handed_on", "active_set_flops",
            # closed-loop statistics of the timed region (mpc_get_stats slots 16-19, 23)
            "stage_cost", "lateral_error_sq", "heading_pair_violation", "input_sq", "closed_loop_steps")
MAX_KEYS = ("elapsed_s", "max_abs_u0", "max_complementarity", "max_residual",
            "max_bound_violation", "max_heading_pair_violation", "max_predicted_heading_pair_violation")


def reduce_statistics(sums: dict, maxima: dict, device=None) -> tuple[dict, dict]:
    """The only exchange between the ranks: the per-rank statistics as one SUM and one MAX vector through
    mpc_amd.sharding.allreduce_stats (RCCL for device tensors, gloo on the CPU)."""
    import torch
    from mpc_amd.sharding import allreduce_stats
    sv = torch.tensor([float(sums[k]) for k in SUM_KEYS], dtype=torch.float64, device=device)
    mv = torch.tensor([float(maxima[k]) for k in MAX_KEYS], dtype=torch.float64, device=device)
    allreduce_stats(sv, mv)
    return dict(zip(SUM_KEYS, sv.tolist())), dict(zip(MAX_KEYS, mv.tolist()))


def spawn_ranks(n: int, argv: list[str]) -> int:
    """Starts n ranks of this script under torch.distributed.run as a child process (nothing in this process has
    touched a GPU) and passes the child's output through; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank: int, world: int) -> None:
    """The multi-rank path without a GPU: every rank fabricates the statistics of "its" scenarios, the vectors go
    through the same reduction, rank 0 prints the line."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    from mpc_amd.sharding import shard_bounds
    B = args.batch or WORKLOADS[args.workload]["B"]
    lo, hi = shard_bounds(B * world, rank, world)
    sums = {"scenario_steps": (hi - lo) * args.steps, "ipm_iterations": 2.0 * (hi - lo) * args.steps,
            "constrained_solves": 0.5 * (hi - lo) * args.steps, "status_nonzero": 0, "sqp_iterations": 0, "sum_abs_u0": 0.1 * (hi - lo),
            "active_set_solves": 0.4 * (hi - lo) * args.steps, "active_set_iterations": 1.0 * (hi - lo) * args.steps,
            "active_set_handed_on": 0.1 * (hi - lo) * args.steps, "active_set_flops": 1e4 * (hi - lo) * args.steps,
            "stage_cost": 3.0 * (hi - lo) * args.steps, "lateral_error_sq": 0.01 * (hi - lo) * args.steps,
            "heading_pair_violation": 1e-15 * (hi - lo) * args.steps, "input_sq": 0.04 * (hi - lo) * args.steps,
            "closed_loop_steps": (hi - lo) * args.steps}
    maxima = {"elapsed_s": 1.0 + rank, "max_abs_u0": 0.1 * (rank + 1), "max_complementarity": 1e-16, "max_residual": 1e-13,
              "max_bound_violation": 0.0, "max_heading_pair_violation": 1e-14 * (rank + 1), "max_predicted_heading_pair_violation": 0.0}
    sums, maxima = reduce_statistics(sums, maxima)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": sums["scenario_steps"] / maxima["elapsed_s"], "unit": "scenario-steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "statistics": {"sum": sums, "max": maxima}}))
    if world > 1:
        dist.destroy_process_group()


METRIC = "MPC QP solves/sec (node, batched skidpad scenarios) + p50 step latency"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="scenarios per GPU (default: the workload's)")
    ap.add_argument("--qp-method", choices=["auto", "interior_point", "active_set"], default="auto",
                    help="c2 / c4: how constrained solves of the fixed-model path are finished (mpc_config.qp_method)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-launch", action="store_true", help="one kernel launch per closed-loop step in the timed region")
    ap.add_argument("--latency-launches", type=int, default=None, help="single-step launches timed after the run (default 200; 0 = none)")
    ap.add_argument("--dry-run", action="store_true", help="multi-rank plumbing only, on the CPU (gloo), no kernel")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = {"c2": 1000, "c3": 100, "c4": 300, "c5": 50}[args.workload]
    transient = {"c2": 50, "c3": 10, "c4": 20, "c5": 5}[args.workload]
    if args.warmup is None:
        args.warmup = transient
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))          # this process never touches a GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting {world}", file=sys.stderr)
    if args.dry_run:
        return dry_run(args, rank, world)
    # The workloads time closed-loop steps from step `transient` of every scenario on (the heavily constrained start from
    # the perturbed initial states is warm-up by definition).  A run asked for fewer warm-up steps advances the scenarios
    # through the missing ones during set-up, so the timed steps are the same steps of the same trajectories.
    settle = max(0, transient - args.warmup)

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    B = args.batch or wl["B"]
    N = wl["N"]
    nlat = args.latency_launches if args.latency_launches is not None else (0 if args.per_step_launch else (50 if args.workload == "c5" else 200))
    extra = nlat + (1000 if args.workload == "c2" else 0)
    sim, run = build_workload(args.workload, B, rank, dev, args.steps + args.warmup + settle + extra, args.qp_method)
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
            run(1)
    elif args.warmup:
        run(args.warmup)
    barrier()
    t_warm = time.perf_counter() - tw0

    # Timed region: exactly `steps` closed-loop steps of every scenario.  Default: ONE launch carries each
    # scenario through all steps (scenarios are independent); --per-step-launch issues one launch per step.
    # Launch durations come from HIP events on the stream the kernels are launched on (torch's current stream).
    nlaunch = args.steps if args.per_step_launch else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlaunch)]
    iters_hist = torch.zeros((nlaunch, B), dtype=torch.int32, device=dev)
    barrier()
    t0 = time.perf_counter()
    if args.per_step_launch:
        for i in range(args.steps):
            sim.iters = iters_hist[i]
            ev[i][0].record()
            run(1)
            ev[i][1].record()
    else:
        sim.iters = iters_hist[0]
        ev[0][0].record()
        run(args.steps)
        ev[0][1].record()
    barrier()
    elapsed = time.perf_counter() - t0

    st16 = sim.ocp.stats()
    status_timed = sim.status.clone()
    n_constr = float((iters_hist > 0).sum().item()) if args.per_step_launch else float(st16["n_constrained"])
    sums = {"scenario_steps": B * args.steps, "ipm_iterations": float(iters_hist.sum().item()), "constrained_solves": n_constr,
            "status_nonzero": float((status_timed != 0).sum().item()), "sqp_iterations": float(st16["sum_sqp_iterations"]),
            "sum_abs_u0": float(st16["sum_abs_u0"]),
            # the active-set counters describe the last launch: the timed region in the fused mode, its last step otherwise
            "active_set_solves": float(st16["n_active_set_solves"]), "active_set_iterations": float(st16["sum_active_set_iterations"]),
            "active_set_handed_on": float(st16["n_active_set_too_large"] + st16["n_active_set_unsettled"]),
            "active_set_flops": float(st16["sum_active_set_flops"]),
            # closed-loop statistics: the last launch too (every step of the timed region in the fused mode)
            "stage_cost": st16["sum_stage_cost"], "lateral_error_sq": st16["sum_lateral_error_sq"],
            "heading_pair_violation": st16["sum_heading_pair_violation"], "input_sq": st16["sum_input_sq"],
            "closed_loop_steps": st16["closed_loop_steps"]}
    maxima = {"elapsed_s": elapsed, "max_abs_u0": float(st16["max_abs_u0"]), "max_complementarity": float(st16["max_complementarity"]),
              "max_residual": float(st16["max_residual"]), "max_bound_violation": st16["max_bound_violation"],
              "max_heading_pair_violation": st16["max_heading_pair_violation"],
              "max_predicted_heading_pair_violation": st16["max_predicted_heading_pair_violation"]}
    sums, maxima = reduce_statistics(sums, maxima, dev)       # RCCL: the only collective of the path
    elapsed = maxima["elapsed_s"]
    kernel_ms = np.array([a.elapsed_time(b) for a, b in ev])
    steps_per_launch = args.steps / nlaunch

    # After the timed region: the step-latency distribution (one launch per closed-loop step, a HIP event pair around
    # each) and, for c2, a 1000-step launch in the steady state that follows the transient.
    lat_ms = kernel_ms if args.per_step_launch else None
    if nlat > 0:
        sim.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        lev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlat)]
        for a, b in lev:
            a.record(); run(1); b.record()
        torch.cuda.synchronize(dev)
        lat_ms = np.array([a.elapsed_time(b) for a, b in lev])
    steady = None
    if args.workload == "c2" and not args.per_step_launch:
        first = settle + args.warmup + args.steps + nlat
        sa, sb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sa.record(); run(1000); sb.record()
        torch.cuda.synchronize(dev)
        steady = {"first_closed_loop_step": first, "steps": 1000, "launch_ms": sa.elapsed_time(sb),
                  "scenario_steps_per_s_per_gpu": B * 1000 / (sa.elapsed_time(sb) * 1e-3),
                  "mean_ipm_iterations_per_scenario_step": float(sim.iters.sum().item()) / (B * 1000)}

    if rank == 0:
        total = sums["scenario_steps"]
        mean_iters = sums["ipm_iterations"] / total
        mean_sqp = sums["sqp_iterations"] / total
        ncomp = weighted_reference_components(sim.ocp) if args.workload in ("c2", "c4") else 3
        as_scale = 1.0 if not args.per_step_launch else float(args.steps)      # per-step launches: the last step stands for all
        flops = algorithmic_flops(args.workload, N, mean_iters, mean_sqp, ncomp, extra=as_scale * sums["active_set_flops"] / total,
                                  aset_qps=as_scale * sums["active_set_solves"] / total if args.workload == "c5" else 0.0)
        launch_s = float(kernel_ms.mean()) * 1e-3
        ach_tflops = flops * B * steps_per_launch / launch_s / 1e12
        out = {
            "metric": METRIC,
            "value": total / elapsed,
            "unit": "scenario-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "p50_step_latency_ms": float(np.median(lat_ms)) if lat_ms is not None else None,
            "p99_step_latency_ms": float(np.percentile(lat_ms, 99)) if lat_ms is not None else None,
            "step_latency_sample": (f"{len(lat_ms)} single-step launches of all {B} scenarios, HIP events per launch, "
                                    + ("the timed region itself" if args.per_step_launch else
                                       f"closed-loop steps {settle + args.warmup + args.steps}.. after the timed region")) if lat_ms is not None else None,
            "launch_mode": "per-step" if args.per_step_launch else f"fused ({args.steps} steps per launch)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["text"], "name": args.workload, "scenarios_per_gpu": B, "horizon": N,
                       "parallelism": f"scenario-sharded x{world}",
                       "first_timed_closed_loop_step": settle + args.warmup, "setup_steps": settle},
            "solver": {"mean_ipm_iterations_per_scenario_step": mean_iters,
                       "constrained_qp_per_scenario_step": sums["constrained_solves"] / total,
                       "mean_sqp_iterations_per_scenario_step": mean_sqp if args.workload == "c5" else None,
                       "active_set": None if args.workload not in ("c2", "c4", "c5") else {
                           "solves_per_scenario_step": as_scale * sums["active_set_solves"] / total,
                           "iterations_per_solve": sums["active_set_iterations"] / max(1.0, sums["active_set_solves"] + sums["active_set_handed_on"]),
                           "handed_to_interior_point_per_scenario_step": as_scale * sums["active_set_handed_on"] / total,
                           "flops_per_scenario_step": as_scale * sums["active_set_flops"] / total},
                       "status_nonzero_last_launch": int(sums["status_nonzero"]),
                       "status_histogram_rank0": {str(k): int(v) for k, v in enumerate(torch.bincount(status_timed.long()).tolist()) if v},
                       "max_abs_u0": maxima["max_abs_u0"], "max_complementarity": maxima["max_complementarity"],
                       "max_residual": maxima["max_residual"],
                       "transient_scenario_steps_per_s": (B * world * args.warmup / t_warm) if args.warmup else None,
                       "steady_state": steady},
            # cost and constraint-violation statistics of the timed steps, all-reduced over the ranks (sums add, maxima max)
            "closed_loop": {"scenario_steps_covered": sums["closed_loop_steps"],
                            "mean_stage_cost_per_step": sums["stage_cost"] / max(1.0, sums["closed_loop_steps"]),
                            "rms_lateral_error_m": float(np.sqrt(sums["lateral_error_sq"] / max(1.0, sums["closed_loop_steps"]))),
                            "rms_input_rad_s": float(np.sqrt(sums["input_sq"] / max(1.0, sums["closed_loop_steps"]))),
                            "mean_heading_pair_violation": sums["heading_pair_violation"] / max(1.0, sums["closed_loop_steps"]),
                            "max_heading_pair_violation": maxima["max_heading_pair_violation"],
                            "max_predicted_heading_pair_violation": maxima["max_predicted_heading_pair_violation"],
                            "max_bound_violation": maxima["max_bound_violation"]},
        }
        fp64 = {"achieved": ach_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach_tflops / PEAK_FP64_TFLOPS,
                "algorithmic_flops_per_scenario_step": flops}
        common = {"traffic": measured_traffic(args), "kernel": wl["kernel"], "avg_launch_ms": launch_s * 1e3,
                  "scenario_steps_per_launch": B * steps_per_launch}
        mbytes = method_bytes(args.workload, N, mean_iters)
        if mbytes is not None:
            # the lane-per-scenario kernel streams its stage-wise workspace from HBM in every sweep: HBM is its nearer roof
            ach = mbytes * B * steps_per_launch / launch_s / 1e9
            out["roofline"] = dict({"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS},
                                   **common, algorithmic_bytes_per_scenario_step=mbytes, fp64=fp64,
                                   note="workspace streaming of the stage-wise interior point (method bytes); at this batch size a wave "
                                        "also runs alone on its compute unit, at the issue rate of its own instruction stream")
        else:
            out["roofline"] = dict({"bound": "mfma"}, **fp64, **common,
                                   note="FP64 vector/matrix peak; the kernels are latency- and issue-bound, not FP64-throughput-bound",
                                   hbm={"achieved_GBs": algorithmic_bytes(args.workload, N) * B * steps_per_launch / launch_s / 1e9,
                                        "peak_GBs": PEAK_HBM_GBS})
        if world == 1 and not args.no_cpu_baseline:
            # sized for 10-30 s of host work each
            sample = {"c2": (4096, 1000), "c3": (2048, 400), "c4": (2048, 300), "c5": (4096, 100)}[args.workload]
            out["cpu_baseline"] = cpu_baseline(args.workload, *sample)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
