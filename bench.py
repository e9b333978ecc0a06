#!/usr/bin/env python3
"""Headline benchmark: closed-loop LMPC skidpad scenario-steps per second.

One "step" = one fused closed-loop step (plan -> linearise/condense -> QP solve -> RK4 plant -> lap count)
for every scenario of the batch, one kernel launch per GPU.  Workload at N=1 (BASELINE.json configs[1]):
LMPC2, B = 4096 perturbed skidpad initial states, horizon 50, dt 0.01, vx 8, numpy default_rng(0) draws,
state resident in HBM.  With --gpus G every rank runs its own 4096 scenarios (weak scaling, no collective
in the loop; one RCCL all-reduce of the statistics vector at the end).

  python bench.py --gpus N --steps K --warmup W        (N>1: under torch.distributed.run, one rank per GPU)
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

B_PER_GPU = 4096
HORIZON = 50
TF = 0.5
VX = 8.0
PEAK_FP64_TFLOPS = 78.6      # MI355X FP64 vector = matrix peak (datasheet); the guide's tables carry no f64 row
PEAK_HBM_GBS = 8000.0


def initial_states(B: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    py = rng.uniform(-0.3, 0.3, B); psi = rng.uniform(-0.1, 0.1, B)
    vy = rng.uniform(-0.2, 0.2, B); r = rng.uniform(-0.2, 0.2, B); dl = rng.uniform(-0.05, 0.05, B)
    return np.stack([np.full(B, 15.0), py, np.cos(psi), np.sin(psi), np.full(B, VX), vy, r, dl], axis=1)


def algorithmic_flops(N: int, mean_iters: float, frac_constrained: float) -> float:
    """FP64 flops one scenario-step needs with this method (symmetry and the zero weights exploited):
    planner 60(N+1); stage model 30N; prediction columns + free response + gradient 16N^2; Hessian
    sum_k 3 weights * k(k+1)/2 pairs * 2 = N^3 plus dense terminal 6N^2 + 72N; unconstrained solve
    N^3/3 + 2N^2; RK4 400.  Each interior-point iteration: H*U 2N^2, Newton matrix 3N^2, LDL' N^3/3,
    two solves 4N^2, vector updates 100N."""
    fixed = 60 * (N + 1) + 30 * N + 16 * N * N + N ** 3 + 6 * N * N + 72 * N + N ** 3 / 3 + 2 * N * N + 400
    per_iter = N ** 3 / 3 + 9 * N * N + 100 * N
    return fixed + per_iter * mean_iters


def algorithmic_bytes(N: int) -> float:
    """HBM bytes one scenario-step must move: plant state 8 + lap/progress/speed 3 + u0 1 doubles, read and
    written once.  (The condensed Hessian scratch, N*N doubles per scenario, is method traffic, not algorithmic.)"""
    return (8 + 3 + 1) * 8 * 2


def cpu_baseline(nscen: int, nsteps: int) -> dict:
    from oracle import oracle as O
    O.build()
    orc = O.Oracle("f64")
    cfg = O.make_cfg("L", HORIZON, TF)
    st = initial_states(nscen, 0)
    laps = np.zeros(nscen, np.int32); pp = np.zeros(nscen)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    t0 = time.perf_counter()
    orc.skidpad_run(cfg, st, laps, pp, VX, nsteps, record=False)
    dtm = time.perf_counter() - t0
    return {"value": nscen * nsteps / dtm, "unit": "scenario-steps/s", "cores": threads, "kind": "port",
            "sample": f"{nscen} scenarios x {nsteps} closed-loop steps from the same initial states "
                      f"(transient included), oracle/mpc_oracle.c double build, OpenMP over scenarios, {dtm:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="scenarios per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-step-launch", action="store_true", help="one kernel launch per closed-loop step")
    args = ap.parse_args()

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

    from mpc_amd.batched import SkidpadBatch

    B = args.batch
    sim = SkidpadBatch(HORIZON, TF, initial_states(B, rank), model="L", device=dev)

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
        sim.run(args.warmup)
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
        sim.run(args.steps)
        ev[0][1].record()
    barrier()
    elapsed = time.perf_counter() - t0

    st16 = sim.ocp.stats()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_constr = float((iters_hist > 0).sum().item()) if args.per_step_launch else float(st16["n_constrained"])
    stats = torch.tensor([float(iters_hist.sum().item()), n_constr, float((sim.status != 0).sum().item()),
                          float(sim.state[:, 1].abs().sum().item())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)       # RCCL: the only collective of the path
    elapsed = float(t.item())
    kernel_ms = np.array([a.elapsed_time(b) for a, b in ev])
    steps_per_launch = args.steps / nlaunch

    if rank == 0:
        total = B * world * args.steps
        mean_iters = float(stats[0].item()) / total
        frac_constrained = float(stats[1].item()) / total
        flops = algorithmic_flops(HORIZON, mean_iters, frac_constrained)
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
            "config": {"workload": "LMPC2 batched: 4096 perturbed skidpad initial states per GPU, fixed A,B (vx=8), "
                                   "N=50, dt=0.01, closed loop", "scenarios_per_gpu": B, "horizon": HORIZON,
                       "parallelism": f"scenario-sharded x{world}"},
            "solver": {"mean_ipm_iterations_per_scenario_step": mean_iters,
                       "fraction_constrained_steps": frac_constrained,
                       "status_nonzero_last_step": int(stats[2].item()),
                       "transient_scenario_steps_per_s": (B * world * args.warmup / t_warm) if args.warmup else None},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / PEAK_FP64_TFLOPS, "traffic": None,
                         "kernel": "k_mpc<50>", "avg_launch_ms": launch_s * 1e3, "scenario_steps_per_launch": B * steps_per_launch,
                         "algorithmic_flops_per_scenario_step": flops,
                         "hbm": {"achieved_GBs": algorithmic_bytes(HORIZON) * B * steps_per_launch / launch_s / 1e9,
                                 "peak_GBs": PEAK_HBM_GBS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(1024, 300)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
