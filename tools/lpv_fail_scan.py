"""Scan the LPV N=80 workload (c3) for closed-loop steps whose interior-point solve ends at the iteration cap, dump the
inputs of those steps (plant state, planner memory, linearisation memory) and replay each alone, cold-started.

  python tools/lpv_fail_scan.py [steps] [B]     -> gpurun_out/lpv_fail/*.npz + a summary on stdout
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bench
from mpc_amd.batched import SkidpadBatch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N, Tf = 80, 0.8
os.makedirs("gpurun_out/lpv_fail", exist_ok=True)


def states(rank=0):
    st = bench.initial_states(B, 1000 + rank)
    st[:, 4] = np.random.default_rng(1 + 7919 * rank).uniform(6.0, 12.0, B)
    return st


def scan(tag, dump, **kw):
    sim = SkidpadBatch(N, Tf, states(), model="LPV", **kw)
    nbad, hist, dumped = 0, np.zeros(101, int), []
    for t in range(steps):
        pre = (sim.state.clone(), sim.lap.clone(), sim.prev_progress.clone(), sim.lin_x.clone(), sim.lin_u.clone())
        sim.step()
        stt = sim.status.cpu().numpy(); it = sim.iters.cpu().numpy()
        hist += np.bincount(np.minimum(it, 100), minlength=101)
        bad = np.nonzero(stt == 2)[0]
        nbad += len(bad)
        for b in bad:
            if dump and len(dumped) < 40:
                f = f"gpurun_out/lpv_fail/{tag}_{len(dumped)}.npz"
                np.savez(f, state=pre[0][b].cpu().numpy(), lap=pre[1][b].cpu().numpy(), pp=pre[2][b].cpu().numpy(),
                         lin_x=pre[3][b].cpu().numpy(), lin_u=pre[4][b].cpu().numpy(), u0=sim.u0[b].cpu().numpy(), t=t, b=b,
                         iters=it[b])
                dumped.append(f)
                print(tag, "bad step", t, "scenario", b, "iters", it[b], "vx", float(pre[0][b, 4]), flush=True)
    print(tag, "status-2 scenario-steps:", nbad, "of", steps * B, "| iteration histogram (nonzero):",
          {i: int(h) for i, h in enumerate(hist) if h and i > 0}, flush=True)
    return dumped


def replay(files, **kw):
    for f in files:
        d = np.load(f)
        sim = SkidpadBatch(N, Tf, d["state"][None], starting_laps=np.array([int(d["lap"])]), model="LPV", **kw)
        sim.prev_progress[:] = float(d["pp"])
        sim.lin_x[0] = torch.as_tensor(d["lin_x"], device=sim.device); sim.lin_u[0] = torch.as_tensor(d["lin_u"], device=sim.device)
        sim.step()
        st = sim.ocp.stats()
        print("replay", os.path.basename(f), kw, "status", int(sim.status[0]), "iters", int(sim.iters[0]), "u0", float(sim.u0[0]),
              "vs", float(d["u0"]), "mu", st["max_complementarity"], "res", st["max_residual"], flush=True)


mode = sys.argv[3] if len(sys.argv) > 3 else "both"
if mode in ("both", "warm"):
    files = scan("warm", True)
    replay(files)                       # alone, cold start (fresh handle: no warm record)
if mode in ("both", "cold"):
    files = scan("cold", True, ipm_warm_start=False)
    replay(files, ipm_warm_start=False)
