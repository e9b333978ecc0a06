"""Throughput of the per-scenario condensing kernel (k_mpc<50, general>): LPV and cache-less L closed loops."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from bench import initial_states
from mpc_amd.batched import SkidpadBatch
B, N, Tf = 8192, 50, 0.5
for model, kw in (("LPV", {}), ("LPV", {"qp_method": "interior_point"}), ("L", {"fixed_model_cache": False}),
                  ("L", {"fixed_model_cache": False, "qp_method": "interior_point"})):
    sim = SkidpadBatch(N, Tf, initial_states(B, 0), model=model, **kw)
    sim.run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); sim.run(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(model, kw, "%.3e scenario-steps/s" % (B * 100 / dt), "iters/step %.2f" % (sim.iters.sum().item() / (B * 100)), "status", torch.bincount(sim.status.long()).tolist())
