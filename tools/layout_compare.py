"""Closed-loop throughput of the two layouts: tools/layout_compare.py <N> <L|LPV>."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
model = sys.argv[2] if len(sys.argv) > 2 else "L"
for B in (256, 1024, 4096, 8192):
    st = initial_states(B, 0)
    if model == "LPV":
        st[:, 4] = np.random.default_rng(1).uniform(6.0, 12.0, B)
    for layout in ("wave", "lane"):
        s = SkidpadBatch(N, N * 0.01, st, model=model, solver_layout=layout)
        s.run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); s.run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{model} N={N} B={B:5d} {layout}: {B * 60 / dt / 1e6:.3f} M scenario-steps/s  ({1e3 * dt / 60:.3f} ms per step)")
