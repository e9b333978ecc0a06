"""Largest per-step distance of the fixed-model GPU path (active-set solve, or the interior point with
qp_method=interior_point) to the 80-bit checker over closed loops from spread starts; a report, not collected by pytest:
  python tests/report_active_set_accuracy.py [qp_method] [scenarios] [steps]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from mpc_amd.batched import SkidpadBatch
from tests.test_gpu_active_set import _starts

method = sys.argv[1] if len(sys.argv) > 1 else "auto"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 90
N, Tf = 50, 0.5
orc80 = O.Oracle("f80")
st = _starts(B, 11)
cfg = O.make_cfg("L", N, Tf)
sim = SkidpadBatch(N, Tf, st, model="L", qp_method=method)
xs, us = sim.run(nsteps, record=True)
torch.cuda.synchronize()
xs = xs.cpu().numpy(); us = us.cpu().numpy()
worst = 0.0; worst_rel = 0.0; n = 0
for b in range(0, B, 4):
    lap = 0
    for t in range(nsteps):
        x = st[b] if t == 0 else xs[t - 1, b]
        _, _, _, _, u_ref, _ = orc80.skidpad_step(cfg, x.copy(), lap, 0.0, 8.0)
        d = abs(us[t, b] - u_ref)
        worst = max(worst, d); worst_rel = max(worst_rel, d / (1e-5 * max(abs(u_ref), 1e-3 * cfg[10]))); n += 1
        if xs[t, b, 0] > 16.75 and x[0] <= 16.75:
            lap += 1
s = sim.ocp.stats()
print(f"{method}: {n} steps, largest |du| {worst:.3e}, largest share of the bar {worst_rel:.3e}; constrained {s['n_constrained']:.0f}, "
      f"active-set solves {s['n_active_set_solves']:.0f}, unsettled {s['n_active_set_unsettled']:.0f}, interior-point iterations {s['sum_iterations']:.0f}")
