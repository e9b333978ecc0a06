"""Solver statistics of the NL workload (c5) per window of steps: how its constrained QPs are finished."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
sim, run = bench.build_workload("c5", B, 0, dev, 320)
keys = ("n_constrained", "n_active_set_solves", "sum_active_set_iterations", "n_active_set_too_large", "n_active_set_unsettled", "sum_iterations")
for n in (5, 25, 25, 50, 100):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); run(n); t1.record(); torch.cuda.synchronize()
    st = sim.ocp.stats()
    tot = B * n
    print(f"{n:4d} steps {t0.elapsed_time(t1):8.2f} ms  " + "  ".join(f"{k[2:] if k.startswith('n_') else k}={float(st[k]) / tot:.3f}" for k in keys), " status", np.bincount(sim.status.cpu().numpy()))
