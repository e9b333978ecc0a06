"""The driver's c2 window (closed-loop steps 50..70) for subsets of the 4096 scenarios: how much of its time is the
dependent chain of one scenario (the time of a single scenario) and how much is throughput."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import initial_states
from mpc_amd.batched import SkidpadBatch

kw = {"qp_method": sys.argv[1]} if len(sys.argv) > 1 else {}
st = initial_states(4096, 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n in (1, 8, 64, 512, 1024, 2048, 4096):
    idx = np.sort(rng.choice(4096, n, replace=False))
    best = 1e9
    for rep in range(3):
        s = SkidpadBatch(50, 0.5, st[idx], model="L", device=dev, **kw)
        s.run(45); s.run(5); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); s.run(20); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    stt = s.ocp.stats()
    print(f"{n:5d} scenarios: {best:.3f} ms; constrained {stt['n_constrained'] / (20 * n):.2f} per step, active-set iterations {stt['sum_active_set_iterations'] / (20 * n):.2f} per step, ipm {stt['sum_iterations'] / (20 * n):.2f}", flush=True)
