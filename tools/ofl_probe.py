"""Active-set statistics of the fused output-feedback loop on its start transient (every input fixed by a bound)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_amd.batched import StepOFLBatch
B, n = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(2)
x = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, -10.0, 0.0]), (B, 1))
x[:, 1] = rng.uniform(-0.2, 0.2, B)
for method in ("auto", "interior_point"):
    sim = StepOFLBatch(50, 0.5, x, qp_method=method)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); xs, us, es = sim.run(n, record=True); t1.record()
    torch.cuda.synchronize()
    st = sim.ocp.stats()
    print(method, f"{t0.elapsed_time(t1):.2f} ms", {k: float(v) for k, v in st.items() if "active" in k or k in ("sum_iterations", "n_constrained")}, np.bincount(sim.status.cpu().numpy()))
