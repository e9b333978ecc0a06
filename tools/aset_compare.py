"""Active-set path against the interior point on the c2 workload: agreement of the closed loops, share of the constrained
solves each method finishes, and the time of the driver's window (steps 50..70) and of the start transient."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = initial_states(B, 0)
dev = torch.device("cuda:0")


def run(method, blocks):
    s = SkidpadBatch(50, 0.5, st, model="L", device=dev, qp_method=method)
    out = []
    for n in blocks:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xs, us = s.run(n, record=True)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
        stt = s.ocp.stats()
        out.append((ms, us.cpu().numpy(), xs.cpu().numpy(), stt, s.status.cpu().numpy().copy()))
    return out


blocks = [50, 20, 200]
for rep in range(2):
    a = run("auto", blocks); b = run("interior_point", blocks)
for (ms_a, ua, xa, sa, sta), (ms_b, ub, xb, sb, stb), n in zip(a, b, blocks):
    du = np.abs(ua - ub).max(); dx = np.abs(xa - xb).max()
    print(f"{n:4d} steps: auto {ms_a:7.2f} ms  ipm {ms_b:7.2f} ms  max|du| {du:.2e} max|dx| {dx:.2e}  status!=0 {int((sta != 0).sum())}/{int((stb != 0).sum())}")
    print("      auto:", {k: float(v) for k, v in sa.items() if k in ("sum_iterations", "n_constrained", "n_active_set_solves", "sum_active_set_iterations", "max_residual", "n_active_set_too_large", "n_active_set_unsettled")})
    print("      ipm :", {k: float(v) for k, v in sb.items() if k in ("sum_iterations", "n_constrained", "n_active_set_solves", "sum_active_set_iterations", "max_residual", "n_active_set_too_large", "n_active_set_unsettled")})
