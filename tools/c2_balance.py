"""Per-scenario work distribution of the c2 workload: interior-point iterations per scenario over the timed launch
(a fused launch ends with its slowest scenario)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch

B = 4096
sim = SkidpadBatch(50, 0.5, initial_states(B, 0), model="L", device=torch.device("cuda:0"))
sim.run(50)
torch.cuda.synchronize()
it50 = sim.iters.cpu().numpy().astype(np.int64)
for chunk in (1000,):
    t0 = time.perf_counter(); sim.run(chunk); torch.cuda.synchronize(); t1 = time.perf_counter()
    it = sim.iters.cpu().numpy().astype(np.int64)
    print(f"{chunk} steps {1e3*(t1-t0):.2f} ms  iters/scenario: mean {it.mean():.1f} p50 {np.percentile(it,50):.0f} "
          f"p90 {np.percentile(it,90):.0f} p99 {np.percentile(it,99):.0f} max {it.max()}  zero-iter scenarios {np.mean(it==0):.3f}")
    blocks = it.reshape(-1, 4).max(axis=1)
    print("   block max: mean", blocks.mean(), "max", blocks.max())
print("warmup(50): mean", it50.mean(), "max", it50.max())
