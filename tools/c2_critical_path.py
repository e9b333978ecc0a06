"""Critical path of the c2 workload: the heaviest scenarios of the batch run alone (one wave each)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch

B = 4096
st = initial_states(B, 0)
dev = torch.device("cuda:0")
sim = SkidpadBatch(50, 0.5, st, model="L", device=dev)
sim.run(50); sim.run(1000); torch.cuda.synchronize()
it = sim.iters.cpu().numpy()
heavy = np.argsort(-it)[:4]; light = np.argsort(it)[:4]
srt = np.argsort(-it)
for name, idx in (("heaviest", heavy[:1]), ("lightest", light[:1]), ("heaviest 1024", srt[:1024]),
                  ("heaviest 1024 + lightest 1024", np.concatenate([srt[:1024], srt[-1024:]])),
                  ("heaviest 2048", srt[:2048]), ("lightest 2048", srt[-2048:]), ("all", np.arange(B))):
    s2 = SkidpadBatch(50, 0.5, st[idx], model="L", device=dev)
    s2.run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); s2.run(1000); torch.cuda.synchronize(); t1 = time.perf_counter()

    i2 = s2.iters.cpu().numpy()
    print(f"{name:30s} B={len(idx):5d}  1000 steps {1e3*(t1-t0):7.2f} ms   iters mean {i2.mean():.0f} max {i2.max()}")
