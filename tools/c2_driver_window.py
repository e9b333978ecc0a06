"""The driver's c2 window (closed-loop steps 50..70 of 4096 scenarios) taken apart: iterations per scenario, the time of
the fused launch against subsets run alone (critical path) and against per-step launches."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch

B = 4096
st = initial_states(B, 0)
dev = torch.device("cuda:0")


def window(idx, per_step=False):
    s = SkidpadBatch(50, 0.5, st[idx], model="L", device=dev)
    s.run(45); s.run(5); torch.cuda.synchronize()
    s.iters.zero_()
    t0 = time.perf_counter()
    if per_step:
        for _ in range(20):
            s.run(1)
    else:
        s.run(20)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), s.iters.cpu().numpy().copy()


ms, it = window(np.arange(B))
print(f"all 4096 fused: {ms:.2f} ms, iterations per scenario over the 20 steps: mean {it.mean():.1f} max {it.max()} "
      f"p90 {np.percentile(it, 90):.0f}; scenarios with none {int((it == 0).sum())}")
ms, _ = window(np.arange(B), per_step=True)
print(f"all 4096 per-step launches: {ms:.2f} ms")
srt = np.argsort(-it)
for name, idx in (("lightest 1024", srt[-1024:]), ("lightest 64", srt[-64:]), ("heaviest 512 + lightest 512", np.concatenate([srt[:512], srt[-512:]])), ("heaviest 1024 + lightest 1024", np.concatenate([srt[:1024], srt[-1024:]])), ("heaviest 2048 + lightest 1024", np.concatenate([srt[:2048], srt[-1024:]])),
                  ("heaviest 2048 + next 1024", srt[:3072]), ("ranks 1024..3072", srt[1024:3072]), ("heaviest 1", srt[:1]), ("heaviest 64", srt[:64]), ("heaviest 1024", srt[:1024]), ("heaviest 2048", srt[:2048]),
                  ("lightest 2048", srt[-2048:])):
    ms, i2 = window(np.sort(idx))
    print(f"{name:16s}: {ms:.2f} ms  (iterations mean {i2.mean():.1f} max {i2.max()})")
