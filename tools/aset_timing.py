"""Per-phase cycles of the constrained step (developer build with MPC_DEV_DEFINES=-DASET_TIMING): a few scenarios of the
c2 workload, steps 50..70, one wave per SIMD at most, so that the stamps measure one wave's chain."""
import ctypes as C, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import initial_states
from mpc_amd.batched import SkidpadBatch
from mpc_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
f = lib.mpc_debug_aset_timing if hasattr(lib, "mpc_debug_aset_timing") else C.CDLL(lib._name).mpc_debug_aset_timing
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
st = initial_states(4096, 0)[:B]
s = SkidpadBatch(50, 0.5, st, model="L", device=torch.device("cuda:0"))
s.run(45); s.run(5); torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
f(buf, 1)
s.run(20); torch.cuda.synchronize()
f(buf, 0)
t = np.array(list(buf), dtype=np.float64)
names = ["lists", "gather", "factor", "solve+mult", "combine", "scan+exchange", "certificate", "check+record", "predict", "backoff read", "aset call", "counters"]
iters, passes, solves = t[16], t[17], t[18]
print(f"B={B}: constrained steps {int(t[19])} ({t[12] / max(t[19], 1):.0f} cycles each), unconstrained {int(t[20])} ({t[13] / max(t[20], 1):.0f} cycles each)")
print(f"solves {int(solves)}, iterations {iters / max(solves, 1):.2f} per solve, refinement passes {passes / max(solves, 1):.2f} per solve")
for i in range(6):
    print(f"  {names[i]:14s} {t[i] / max(iters, 1):8.0f} cycles per iteration   {t[i] / max(solves, 1):8.0f} per solve")
for i in range(6, 12):
    print(f"  {names[i]:14s} {'':8s}                        {t[i] / max(solves, 1):8.0f} per solve")
