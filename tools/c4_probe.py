"""Active-set statistics of the output-feedback workload (c4: noisy measurements) per window of steps."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
sim, run = bench.build_workload("c4", B, 0, dev, 320)
keys = ("n_constrained", "n_active_set_solves", "sum_active_set_iterations", "n_active_set_too_large", "n_active_set_unsettled", "sum_iterations")
import ctypes as C
from mpc_amd import _lib
lib = _lib.load()
dbg = getattr(lib, "mpc_debug_aset_timing", None) if hasattr(lib, "mpc_debug_aset_timing") else None
if dbg is not None:
    dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 24)()
for n in (20, 50, 50, 100, 100):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); run(n); t1.record(); torch.cuda.synchronize()
    st = sim.ocp.stats()
    tot = B * n
    if dbg is not None:
        dbg(buf, 1)
        print("      developer counters: iterations", buf[16], "not settled", buf[21], "not certified", buf[22], "final check failed", buf[23])
    print(f"{n:4d} steps {t0.elapsed_time(t1):8.2f} ms  " + "  ".join(f"{k[2:] if k.startswith('n_') else k}={float(st[k]) / tot:.3f}" for k in keys))
