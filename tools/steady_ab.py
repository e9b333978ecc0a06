"""Steady-state (no bound active) and driver-window time of the fixed-model kernel on the c2 workload, for the library
MPC_AMD_LIBRARY names or the in-tree one: tools/steady_ab.py [qp_method]"""
import sys, time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import initial_states
from mpc_amd.batched import SkidpadBatch
kw = {"qp_method": sys.argv[1]} if len(sys.argv) > 1 else {}
B = 4096
st = initial_states(B, 0)
dev = torch.device("cuda:0")
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    s = SkidpadBatch(50, 0.5, st, model="L", device=dev, **kw)
    s.run(45); s.run(5); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); s.run(20); ev[1].record(); s.run(200); ev[2].record(); s.run(1000); ev[3].record()
    torch.cuda.synchronize()
    print(f"rep {rep}: window 20 steps {ev[0].elapsed_time(ev[1]):.3f} ms, next 200 {ev[1].elapsed_time(ev[2]):.3f} ms, steady 1000 {ev[2].elapsed_time(ev[3]):.3f} ms", flush=True)
