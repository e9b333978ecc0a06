"""Latency of the single-controller class surface (host arrays in, host arrays out, one synchronising call per step):
LOcp.optimize and SkidpadSimulator.step at N = 50, Tf = 0.5 from the default skidpad start."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from mpc_amd.skidpad_simulator import SkidpadSimulator

sim = SkidpadSimulator(50, 0.5, model="L")
for _ in range(50):
    sim.step()
lat = []
for _ in range(500):
    t0 = time.perf_counter()
    sim.step()
    lat.append(time.perf_counter() - t0)
print("SkidpadSimulator.step  p50 %.3f ms  p99 %.3f ms" % (1e3 * np.median(lat), 1e3 * np.percentile(lat, 99)))
ocp = sim.MPC_controller
wp, speeds = sim.get_waypoints()[:2]
x0 = sim.red_state
lat = []
for _ in range(500):
    t0 = time.perf_counter()
    ocp.optimize(x0, wp, speeds)
    lat.append(time.perf_counter() - t0)
print("LOcp.optimize          p50 %.3f ms  p99 %.3f ms" % (1e3 * np.median(lat), 1e3 * np.percentile(lat, 99)))

# the entry point alone (ctypes call with prepared arrays): what a compiled caller of mpc_optimize_host sees
from mpc_amd import _dev
N = 50
x0c = np.ascontiguousarray(x0[:7]); wpc = np.ascontiguousarray(wp); pc = np.ascontiguousarray(np.asarray(speeds, dtype=np.float64)[:N + 1])
U = np.empty(N); traj = np.empty((N + 1, 7)); st = np.zeros(2, np.int32)
f = ocp.solver._lib.mpc_optimize_host
strm = _dev.stream_ptr(ocp.device)
lat = []
for _ in range(1000):
    t0 = time.perf_counter()
    f(ocp.solver.h, strm, 1, x0c.ctypes.data, wpc.ctypes.data, pc.ctypes.data, None, None, None, U.ctypes.data, traj.ctypes.data, st.ctypes.data, st.ctypes.data + 4)
    lat.append(time.perf_counter() - t0)
print("mpc_optimize_host (L)  p50 %.3f ms  p99 %.3f ms" % (1e3 * np.median(lat), 1e3 * np.percentile(lat, 99)))
lat = []
for _ in range(1000):
    t0 = time.perf_counter()
    f(ocp.solver.h, strm, 1, x0c.ctypes.data, wpc.ctypes.data, pc.ctypes.data, None, None, None, U.ctypes.data, None, st.ctypes.data, st.ctypes.data + 4)
    lat.append(time.perf_counter() - t0)
print("  without trajectory   p50 %.3f ms  p99 %.3f ms" % (1e3 * np.median(lat), 1e3 * np.percentile(lat, 99)))

from mpc_amd.controllers import LPVOcp
for NL_, TF_ in ((50, 0.5), (80, 0.8)):
  lpv = LPVOcp(NL_, TF_)
  sim80 = SkidpadSimulator(NL_, TF_, model="L")
  wp80, sp80 = sim80.get_waypoints()[:2]
  for _ in range(50):
      lpv.optimize(sim80.red_state, wp80, sp80)
  lat = []
  for _ in range(500):
      t0 = time.perf_counter()
      lpv.optimize(sim80.red_state, wp80, sp80)
      lat.append(time.perf_counter() - t0)
  print("LPVOcp.optimize (N=%d) p50 %.3f ms  p99 %.3f ms" % (NL_, 1e3 * np.median(lat), 1e3 * np.percentile(lat, 99)))
