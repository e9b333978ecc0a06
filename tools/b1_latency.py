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
