"""Iteration statistics of the LPV workload, step by step: histogram over scenarios and the mean over wavefronts of the
largest count among a wavefront's lanes (what a lane-per-scenario wavefront has to run)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from bench import build_workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sim, run = build_workload("c3", B, 0, torch.device("cuda", 0), 400)
run(10)
prev = sim.iters.cpu().numpy().copy()
tot = np.zeros(64, int); wmax = {16: [], 32: [], 64: []}; means = []
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
step_max = []
for t in range(steps):
    run(1)
    it = sim.iters.cpu().numpy()
    d = it - prev; prev = it.copy()
    if d.min() < 0: d = it          # counters restart with every launch
    tot += np.bincount(d, minlength=64)[:64]
    means.append(d.mean()); step_max.append(int(d.max()))
    for l in wmax:
        wmax[l].append(d.reshape(-1, l).max(axis=1).mean())
print("mean iterations per scenario-step", np.mean(means))
print("histogram", {i: int(v) for i, v in enumerate(tot) if v})
for l in wmax:
    print("lanes per wave", l, "mean over waves of max over lanes", np.mean(wmax[l]), " worst wave", np.max(wmax[l]))
print("largest count in a step, per step:", step_max)
