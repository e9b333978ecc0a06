"""Health check of the LPV workload: non-finite states and iteration-cap flags over a run, step by step."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from bench import build_workload
sim, run = build_workload("c3", 8192, 0, torch.device("cuda", 0), 120)
bad_total = 0
for t in range(0, 120, 10):
    run(10)
    st = sim.status.cpu().numpy(); x = sim.state.cpu().numpy()
    nonfinite = (~np.isfinite(x).all(axis=1)).sum()
    flagged = np.nonzero(st)[0]
    print("steps", t + 10, "status hist", np.bincount(st).tolist(), "non-finite scenarios", int(nonfinite),
          "max |y|", float(np.nanmax(np.abs(x[:, 1]))), "flagged sample", flagged[:5].tolist(),
          "their vx", x[flagged[:5], 4].round(2).tolist(), "their delta", x[flagged[:5], 7].round(3).tolist())
