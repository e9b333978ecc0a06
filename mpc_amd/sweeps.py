"""Tuning sweeps of the reference's plot_generator.py as batched runs: every point of a sweep is one scenario of a
single fused launch (per-scenario cost weights), instead of one rewritten parameters_*.yaml, one rebuilt controller and
one 300-step Python loop per point (plot_generator.py:92-515).  The step-response metrics and run-time statistics the
reference prints (plot_generator.py:1057-1072, 1160-1221) are restated on the returned arrays; plotting is not."""
from __future__ import annotations

import copy

import numpy as np
import pandas as pd

from .batched import StepBatch
from .controllers import LOcp, LPVOcp, read_parameters

dt = 0.01                                    # plot_generator.py:15-29
sim_len = 300
starting_state = [-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]

_CLASSES = {"L": LOcp, "LPV": LPVOcp}     # the models plot_generator sweeps on the state-feedback loop


# ------------------------------------------------------------------------------------------- metrics
def performance_metrics(y, i, dt=dt, sim_len=sim_len):
    """Rise time (first sample at 90% of the final value), settling time (last sample outside 2% of the final value)
    and overshoot of one response, exactly as plot_generator.py:1057-1072 computes them: the settling search runs over
    the reversed series and a result of 0 (the last sample itself outside the band) reads as None there too."""
    rise_time = next((t for t, val in enumerate(y) if val >= 0.9 * y[-1]), None) * dt
    settling_time = next((t for t, val in enumerate(y[::-1]) if abs(val - y[-1]) > 0.02 * y[-1]), None)
    settling_time = (sim_len - settling_time) * dt if settling_time else None
    overshoot = max(y) - y[-1]
    return {"State": f"x{i}", "Rise Time (s)": rise_time, "Settling Time (s)": settling_time, "Overshoot": overshoot}


def compute_performance_metrics(states: list, models: list = ["unknown"], dt=dt, sim_len=sim_len) -> pd.DataFrame:
    """plot_generator.py:1194-1221 (states numbered from x1 there); returns the table the reference prints."""
    results = []
    num_states = states[0].shape[1]
    for i in range(num_states):
        for model_idx, model in enumerate(models):
            m = performance_metrics(states[model_idx][:, i], i + 1, dt, sim_len)
            results.append({"Model": model, **m})
    return pd.DataFrame(results)


def compute_time_metrics(time_data_list: list) -> pd.DataFrame:
    """plot_generator.py:1160-1191: mean / median / std / min / max / 90th percentile per data set."""
    results = []
    for idx, time_data in enumerate(time_data_list):
        results.append({
            "Dataset": f"Dataset {idx + 1}",
            "Mean Runtime (ms)": np.mean(time_data),
            "Median Runtime (ms)": np.median(time_data),
            "Standard Deviation (ms)": np.std(time_data),
            "Minimum Runtime (ms)": np.min(time_data),
            "Maximum Runtime (ms)": np.max(time_data),
            "90th Percentile Runtime (ms)": np.percentile(time_data, 90),
        })
    return pd.DataFrame(results)


# ------------------------------------------------------------------------------------------- sweeps
def sweep_parameters(model: str, param: str, value, params: dict | None = None) -> dict:
    """The parameter set one sweep point runs with: what plot_generator.py writes into parameters_<model>.yaml before
    it rebuilds the controller (q: :109; q_y: :186-187; beta: :270; r: :432)."""
    base = read_parameters(_CLASSES[model].param_file, params)
    p = copy.deepcopy(base)
    c = p["controller"]
    if param == "q":
        c["q"] = value
    elif param == "q_y":
        c["q"] = 1
        c["Q"][1][1] = value
    elif param == "r":
        c["r"] = value
    elif param == "beta":
        c["beta"] = value
    else:
        raise ValueError("param must be 'q', 'q_y', 'r' or 'beta'")
    return p


def decade_values(low, high):
    """low * 10**idx for idx < log10(high // low), the grids of plot_generator.py:104-108, 181-185, 265-269, 424-431."""
    return [low * 10 ** idx for idx in range(int(np.log10(int(high // low))))]


def step_reference(states: np.ndarray, start_px: float) -> np.ndarray:
    """absolute_waypoints[0] of every step (simulator.py:186): the step planner's first waypoint is the car's own
    progress px before the step, with y = 0 left of the step and 1 from it on (step_planning.py:63-80)."""
    px = np.concatenate([[start_px], states[:-1, 0]])
    return np.stack([px, np.where(px < 0.0, 0.0, 1.0), np.ones_like(px), np.zeros_like(px)], axis=1)


class TuningSweep:
    """A sweep of one cost parameter over `values` for the step-reference experiment of plot_generator.py: all points
    start from the same state and run `n_steps` closed-loop steps in one launch."""

    def __init__(self, model: str, N: int, Tf: float, param: str, values, start=None, params: dict | None = None,
                 device=None, **ctl_kw):
        if model not in _CLASSES:
            raise ValueError("model must be 'L' or 'LPV'")
        self.model, self.N, self.Tf, self.param = model, int(N), float(Tf), param
        self.values = list(values)
        B = len(self.values)
        start = starting_state if start is None else start
        self.start = np.asarray(start, dtype=np.float64)[:8]
        w = np.zeros((B, 7)); We = np.zeros((B, 6, 6))
        self.controllers = []
        for b, v in enumerate(self.values):
            host = _CLASSES[model](N, Tf, params=sweep_parameters(model, param, v, params), _host_only=True)
            w[b] = np.diag(host.W); We[b] = np.asarray(host.W_e)[:6, :6]
            self.controllers.append(host)
        self.batch = StepBatch(N, Tf, np.tile(self.start, (B, 1)), model=model, device=device, fixed_model_cache=False,
                               params=params, **ctl_kw)
        self.batch.ocp.set_cost_weights(w, We)
        self.w, self.We = w, We

    def run(self, n_steps: int = sim_len):
        """Returns (states, inputs, reference, models): per point the (n, 8) states and (n, 1) inputs that
        StepSimulator.simulate returns, the (n, 4) reference of the last point (what the reference plots), and the
        labels plot_generator gives the points."""
        xs, us = self.batch.run(n_steps, record=True)
        xs = xs.cpu().numpy(); us = us.cpu().numpy()
        states = [np.ascontiguousarray(xs[:, b]) for b in range(len(self.values))]
        inputs = [np.ascontiguousarray(us[:, b:b + 1]) for b in range(len(self.values))]
        reference = step_reference(states[-1], self.start[0])
        models = [f"{self.model}-q:{v}" for v in self.values]
        return states, inputs, reference, models


def horizon_sweep(model: str, N_values, start=None, n_steps: int = sim_len, device=None, **ctl_kw):
    """plot_n_tuning (plot_generator.py:46-75): one controller per horizon N (Tf = dt * N); returns the state histories."""
    start = np.asarray(starting_state if start is None else start, dtype=np.float64)[:8]
    out = []
    for N in N_values:
        sim = StepBatch(int(N), dt * int(N), start[None, :], model=model, device=device, **ctl_kw)
        xs, us = sim.run(n_steps, record=True)
        out.append(xs.cpu().numpy()[:, 0])
    return out
