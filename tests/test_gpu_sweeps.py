"""Tuning sweeps as one launch (mpc_amd/sweeps.py) against the same points run one controller at a time, and against the
checker's loop with the modified parameter set (plot_generator.py:168-515: rewrite parameters_*.yaml, rebuild, simulate)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _utol(u_ref, umax=0.9):
    return 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)


@pytest.mark.parametrize("model,N,param,values", [("L", 50, "beta", [1, 10, 100, 1000]),
                                                  ("LPV", 80, "q_y", [0.1, 1.0, 10.0, 100.0, 1000.0]),
                                                  ("L", 80, "r", [0.01, 1.0, 100.0])])
def test_sweep_equals_one_controller_per_point(orc64, model, N, param, values):
    import torch
    from oracle import oracle as O
    from mpc_amd import sweeps
    from mpc_amd.batched import StepBatch
    n = 60
    Tf = sweeps.dt * N
    sw = sweeps.TuningSweep(model, N, Tf, param, values)
    states, inputs, reference, models = sw.run(n)
    assert len(states) == len(values) and states[0].shape == (n, 8) and inputs[0].shape == (n, 1) and reference.shape == (n, 4)
    start = np.asarray(sweeps.starting_state)[:8]
    for b, v in enumerate(values):
        p = sweeps.sweep_parameters(model, param, v)
        one = StepBatch(N, Tf, start[None, :], model=model, fixed_model_cache=False, params=p)
        xs, us = one.run(n, record=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(xs.cpu().numpy()[:, 0], states[b])        # same arithmetic, weights from a table
        np.testing.assert_array_equal(us.cpu().numpy()[:, 0], inputs[b][:, 0])
    # distinct points give distinct responses
    assert np.abs(states[0][:, 1] - states[-1][:, 1]).max() > 1e-3
    # one point against the checker's loop with that parameter set
    b = len(values) // 2
    cfg = O.make_cfg(model, N, Tf, params=sweeps.sweep_parameters(model, param, values[b]))
    x = start.copy(); lx = np.zeros((6, N + 1)); lu = np.zeros(N)
    for i in range(25):
        wp, sp, p0, ab = orc64.step_waypoints(x[0], x[1], np.arctan2(x[3], x[2]), 15.0, N, Tf / N)
        st, traj, U, lam, info = orc64.optimize(cfg, x[[0, 1, 2, 3, 5, 6, 7]], wp, sp, 0.0, lx, lu)
        assert abs(inputs[b][i, 0] - U[0]) <= 10 * _utol(np.array([U[0]]))[0], (i, inputs[b][i, 0], U[0])
        np.testing.assert_allclose(reference[i], ab[0], rtol=0, atol=1e-6) if b == len(values) - 1 else None
        x = orc64.plant_rk4(x, U[0], Tf / N, cfg[3:9])
        np.testing.assert_allclose(states[b][i], x, rtol=0, atol=1e-7)
    df = sweeps.compute_performance_metrics(states, models, sim_len=n)
    assert len(df) == 8 * len(values)


def test_horizon_sweep_runs_every_horizon():
    from mpc_amd import sweeps
    out = sweeps.horizon_sweep("L", [50, 54, 58], n_steps=30)
    assert len(out) == 3 and all(o.shape == (30, 8) and np.isfinite(o).all() for o in out)
    assert np.abs(out[0] - out[2]).max() > 1e-6
