"""The active-set solve of the fixed-model path (mpc_amd/csrc/mpc_aset.hpp) against the 80-bit checker and against the
interior point.  Both GPU methods and the checker's stage-wise Riccati interior point are three different algorithms for
the unique minimiser of the problem LMPC2.py:196-260 poses; the bar is the per-step input bar of the other parity tests,
1e-5 relative with a floor of 1e-3 of the rate bound (parity unpinned, as for every QP solve: the reference's own solver is
not available)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tol(u_ref, umax):
    return 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)


def _starts(B, seed):
    rng = np.random.default_rng(seed)
    psi = rng.uniform(-0.1, 0.1, B)
    return np.stack([np.full(B, 15.0), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                     rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.05, 0.05, B)], axis=1)


@pytest.mark.parametrize("N,Tf,nsteps", [(50, 0.5, 90), (10, 0.1, 40), (20, 0.2, 40), (30, 0.3, 40), (40, 0.4, 40), (64, 0.64, 60)])
def test_every_closed_loop_step_matches_checker(orc80, N, Tf, nsteps):
    """Every step of every sampled scenario is solved again by the checker from the state the GPU run recorded (so the
    comparison is per step, with no feedback of differences), through the start transient (sets of up to ~2N - 1 bounds,
    all three size classes, hand-overs to the interior point) and the steps behind it (a few bounds)."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    B = 96
    st = _starts(B, 11)
    cfg = O.make_cfg("L", N, Tf)
    umax = cfg[10]
    sim = SkidpadBatch(N, Tf, st, model="L")
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    xs = xs.cpu().numpy(); us = us.cpu().numpy()
    assert (sim.status.cpu().numpy() == 0).all()
    stats = sim.ocp.stats()
    assert stats["n_active_set_solves"] >= 0.5 * stats["n_constrained"] > 0          # the method under test did the work
    for b in range(0, B, 4):
        lap = 0
        for t in range(nsteps):
            x = st[b] if t == 0 else xs[t - 1, b]
            _, _, _, _, u_ref, _ = orc80.skidpad_step(cfg, x.copy(), lap, 0.0, 8.0)
            assert abs(us[t, b] - u_ref) <= _tol(u_ref, umax), (N, b, t, us[t, b], u_ref)
            if xs[t, b, 0] > 16.75 and x[0] <= 16.75:                                  # skidpad_simulator.py:91-93
                lap += 1


def test_active_set_and_interior_point_agree_on_the_skidpad():
    """The same 512 closed loops with both methods: inputs within 10x the bar (differences feed back), states 1e-7."""
    import torch
    from mpc_amd.batched import SkidpadBatch
    st = _starts(512, 5)
    runs = {}
    for method in ("auto", "interior_point"):
        sim = SkidpadBatch(50, 0.5, st, model="L", qp_method=method)
        xs, us = sim.run(120, record=True)
        torch.cuda.synchronize()
        assert (sim.status.cpu().numpy() == 0).all()
        runs[method] = (xs.cpu().numpy(), us.cpu().numpy(), sim.ocp.stats())
    assert runs["interior_point"][2]["n_active_set_solves"] == 0
    assert runs["auto"][2]["n_active_set_solves"] > 0
    assert runs["auto"][2]["sum_iterations"] < 0.5 * runs["interior_point"][2]["sum_iterations"]
    ua, ub = runs["auto"][1], runs["interior_point"][1]
    assert (np.abs(ua - ub) <= 10 * _tol(ub, 0.9)).all(), np.abs(ua - ub).max()
    assert np.abs(runs["auto"][0] - runs["interior_point"][0]).max() <= 1e-7


def test_output_feedback_loop_methods(orc64):
    """The fused output-feedback loop (simulator.py:206-281).  On the way to the step every input of its problems is fixed
    by a bound (full rate until the steering angle saturates, then the angle bound to the end of the horizon: as many
    active bounds as inputs); a run of active steering bounds enters the multiplier system as unit rows, which keeps that
    system as well conditioned as inv(H).  Without noise nearly every solve ends in the active-set method ("auto", and
    "active_set", which is the same thing), "interior_point" switches it off, and all follow the checker's loop."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import StepOFLBatch
    B, n = 64, 60
    rng = np.random.default_rng(2)
    x = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, -10.0, 0.0]), (B, 1))
    x[:, 1] = rng.uniform(-0.2, 0.2, B)
    cfg = O.make_cfg("OFL", 50, 0.5)
    est = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
    P = np.tile(np.diag(O.EKF_P0_DIAG), (B, 1, 1))
    xr = x.copy()
    u_ref, x_ref, e_ref, st_ref, _ = orc64.ofl_run(cfg, xr, est, P, 15.0, n)
    out = {}
    for method in ("auto", "interior_point", "active_set"):
        sim = StepOFLBatch(50, 0.5, x, qp_method=method)
        xs, us, es = sim.run(n, record=True)
        torch.cuda.synchronize()
        assert (sim.status.cpu().numpy() == st_ref).all()
        out[method] = (xs.cpu().numpy(), us.cpu().numpy(), sim.ocp.stats())
        assert (np.abs(out[method][1] - u_ref) <= 10 * _tol(u_ref, cfg[10])).all(), (method, np.abs(out[method][1] - u_ref).max())
        assert np.abs(out[method][0] - x_ref).max() <= 1e-7, method
    assert np.array_equal(out["auto"][1], out["active_set"][1])                  # the same arithmetic
    sa = out["active_set"][2]
    assert out["interior_point"][2]["n_active_set_solves"] == 0 and sa["n_active_set_solves"] >= 0.8 * sa["n_constrained"] > 0
    assert sa["sum_iterations"] < 0.25 * out["interior_point"][2]["sum_iterations"]


def test_fused_launch_equals_per_step_launches_bit_for_bit():
    """The set a solve starts from travels in the scenario's record in HBM between launches exactly as inside one."""
    import torch
    from mpc_amd.batched import SkidpadBatch
    st = _starts(300, 9)
    a = SkidpadBatch(50, 0.5, st, model="L")
    b = SkidpadBatch(50, 0.5, st, model="L")
    xa, ua = a.run(70, record=True)
    ub = []
    for _ in range(70):
        _, u = b.run(1, record=True)
        ub.append(u)
    torch.cuda.synchronize()
    assert torch.equal(ua, torch.cat(ub, 0))
    assert torch.equal(a.state, b.state)
    sa, sb = a.ocp.stats(), b.ocp.stats()
    assert sa["n_active_set_solves"] > 0


@pytest.mark.parametrize("scale_q,scale_r", [(100.0, 0.01), (0.01, 100.0), (1e4, 1e-3)])
def test_stiff_weights_at_the_longest_horizon(orc80, scale_q, scale_r):
    """Weights far from the parameter file's (state weights q and input weight r moved apart or together by 1e4 .. 1e7) at
    N = 64: the table inv(H) the method certifies its points through is checked against H when it is built
    (mpc_get_stats: max_table_residual); where it is not accurate enough the method stays off.  Either way every step
    matches the 80-bit checker."""
    import copy
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    N, Tf, nsteps, B = 64, 0.64, 40, 48
    params = copy.deepcopy(O.load_params("L"))
    params["controller"]["q"] *= scale_q
    params["controller"]["r"] *= scale_r
    cfg = O.make_cfg("L", N, Tf, params=params)
    umax = cfg[10]
    st = _starts(B, 21)
    sim = SkidpadBatch(N, Tf, st, model="L", params=params)
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    xs = xs.cpu().numpy(); us = us.cpu().numpy()
    assert (sim.status.cpu().numpy() == 0).all()
    stats = sim.ocp.stats()
    off = stats["max_table_residual"] > 1e-9
    assert stats["n_active_set_switched_off"] == float(off)
    assert off == (scale_q / scale_r >= 1e4)           # cond(H) 1.5e9 and 3e10 against 1.6e5 with the file's weights
    if off:
        assert stats["n_active_set_solves"] == 0 and stats["n_constrained"] > 0
    elif stats["n_constrained"] > 0:
        assert stats["n_active_set_solves"] > 0
    for b in range(0, B, 4):
        lap = 0
        for t in range(nsteps):
            x = st[b] if t == 0 else xs[t - 1, b]
            _, _, _, _, u_ref, _ = orc80.skidpad_step(cfg, x.copy(), lap, 0.0, 8.0)
            assert abs(us[t, b] - u_ref) <= _tol(u_ref, umax), (scale_q, scale_r, b, t, us[t, b], u_ref, stats["max_table_residual"])
            if xs[t, b, 0] > 16.75 and x[0] <= 16.75:
                lap += 1
