"""Cost and constraint-violation statistics of the closed loops (mpc_get_stats slots 16..23) against the checker's loop:
the stage cost of the applied steps, the tracking error, the heading pair's violation (skidpad_simulator.py:134) and the
bound excess, summed / maximised over 32 scenarios x 100 steps for every kernel that runs a closed loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B, STEPS = 32, 100


def _starts(seed, n=B, vx=8.0):
    rng = np.random.default_rng(seed)
    psi = rng.uniform(-0.1, 0.1, n)
    return np.stack([np.full(n, 15.0), rng.uniform(-0.3, 0.3, n), np.cos(psi), np.sin(psi), np.full(n, vx),
                     rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), rng.uniform(-0.05, 0.05, n)], axis=1)


def _check(st, cl, steps=STEPS, cost_rtol=1e-6, pred=False):
    """st: the library's reduced statistics; cl: the checker's (B, 8) per-scenario statistics."""
    n = cl.shape[0]
    assert st["closed_loop_steps"] == n * steps == cl[:, 7].sum()
    # the closed loops of library and checker agree to ~1e-8 in the states, so the sums agree far below 1e-6 relative
    assert st["sum_stage_cost"] == pytest.approx(cl[:, 0].sum(), rel=cost_rtol)
    assert st["sum_lateral_error_sq"] == pytest.approx(cl[:, 1].sum(), rel=cost_rtol)
    assert st["sum_input_sq"] == pytest.approx(cl[:, 3].sum(), rel=1e-5)
    # | |(cos, sin)| - 1 | is a few ulp of 1: double precision evaluates it to +-2.2e-16 per step
    assert st["sum_heading_pair_violation"] == pytest.approx(cl[:, 2].sum(), abs=3e-16 * n * steps)
    # (where the plant's integration error shows in the pair, ~1e-9 at 15 m/s, it follows the states' agreement instead)
    assert st["max_heading_pair_violation"] == pytest.approx(cl[:, 5].max(), abs=5e-16, rel=1e-6)
    assert st["max_bound_violation"] == pytest.approx(max(cl[:, 4].max(), 0.0), abs=1e-9)
    if pred:
        assert st["max_predicted_heading_pair_violation"] == pytest.approx(cl[:, 6].max(), abs=1e-12)
    else:
        assert st["max_predicted_heading_pair_violation"] == 0.0


@pytest.mark.parametrize("model,N,Tf,kw", [
    ("L", 50, 0.5, {}),                                    # fixed-model cache, register kernel (work queue)
    ("L", 50, 0.5, {"fixed_model_cache": False}),          # per-scenario condensing, register kernel
    ("L", 25, 0.5, {}),                                    # any-horizon kernel
    ("LPV", 80, 0.8, {"solver_layout": "lane"}),           # lane per scenario
    ("LPV", 50, 0.5, {}),
    ("NL", 50, 0.5, {}),
])
def test_skidpad_run_statistics_match_checker(orc80, model, N, Tf, kw):
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    st0 = _starts(11)
    sim = SkidpadBatch(N, Tf, st0, model=model, **kw)
    sim.run(STEPS)
    got = sim.ocp.stats()
    cfg = O.make_cfg(model, N, Tf)
    ref = st0.copy(); laps = np.zeros(B, np.int32); pp = np.zeros(B); cl = np.zeros((B, 8))
    orc80.skidpad_run(cfg, ref, laps, pp, 8.0, STEPS, cl_stats=cl)
    np.testing.assert_allclose(sim.state.cpu().numpy(), ref, rtol=0, atol=1e-6)      # same closed loops
    _check(got, cl, pred=model == "NL")
    assert got["sum_stage_cost"] > 0 and got["sum_lateral_error_sq"] > 0
    # one more launch: the statistics describe the last call only
    sim.run(3)
    assert sim.ocp.stats()["closed_loop_steps"] == 3 * B


def test_step_and_single_step_launches(orc80):
    """mpc_closed_loop_step (one step per call) carries the same statistics as the first step of the checker's loop, and an
    optimize() call in between reports none."""
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    st0 = _starts(12)
    sim = SkidpadBatch(50, 0.5, st0, model="L")
    sim.step()
    got = sim.ocp.stats()
    cfg = O.make_cfg("L", 50, 0.5)
    ref = st0.copy(); laps = np.zeros(B, np.int32); pp = np.zeros(B); cl = np.zeros((B, 8))
    orc80.skidpad_run(cfg, ref, laps, pp, 8.0, 1, cl_stats=cl)
    _check(got, cl, steps=1)


def test_relaxed_start_reports_a_bound_violation(orc80):
    """A start steering angle outside its bound (status 4: the bound is widened on the way back) shows up as a positive
    bound excess, the same as the checker's."""
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    st0 = _starts(13, n=8)
    st0[:, 7] = np.linspace(0.305, 0.33, 8)
    sim = SkidpadBatch(50, 0.5, st0, model="L")
    sim.run(10)
    got = sim.ocp.stats()
    cfg = O.make_cfg("L", 50, 0.5)
    ref = st0.copy(); laps = np.zeros(8, np.int32); pp = np.zeros(8); cl = np.zeros((8, 8))
    orc80.skidpad_run(cfg, ref, laps, pp, 8.0, 10, cl_stats=cl)
    assert cl[:, 4].max() > 0.01
    assert got["max_bound_violation"] == pytest.approx(cl[:, 4].max(), abs=1e-8)


def test_output_feedback_statistics_match_checker(orc64):
    from oracle import oracle as O
    from mpc_amd.batched import StepOFLBatch
    N, Tf, n, nb = 50, 0.5, 60, 8
    rng = np.random.default_rng(5)
    true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (nb, 1))
    true0[:, 9] = rng.uniform(-12, -8, nb)
    est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (nb, 1))
    sigma = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])
    noise = 0.05 * sigma * rng.standard_normal((n, nb, 7))
    sim = StepOFLBatch(N, Tf, true0, est0)
    sim.run(n, meas_noise=noise)
    got = sim.ocp.stats()
    cfg = O.make_cfg("OFL", N, Tf)
    x = true0.copy(); e = est0.copy(); P = np.tile(np.diag(O.EKF_P0_DIAG), (nb, 1, 1)); cl = np.zeros((nb, 8))
    orc64.ofl_run(cfg, x, e, P, 15.0, n, noise=noise, cl_stats=cl)
    np.testing.assert_allclose(sim.state.cpu().numpy(), x, rtol=0, atol=1e-6)
    _check(got, cl, steps=n, cost_rtol=1e-5)


def test_chained_launches_accumulate():
    """A run long enough to go out as a chain of launches (1024 steps per launch from 1024 scenarios up) reports the
    statistics of the whole run: the same sums as two shorter runs of the same scenarios added up."""
    from mpc_amd.batched import SkidpadBatch
    st0 = _starts(14, n=1024)
    a = SkidpadBatch(50, 0.5, st0, model="L")
    a.run(1100)
    sa = a.ocp.stats()
    b = SkidpadBatch(50, 0.5, st0, model="L")
    b.run(600); s1 = b.ocp.stats()
    b.run(500); s2 = b.ocp.stats()
    assert sa["closed_loop_steps"] == 1100 * 1024
    for k in ("sum_stage_cost", "sum_lateral_error_sq", "sum_input_sq", "sum_heading_pair_violation"):
        assert sa[k] == pytest.approx(s1[k] + s2[k], rel=1e-12), k
    for k in ("max_bound_violation", "max_heading_pair_violation"):
        assert sa[k] == max(s1[k], s2[k]), k
