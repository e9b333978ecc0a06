"""Pin the CPU checker (oracle/) against outputs recorded from the reference's own modules.

Fixtures under tests/golden/ were produced by oracle/make_golden.py, which imports
continuous_dynamics.Dynamics, EKF.CarEKF, SkidpadPlanner and StepPlanner from the
reference checkout; utils/waypoints.npy and terminal_set.m hold the reference's own
known answers.  Floating point: the checker re-associates nothing on purpose, but
libm/x87 differences allow a few ulp, so tolerances are 1e-12 relative unless noted.
"""
import numpy as np
import pytest

from oracle import oracle as O

MP_KEYS = dict(m=180.0, I_z=294.0, wbase=1.53, x_cg=0.57)


def _mp(orc):
    Cf, Cr = orc.tyre_stiffness(180.0, 0.57)
    return np.array([180.0, 294.0, 0.57 * 1.53, (1 - 0.57) * 1.53, Cf, Cr])


def test_tyre_stiffness_known_answer(orc64, golden):
    Cf, Cr = orc64.tyre_stiffness(180.0, 0.57)
    assert (Cf, Cr) == tuple(golden["plant"]["Cf_Cr"])           # bit-exact with np.interp
    assert Cf == pytest.approx(18879.96006089543, rel=0, abs=1e-9)
    assert Cr == pytest.approx(24292.066639693276, rel=0, abs=1e-9)


@pytest.mark.parametrize("nx", [8, 10])
def test_plant_model_and_rk4(orc64, orc80, golden, nx):
    g = golden["plant"]
    for orc, tol in ((orc64, 1e-13), (orc80, 1e-12)):
        mp = _mp(orc)
        for x, u, dt, f, x1 in zip(g[f"x{nx}"], g[f"u{nx}"], g[f"dt{nx}"], g[f"f{nx}"], g[f"x1_{nx}"]):
            np.testing.assert_allclose(orc.plant_f(x, u, mp), f, rtol=tol, atol=tol)
            np.testing.assert_allclose(orc.plant_rk4(x, u, dt, mp), x1, rtol=tol, atol=tol)


def test_plant_rk4_known_answer(orc64, golden):
    x1 = orc64.plant_rk4([15, 0.1, 1.0, 0, 8.0, 0, 0, 0], 0.5, 0.01, _mp(orc64))
    np.testing.assert_allclose(x1, golden["plant"]["kat_x1"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(x1, [15.08, 0.0999919183, 1.0, -4.53912824e-06, 8.0, -2.34488482e-03, -1.35045733e-03,
                                    5e-03], rtol=1e-8, atol=1e-12)


def test_ekf_jacobian(orc64, golden):
    g = golden["ekf"]
    mp = _mp(orc64)
    for x, F in zip(g["jac_x"], g["jac_F"]):
        np.testing.assert_allclose(orc64.ekf_jacobian(x, float(g["jac_dt"]), mp), F, rtol=1e-13, atol=1e-15)


def test_ekf_trace(orc64, golden):
    """200 predict/update steps driven like StepSimulator.lsim (simulator.py:334-376)."""
    g = golden["ekf"]
    mp = _mp(orc64)
    x, P = g["ekf_x0"].copy(), g["ekf_P0"].copy()
    dt = float(g["ekf_dt"])
    for i in range(len(g["ekf_u"])):
        x, P = orc64.ekf_time_update(x, P, g["ekf_u"][i], dt, mp, g["ekf_Q"])
        np.testing.assert_allclose(x, g["ekf_x_pred"][i], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(P, g["ekf_P_pred"][i], rtol=1e-8, atol=1e-9)
        x, P = orc64.ekf_meas_update(x, P, g["ekf_y"][i], g["ekf_R"])
        # P carries a 1e6 prior variance and is never re-symmetrised: compare relative to its scale
        np.testing.assert_allclose(x, g["ekf_x"][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(P, g["ekf_P"][i], rtol=1e-7, atol=1e-8)


def test_skidpad_planner_saved_waypoints(orc64, golden):
    """utils/waypoints.npy = SkidpadPlanner(9.0, Nt=500, dt=0.01).request_waypoints(18, 0, 0, 5)[0]."""
    wp, sp, p0, prev = orc64.skidpad_waypoints(18, 0, 0.0, 5, 0.0, 9.0, 500, 0.01)
    np.testing.assert_allclose(wp, golden["planners"]["saved_waypoints"], rtol=1e-13, atol=1e-13)
    assert np.all(sp == 9.0)


def test_skidpad_planner_full_track(orc64, golden):
    g = golden["planners"]
    N, dt, v = int(g["sk_N"]), float(g["sk_dt"]), float(g["sk_v"])
    laps_seen = set()
    for pose, lap, prev, wp_ref, p0_ref, prev_after in zip(g["sk_pose"], g["sk_lap"], g["sk_prev"], g["sk_wp"],
                                                           g["sk_p0"], g["sk_prev_after"]):
        wp, sp, p0, pa = orc64.skidpad_waypoints(pose[0], pose[1], pose[2], int(lap), float(prev), v, N, dt)
        np.testing.assert_allclose(wp, wp_ref, rtol=1e-12, atol=1e-12)
        assert p0 == pytest.approx(p0_ref, rel=1e-14, abs=1e-13)
        assert pa == pytest.approx(prev_after, rel=1e-14, abs=1e-13)
        laps_seen.add(int(lap))
    assert laps_seen >= {0, 1, 2, 3, 4, 5}          # all four track regions exercised


def _profile_cases(g):
    for ci in range(int(g["ncases"])):
        reqs = []
        for ri in range(int(g[f"c{ci}_nreq"])):
            reqs.append({k: g[f"c{ci}_r{ri}_{k}"] for k in ("pose", "lap", "N", "dt", "prev", "wp", "speeds", "p0", "prev_after")})
        yield float(g[f"c{ci}_v"]), g[f"c{ci}_zones"], g[f"c{ci}_qs"], g[f"c{ci}_sp"], g[f"c{ci}_ds"], reqs


def test_skidpad_planner_speed_profile(orc64, golden):
    """Laps outside 0..7 (utils/path_planning.py:241-248) and progress2speed (:93-126) against the reference's outputs;
    the zone edges recorded from the reference's planner objects pin planner_zones (:63-91) as well."""
    g = golden["planner_speed_profile"]
    params = ((9.0, 10.0, 0.5, None), (8.0, 10.0, 0.5, 3.0), (12.0, 6.0, 0.3, None), (5.0, 10.0, 0.7, None))
    branches = set()
    for (v, zones, qs, sp_ref, ds_ref, reqs), (pv, acc, fac, start) in zip(_profile_cases(g), params):
        assert v == pv
        np.testing.assert_array_equal(O.planner_zones(pv, acc, fac, start), zones)
        sp = np.array([orc64.progress2speed(zones, v, q) for q in qs])
        np.testing.assert_allclose(sp, sp_ref, rtol=1e-15, atol=1e-15)
        branches |= set(np.round(sp_ref, 6))
        for q in reqs:
            wp, spd, p0, pa = orc64.skidpad_waypoints_profile(q["pose"][0], q["pose"][1], q["pose"][2], int(q["lap"]),
                                                              float(q["prev"]), v, int(q["N"]), float(q["dt"]), zones)
            # the checker is compiled with fused multiply-adds; the ramps subtract nearby progresses (~250 m), so the
            # recurrence carries a few 1e-13 of that rounding difference
            np.testing.assert_allclose(spd, q["speeds"], rtol=2e-12, atol=2e-12)
            np.testing.assert_allclose(wp, q["wp"], rtol=1e-11, atol=1e-11)
            assert p0 == pytest.approx(float(q["p0"]), rel=1e-14, abs=1e-13)
            assert pa == pytest.approx(float(q["prev_after"]), rel=1e-14, abs=1e-13)
            assert spd[-1] == float(q["speeds"][-1])           # 0 on the profile branch, v when lap + 1 lands in 0..7
    assert {0.0, 0.1} <= branches and len(branches) > 20       # stop zone, off-track default and the ramps


def test_step_planner(orc64, golden):
    g = golden["planners"]
    N, dt, v = int(g["st_N"]), float(g["st_dt"]), float(g["st_v"])
    for pose, wp_ref, ab_ref in zip(g["st_pose"], g["st_wp"], g["st_abs"]):
        wp, sp, p0, ab = orc64.step_waypoints(pose[0], pose[1], pose[2], v, N, dt)
        np.testing.assert_allclose(wp, wp_ref, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(ab, ab_ref, rtol=1e-13, atol=1e-13)


def test_step_planner_ramped_step(orc64, golden):
    """StepPlanner(ramp_length > 0): the reference's simulators never set it, its planner supports it
    (step_planning.py:75-79); outputs recorded from the reference's planner."""
    g = golden["step_ramp"]
    on_ramp = 0
    for ci in range(int(g["ncases"])):
        N, dt, v, ramp = int(g[f"c{ci}_N"]), float(g[f"c{ci}_dt"]), float(g[f"c{ci}_v"]), float(g[f"c{ci}_ramp"])
        for pose, wp_ref, ab_ref, sp_ref, p0_ref in zip(g[f"c{ci}_pose"], g[f"c{ci}_wp"], g[f"c{ci}_abs"],
                                                        g[f"c{ci}_speeds"], g[f"c{ci}_p0"]):
            wp, sp, p0, ab = orc64.step_waypoints(pose[0], pose[1], pose[2], v, N, dt, ramp)
            np.testing.assert_allclose(wp, wp_ref, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(ab, ab_ref, rtol=1e-13, atol=1e-13)
            np.testing.assert_array_equal(sp, sp_ref)
            assert p0 == p0_ref
            on_ramp += int(np.sum((ab_ref[:, 1] > 0) & (ab_ref[:, 1] < 1)))
    assert on_ramp > 100


def test_linear_model_known_answer_terminal_set(orc64, golden):
    """terminal_set.m:4-10 prints the float32 forward-Euler (A,B) at vx=15, dt=0.02 without the px row."""
    g = golden["kat_terminal_set"]
    mp = _mp(orc64)
    A = (np.eye(6) + float(g["dt"]) * O.l_model_continuous(mp, float(g["vx"]))).astype(np.float32)[1:, 1:]
    np.testing.assert_allclose(A, g["A"], rtol=2e-7, atol=5e-9)        # 8 printed digits
    B = np.zeros(6); B[5] = float(g["dt"])
    np.testing.assert_allclose(B[1:], g["B"], rtol=0, atol=0)


def test_dare_terminal_weight(orc64):
    """DARE with R'=0 on float32 data (LMPC2.py:349-440): check the Riccati equation itself."""
    mp = _mp(orc64)
    W = np.diag([0, 1e5, 1, 0, 0, 1.0, 10.0])
    for dt in (0.01, 0.02, 0.05):
        P, K, A, B = O.dare_terminal(mp, dt, W)
        A = A.astype(float); B = B.astype(float)
        Q = np.diag([1e5, 1, 0, 0, 0.0])
        res = A.T @ P @ A - P - (A.T @ P @ B) @ np.linalg.solve(B.T @ P @ B, B.T @ P @ A) + Q
        assert np.abs(res).max() <= 1e-7 * np.abs(P).max()
        assert np.all(np.abs(np.linalg.eigvals(A - B @ K)) < 1.0)
    P, _, _, _ = O.dare_terminal(mp, 0.01, W)
    # order of magnitude recorded when the reference's formulas were first evaluated (dt = 0.01)
    assert 2.5e5 < P[0, 0] < 3.5e5 and 1.0e4 < P[1, 1] < 1.3e4
