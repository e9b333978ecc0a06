"""GPU parity of the pieces called through the reference's class surface (plant, EKF, planners, simulators):
HIP path through the C ABI vs the golden vectors recorded from the reference's own modules and vs the CPU checker.
Floating point: tolerances are stated per test; plant/planner agree to a few ulp, the EKF covariance (1e6 prior,
never re-symmetrised) to 1e-7 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_plant_matches_reference_vectors(golden):
    from mpc_amd.continuous_dynamics import Dynamics
    g = golden["plant"]
    for nx, dist in ((8, False), (10, True)):
        for dt in np.unique(g[f"dt{nx}"]):
            dyn = Dynamics(float(dt), disturbance=dist)
            sel = g[f"dt{nx}"] == dt
            x, u = g[f"x{nx}"][sel], g[f"u{nx}"][sel]
            np.testing.assert_allclose(dyn.model_batched(x, u).cpu().numpy(), g[f"f{nx}"][sel], rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(dyn.rk4_batched(x, u).cpu().numpy(), g[f"x1_{nx}"][sel], rtol=1e-13, atol=1e-13)
    dyn = Dynamics(0.01)
    assert (dyn.Cf, dyn.Cr) == tuple(g["Cf_Cr"])
    x1 = dyn.rk4_integraton(np.array([15, 0.1, 1.0, 0, 8.0, 0, 0, 0]), 0.5)
    np.testing.assert_allclose(x1, g["kat_x1"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(dyn.single_track_model(g["x8"][0], g["u8"][0]), g["f8"][0], rtol=1e-13, atol=1e-13)


def test_ekf_jacobian_and_trace_match_reference_vectors(golden):
    from mpc_amd.continuous_dynamics import Dynamics
    from mpc_amd.EKF import CarEKF
    g = golden["ekf"]
    dyn = Dynamics(float(g["jac_dt"]), disturbance=True)
    for x, F in zip(g["jac_x"][:6], g["jac_F"][:6]):
        A, B, Fg = dyn.jacobian_forward_euler(x)
        np.testing.assert_allclose(Fg, F, rtol=1e-13, atol=1e-15)
        assert B[7] == 1 and B.sum() == 1
    ekf = CarEKF(float(g["ekf_dt"]), True, inital_state=g["ekf_x0"])
    np.testing.assert_array_equal(ekf.P, g["ekf_P0"])
    np.testing.assert_array_equal(np.diag(ekf.Q), g["ekf_Q"]); np.testing.assert_array_equal(np.diag(ekf.R), g["ekf_R"])
    for i in range(len(g["ekf_u"])):
        ekf.time_update(g["ekf_u"][i])
        if i % 20 == 0:
            np.testing.assert_allclose(ekf.x_est, g["ekf_x_pred"][i], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(ekf.P, g["ekf_P_pred"][i], rtol=1e-7, atol=1e-8)
        if i == 3:
            K = ekf.kalman_gain()          # K = P H' inv(H P H' + R)
            P = ekf.P; H = ekf.dynamics.measurement_matrix
            np.testing.assert_allclose(K @ (H @ P @ H.T + ekf.R), P @ H.T, rtol=1e-9, atol=1e-9)
        ekf.measurement_update(g["ekf_y"][i])
        if i % 20 == 0 or i == len(g["ekf_u"]) - 1:
            np.testing.assert_allclose(ekf.x_est, g["ekf_x"][i], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(ekf.P, g["ekf_P"][i], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(ekf.estimated_red_state(), g["ekf_red"], rtol=1e-8, atol=1e-9)


def test_planners_match_reference_vectors(golden):
    import torch
    from mpc_amd.utils.path_planning import SkidpadPlanner
    from mpc_amd.utils.step_planning import StepPlanner
    g = golden["planners"]
    pl = SkidpadPlanner(target_vel=9.0, Nt=500, dt=0.01)
    wp, sp, p0, R = pl.request_waypoints(18, 0, 0, 5)          # the call that made utils/waypoints.npy
    np.testing.assert_allclose(wp, g["saved_waypoints"], rtol=1e-12, atol=1e-12)
    assert wp.shape == (501, 4) and np.all(sp == 9.0) and R.shape == (2, 2)
    # full figure-8 drive, batched: one request per recorded pose
    N, dt, v = int(g["sk_N"]), float(g["sk_dt"]), float(g["sk_v"])
    pl = SkidpadPlanner(target_vel=v, Nt=N, dt=dt)
    dev = pl.device
    pose = torch.as_tensor(g["sk_pose"], device=dev)
    lap = torch.as_tensor(g["sk_lap"].astype(np.int32), device=dev)
    prev = torch.as_tensor(g["sk_prev"].astype(np.float64), device=dev).clone()
    tv = torch.full((pose.shape[0],), v, dtype=torch.float64, device=dev)
    wps, prog = pl.request_waypoints_batched(pose, lap, prev, tv)
    np.testing.assert_allclose(wps.cpu().numpy(), g["sk_wp"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(prog.cpu().numpy(), g["sk_p0"], rtol=1e-14, atol=1e-13)
    np.testing.assert_allclose(prev.cpu().numpy(), g["sk_prev_after"], rtol=1e-14, atol=1e-13)
    # stateful single-request surface: prev_progress carried like the reference object
    pl2 = SkidpadPlanner(target_vel=v, Nt=N, dt=dt)
    for i in range(0, 40):
        w, _, p0, _ = pl2.request_waypoints(*g["sk_pose"][i], int(g["sk_lap"][i]))
        np.testing.assert_allclose(w, g["sk_wp"][i], rtol=1e-11, atol=1e-11)
        assert pl2.prev_progress == pytest.approx(g["sk_prev_after"][i], rel=1e-14, abs=1e-13)
    N, dt, v = int(g["st_N"]), float(g["st_dt"]), float(g["st_v"])
    spn = StepPlanner(target_vel=v, Nt=N, dt=dt)
    for pose, wp_ref, ab_ref in zip(g["st_pose"][:10], g["st_wp"][:10], g["st_abs"][:10]):
        w, speeds, p0, R, ab = spn.request_waypoints(*pose)
        np.testing.assert_allclose(w, wp_ref, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(ab, ab_ref, rtol=1e-13, atol=1e-13)
        assert p0 == pose[0] and np.all(speeds == v)


def test_step_planner_ramped_step(golden):
    """StepPlanner(ramp_length > 0) (utils/step_planning.py:75-79), batched, against the reference's recorded outputs."""
    import torch
    from mpc_amd.utils.step_planning import StepPlanner
    g = golden["step_ramp"]
    for ci in range(int(g["ncases"])):
        N, dt, v, ramp = int(g[f"c{ci}_N"]), float(g[f"c{ci}_dt"]), float(g[f"c{ci}_v"]), float(g[f"c{ci}_ramp"])
        spn = StepPlanner(target_vel=v, Nt=N, dt=dt, ramp_length=ramp)
        pose = torch.as_tensor(g[f"c{ci}_pose"], device=spn.device)
        tv = torch.full((pose.shape[0],), v, dtype=torch.float64, device=spn.device)
        wp, prog, ab = spn.request_waypoints_batched(pose, tv)
        np.testing.assert_allclose(wp.cpu().numpy(), g[f"c{ci}_wp"], rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal(ab.cpu().numpy(), g[f"c{ci}_abs"])
        np.testing.assert_array_equal(prog.cpu().numpy(), g[f"c{ci}_p0"])
        w, speeds, p0, R, ab1 = spn.request_waypoints(*g[f"c{ci}_pose"][3])
        np.testing.assert_allclose(w, g[f"c{ci}_wp"][3], rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal(speeds, g[f"c{ci}_speeds"][3])
    with pytest.raises(ValueError):
        StepPlanner(target_vel=8.0, ramp_length=-1.0)


@pytest.mark.parametrize("model,N,cache", [("L", 50, True), ("L", 50, False), ("LPV", 80, False), ("NL", 20, False)])
def test_closed_loop_on_ramped_step_matches_checker(orc64, model, N, cache):
    """The fused step-reference loop with StepPlanner's ramp_length (every kernel's planner stage) against the checker's
    loop: plan on the ramp -> solve -> RK4 plant."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import StepBatch
    ramp, nsteps, v = 3.0, 40, 8.0
    dt = 0.01
    Tf = dt * N
    hd = np.arctan2(1.0, ramp)                                             # second scenario: already driving up the ramp
    start = np.array([[-1.0, 0.0, 1.0, 0.0, v, 0.0, 0.0, 0.0], [0.9, 0.3, np.cos(hd), np.sin(hd), v, 0.0, 0.0, 0.0],
                      [-4.5, -0.05, 1.0, 0.0, v, 0.05, 0.0, -0.02]])
    kw = dict(sqp_tol=1e-10, sqp_max_iter=50) if model == "NL" else {}
    a = StepBatch(N, Tf, start, model=model, fixed_model_cache=cache, step_ramp_length=ramp, **kw)
    xs, us = a.run(nsteps, record=True)
    plain = StepBatch(N, Tf, start, model=model, fixed_model_cache=cache, **kw)
    xs0, us0 = plain.run(nsteps, record=True)
    torch.cuda.synchronize()
    xs, us = xs.cpu().numpy(), us.cpu().numpy()
    assert np.abs(us - us0.cpu().numpy()).max() > 1e-3                      # the ramp changes the response
    cfg = O.make_cfg(model, N, Tf, **({"sqp_tol": 1e-11, "sqp_max_iter": 50} if model == "NL" else {}))
    for b in range(start.shape[0]):
        x = start[b].copy(); lx = np.zeros((6, N + 1)); lu = np.zeros(N); warm = None
        for i in range(nsteps):
            wp, sp, p0, ab = orc64.step_waypoints(x[0], x[1], np.arctan2(x[3], x[2]), v, N, dt, ramp)
            if model == "NL":
                st, traj, U, lam, info = orc64.nl_optimize(cfg, x[[0, 1, 2, 3, 5, 6, 7]], wp, sp, warm)
                warm = U
            else:
                st, traj, U, lam, info = orc64.optimize(cfg, x[[0, 1, 2, 3, 5, 6, 7]], wp, sp, 0.0, lx, lu)
            tol = 10 * 1e-5 * max(abs(U[0]), 1e-3 * 0.9)
            assert abs(us[i, b] - U[0]) <= tol, (model, b, i, us[i, b], U[0])
            x = orc64.plant_rk4(x, U[0], dt, cfg[3:9])
            np.testing.assert_allclose(xs[i, b], x, rtol=0, atol=1e-7)


def test_skidpad_planner_speed_profile(orc64, golden):
    """Laps outside 0..7: the speed profile of utils/path_planning.py:241-248 through the class surface and batched,
    against the reference's recorded outputs and the CPU restatement."""
    import torch
    from mpc_amd.utils.path_planning import SkidpadPlanner
    g = golden["planner_speed_profile"]
    params = ((9.0, 10.0, 0.5, None), (8.0, 10.0, 0.5, 3.0), (12.0, 6.0, 0.3, None), (5.0, 10.0, 0.7, None))
    for ci, (v, acc, fac, start) in enumerate(params):
        dt = 0.05 if fac == 0.5 else 0.1
        pl = SkidpadPlanner(target_vel=v, max_accel=acc, Nt=40, dt=dt, slowdown_speed_factor=fac)
        if start is not None:
            pl.set_starting_position(start)
        np.testing.assert_array_equal(pl.zones(), g[f"c{ci}_zones"])
        np.testing.assert_allclose([pl.progress2speed(q) for q in g[f"c{ci}_qs"]], g[f"c{ci}_sp"], rtol=1e-15, atol=1e-15)
        np.testing.assert_array_equal([pl.request_desired_speed(q) for q in g[f"c{ci}_qs"]], g[f"c{ci}_ds"])
        nreq = int(g[f"c{ci}_nreq"])
        for ri in range(nreq):
            q = {k: g[f"c{ci}_r{ri}_{k}"] for k in ("pose", "lap", "N", "prev", "wp", "speeds", "p0", "prev_after")}
            assert pl.prev_progress == pytest.approx(float(q["prev"]), rel=1e-14, abs=1e-13)
            w, spd, p0, R = pl.request_waypoints(*q["pose"], int(q["lap"]), Nt=int(q["N"]))
            np.testing.assert_allclose(spd, q["speeds"], rtol=1e-14, atol=1e-14)
            np.testing.assert_allclose(w, q["wp"], rtol=1e-11, atol=1e-11)
            assert p0 == pytest.approx(float(q["p0"]), rel=1e-14, abs=1e-13)
    # batched: mixed laps (constant-speed and profile scenarios in one launch), per-scenario zones and speeds
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    B, N, dt = 64, 70, 0.04
    pl = SkidpadPlanner(target_vel=9.0, Nt=N, dt=dt)
    dev = pl.device
    lap = rng.choice([0, 5, 7, 8, 9, -1], B).astype(np.int32)
    pose = np.stack([rng.uniform(-1, 50, B), rng.uniform(-0.4, 0.4, B), rng.uniform(-0.2, 0.2, B)], axis=1)
    tv = rng.uniform(5.0, 12.0, B)
    zones = np.stack([O.planner_zones(tv[b], rng.uniform(6, 12), rng.uniform(0.3, 0.7)) for b in range(B)])
    prev0 = rng.uniform(0.0, 5.0, B)
    prev = torch.as_tensor(prev0, device=dev).clone()
    wps, prog, speeds = pl.request_waypoints_batched(torch.as_tensor(pose, device=dev), torch.as_tensor(lap, device=dev), prev,
                                                     torch.as_tensor(tv, device=dev), zones=zones, return_speeds=True)
    wps = wps.cpu().numpy(); speeds = speeds.cpu().numpy(); prog = prog.cpu().numpy(); prev = prev.cpu().numpy()
    for b in range(B):
        wr, sr, p0, pa = orc64.skidpad_waypoints_profile(*pose[b], int(lap[b]), prev0[b], tv[b], N, dt, zones[b])
        np.testing.assert_allclose(speeds[b], sr, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(wps[b], wr, rtol=1e-10, atol=1e-10)
        assert prog[b] == pytest.approx(p0, rel=1e-14, abs=1e-13) and prev[b] == pytest.approx(pa, rel=1e-14, abs=1e-13)


def _utol(u_ref, umax=0.9):
    return 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)


def test_skidpad_simulator_single_scenario(orc80):
    """SkidpadSimulator.step() composed from planner, controller and plant calls vs the checker's loop."""
    from oracle import oracle as O
    from mpc_amd.skidpad_simulator import SkidpadSimulator
    N, Tf, n = 10, 0.5, 25                       # the interactive demo setting of the reference (simulator.py:435-436)
    sim = SkidpadSimulator(N, Tf, model="L")
    cfg = O.make_cfg("L", N, Tf)
    st = sim.full_state.copy(); lap = 0; pp = 0.0
    for i in range(n):
        sim.step()
        _, st, lap, pp, u0, info = orc80.skidpad_step(cfg, st, lap, pp, 8.0)
        np.testing.assert_allclose(sim.full_state, st, rtol=0, atol=1e-8)
        assert sim.lap == lap
    assert sim.planned_trajectory.shape == (N + 1, 7) and sim.planned_references.shape == (N + 1, 4)
    assert len(sim.MPC_controller.metrics["runtime"]) == n


def test_lpv_controller_sequence(orc80):
    """LPVOcp.optimize over consecutive steps with prev_x/prev_u carried, both linearisation rules of LPVMPC.py:258-297."""
    from oracle import oracle as O
    from mpc_amd.controllers import LPVOcp
    from mpc_amd.skidpad_simulator import SkidpadSimulator
    _lpv_sequence(orc80, 50, 0.5)


def test_lpv_controller_sequence_horizon_80(orc80):
    """The LPV controller at its canonical N = 80, Tf = 0.8 (plot_generator.py:604): runs on the any-horizon kernel."""
    _lpv_sequence(orc80, 80, 0.8, nsteps=6)


@pytest.mark.parametrize("N,Tf,nsteps", [(10, 0.5, 8), (64, 0.64, 6), (100, 1.0, 4), (128, 1.28, 3)])
def test_lpv_controller_sequence_other_horizons(orc80, N, Tf, nsteps):
    """Smallest and largest register-resident horizon, and the any-horizon kernel up to its largest size."""
    _lpv_sequence(orc80, N, Tf, nsteps=nsteps)


def _lpv_sequence(orc80, N, Tf, nsteps=12):
    from oracle import oracle as O
    from mpc_amd.controllers import LPVOcp
    for mode in ("reference", "prev_iter"):
        cfg = O.make_cfg("LPV", N, Tf, lin_mode=mode)
        ctl = LPVOcp(N, Tf)
        lin_x = np.zeros((6, N + 1)); lin_u = np.zeros(N)
        st = np.array([15.0, 0.25, np.cos(0.05), np.sin(0.05), 8.0, 0.1, -0.1, 0.02]); lap = 0; pp = 0.0
        for i in range(nsteps):
            hd = np.arctan2(st[3], st[2])
            wp, sp, p0, pp = orc80.skidpad_waypoints(st[0], st[1], hd, lap, pp, 8.0, N, Tf / N)
            red = st[[0, 1, 2, 3, 5, 6, 7]]
            stt, traj_ref, U_ref, lam, info = orc80.optimize(cfg, red, wp, sp, 0.0, lin_x, lin_u)
            status, traj, inputs = ctl.optimize(red, wp, sp, lin_mode=mode)
            assert status == 0 and stt == 0
            assert (np.abs(inputs[:, 0] - U_ref) <= _utol(U_ref)).all(), (i, np.abs(inputs[:, 0] - U_ref).max())
            np.testing.assert_allclose(traj, traj_ref, rtol=1e-7, atol=1e-8)
            np.testing.assert_allclose(ctl.prev_x, lin_x, rtol=1e-7, atol=1e-8)
            st = orc80.plant_rk4(st, U_ref[0], Tf / N, cfg[3:9])


def _oracle_of_loop(orc, n, N, Tf, true0, est0):
    """Output-feedback loop of simulator.py:206-281 composed from checker primitives."""
    from oracle import oracle as O
    cfg = O.make_cfg("OFL", N, Tf); mp = cfg[3:9]; dt = Tf / N
    x = np.array(true0, dtype=np.float64); xe = np.array(est0, dtype=np.float64)
    P = np.diag([1, 1, 1, 1, 1, 10.0, 1.0, 1.0, 0.0, 1000000]); Qd = np.array([0.05, 0.05, 0.01, 0.01, 0.001, 0.1, 0.01, 0.01, 0, 0])
    Rd = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])
    xs, us, es = [], [], []
    for i in range(n):
        hd = np.arctan2(xe[3], xe[2])
        wp, sp, p0, ab = orc.step_waypoints(xe[0], xe[1], hd, true0[4], N, dt)
        red = xe[[0, 1, 2, 3, 5, 6, 7, 8, 9]]
        st, traj, U, lam, info = orc.optimize(cfg, red[:7], wp, sp, red[-1])
        x = orc.plant_rk4(x, U[0], dt, mp)
        xe, P = orc.ekf_time_update(xe, P, U[0], dt, mp, Qd)
        xe, P = orc.ekf_meas_update(xe, P, x[[0, 1, 2, 3, 4, 6, 7]], Rd)
        xs.append(x.copy()); us.append(U[0]); es.append(xe.copy())
    return np.array(xs), np.array(us), np.array(es)


def test_step_simulator_output_feedback(orc64):
    """StepSimulator(model='OFL').simulate_of: OFL controller on the EKF estimate, plant with side-force disturbance
    (the canonical experiment of plot_generator.py:16-29, shortened)."""
    from mpc_amd.simulator import StepSimulator
    N, Tf, n = 50, 0.5, 30
    true0 = [-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]
    sim = StepSimulator(N, Tf, starting_state=true0, model="OFL")
    xs, us, es, path = sim.simulate_of(n)
    est0 = [-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    xr, ur, er = _oracle_of_loop(orc64, n, N, Tf, true0, est0)
    # the loop feeds estimator and controller differences back into each other: 10x the per-step input bar
    assert (np.abs(us[:, 0] - ur) <= 10 * _utol(ur)).all(), np.abs(us[:, 0] - ur).max()
    np.testing.assert_allclose(xs, xr, rtol=0, atol=1e-7)
    np.testing.assert_allclose(es, er, rtol=1e-6, atol=1e-6)
    assert path.shape == (n, 2)


def test_batched_output_feedback_loop_fused(orc64):
    """mpc_ofl_closed_loop_run: planner + OFL solve + 10-state plant + EKF fused per scenario, several disturbance levels,
    with and without measurement noise, against the loop composed from checker primitives."""
    from oracle import oracle as O
    from mpc_amd.batched import StepOFLBatch
    N, Tf, n = 50, 0.5, 25
    rng = np.random.default_rng(2)
    B = 6
    true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))
    true0[:, 9] = rng.uniform(-12, -8, B); true0[:, 1] = rng.uniform(-0.1, 0.1, B)
    est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
    est0[:, 1] += rng.normal(0, 0.05, B)
    for cached in (True, False):
        sim = StepOFLBatch(N, Tf, true0, est0, fixed_model_cache=cached)
        xs, us, es = sim.run(n, record=True)
        xs, us, es = xs.cpu().numpy(), us.cpu().numpy(), es.cpu().numpy()
        assert (sim.status.cpu().numpy() == 0).all()
        for b in range(B):
            xr, ur, er = _oracle_of_loop(orc64, n, N, Tf, list(true0[b]), list(est0[b]))
            assert (np.abs(us[:, b] - ur) <= 10 * _utol(ur)).all(), (cached, b, np.abs(us[:, b] - ur).max())
            np.testing.assert_allclose(xs[:, b], xr, rtol=0, atol=1e-7)
            np.testing.assert_allclose(es[:, b], er, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("literal", [False, True])
def test_batched_output_feedback_loop_with_measurement_noise(orc64, literal):
    """The noisy output-feedback loop (OFL controller on the EKF estimate, simulator.py:206-281, with the measurements of
    continuous_dynamics.py:216-220 instead of the noiseless ones): library and checker are given the SAME noise tensor,
    either sigma * N(0,1) per channel or the reference's literal form (one N(10,1) scalar per measurement, i.e. every
    channel biased by ten sigma).  In the first form two scenarios carry thirty times the steering-channel noise; in both
    the estimated steering angle leaves its bound on some steps (status 4: both sides widen the bound the same way)."""
    from oracle import oracle as O
    from mpc_amd.batched import StepOFLBatch
    from mpc_amd.continuous_dynamics import Dynamics
    N, Tf, n, B = 50, 0.5, 80, 8
    rng = np.random.default_rng(12)
    true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))       # plot_generator.py:18-29
    true0[:, 9] = rng.uniform(-12, -8, B); true0[:, 1] = rng.uniform(-0.1, 0.1, B)
    est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))          # simulator.py:214-224
    est0[:, 1] += rng.normal(0, 0.05, B); est0[:, 5] += rng.normal(0, 0.02, B)
    noise = Dynamics(Tf / N, True).measurement_noise(n, B, literal=literal)
    if not literal:
        noise[:, :2, 6] *= 30.0
    sim = StepOFLBatch(N, Tf, true0, est0)
    xs, us, es = (t.cpu().numpy() for t in sim.run(n, record=True, meas_noise=noise))
    status = sim.status.cpu().numpy()
    x = true0.copy(); e = est0.copy(); P = np.tile(np.diag(O.EKF_P0_DIAG), (B, 1, 1))
    ur, xr, er, st_ref, _ = orc64.ofl_run(O.make_cfg("OFL", N, Tf), x, e, P, 15.0, n, noise=noise)
    assert set(np.unique(status)) <= {0, 4} and (status == st_ref).all(), (status, st_ref)
    assert (status == 4).any()                                   # the widened-bound steps are part of the comparison
    assert (np.abs(us - ur) <= 10 * _utol(ur)).all(), np.abs(us - ur).max()
    np.testing.assert_allclose(xs, xr, rtol=0, atol=1e-7)
    np.testing.assert_allclose(es, er, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sim.state.cpu().numpy(), x, rtol=0, atol=1e-7)
    np.testing.assert_allclose(sim.P.cpu().numpy(), P, rtol=1e-6, atol=1e-9)


def test_batched_output_feedback_loop_any_horizon(orc64):
    """The fused output-feedback loop at a horizon without a register-resident kernel (N = 70)."""
    from mpc_amd.batched import StepOFLBatch
    N, Tf, n, B = 70, 0.7, 12, 3
    true0 = np.tile(np.array([-5.0, 0.05, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))
    true0[:, 9] = [-12.0, -10.0, -8.0]
    est0 = true0.copy(); est0[:, 9] = 0.0
    sim = StepOFLBatch(N, Tf, true0, est0)
    xs, us, es = (t.cpu().numpy() for t in sim.run(n, record=True))
    assert (sim.status.cpu().numpy() == 0).all()
    for b in range(B):
        xr, ur, er = _oracle_of_loop(orc64, n, N, Tf, list(true0[b]), list(est0[b]))
        assert (np.abs(us[:, b] - ur) <= 10 * _utol(ur)).all(), (b, np.abs(us[:, b] - ur).max())
        np.testing.assert_allclose(xs[:, b], xr, rtol=0, atol=1e-7)
        np.testing.assert_allclose(es[:, b], er, rtol=1e-6, atol=1e-6)


def test_step_simulator_simulate_lsim_dlqr(orc64, golden):
    from oracle import oracle as O
    from mpc_amd.simulator import StepSimulator
    N, Tf, n = 50, 0.5, 20
    start = [-1.0, 0.0, 1.0, 0, 8.0, 0.0, 0.0, 0.0]
    sim = StepSimulator(N, Tf, starting_state=start, model="L")
    xs, us, ref = sim.simulate(n)
    cfg = O.make_cfg("L", N, Tf); x = np.array(start)
    for i in range(n):
        hd = np.arctan2(x[3], x[2])
        wp, sp, p0, ab = orc64.step_waypoints(x[0], x[1], hd, 8.0, N, Tf / N)
        st, traj, U, lam, info = orc64.optimize(cfg, x[[0, 1, 2, 3, 5, 6, 7]], wp, sp)
        assert abs(us[i, 0] - U[0]) <= 10 * _utol(np.array([U[0]]))[0]
        np.testing.assert_allclose(ref[i], ab[0], rtol=1e-13, atol=1e-13)
        x = orc64.plant_rk4(x, U[0], Tf / N, cfg[3:9])
        np.testing.assert_allclose(xs[i], x, rtol=0, atol=1e-7)
    # lsim: the reference-recorded EKF trace was produced by exactly this call pattern
    g = golden["ekf"]
    nn = len(g["ekf_u"])
    sim2 = StepSimulator(nn, nn * float(g["ekf_dt"]), starting_state=list(g["ekf_true0"]), model="none")
    st, ui, est = sim2.lsim(g["ekf_u"], 40, initial_state_estimate=list(g["ekf_x0"]))
    np.testing.assert_allclose(st, g["ekf_true"][:40], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(est, g["ekf_x"][:40], rtol=1e-8, atol=1e-9)
    # saturated LQR baseline with the DARE gain
    sim3 = StepSimulator(N, Tf, starting_state=start, model="L")
    xs3, us3, _ = sim3.dlqr_sim(15)
    K = sim3.MPC_controller.K; x = np.array(start)
    for i in range(15):
        u = K @ (np.array([1 / 500, 0, 0, 0, 0]) - x[[1, 3, 5, 6, 7]])
        u = np.clip(u, -0.9, 0.9); u = np.clip(u, (-0.3 - x[7]) / (Tf / N), (0.3 - x[7]) / (Tf / N))
        x = orc64.plant_rk4(x, float(u[0]), Tf / N, cfg[3:9])
        np.testing.assert_allclose(xs3[i], x, rtol=1e-12, atol=1e-12)


def test_batched_lqr_baseline_and_lsim(orc64, golden):
    """StepLQRBatch / LsimBatch: every scenario of one launch follows the per-scenario loops of StepSimulator.dlqr_sim
    and StepSimulator.lsim (checked against the CPU plant / EKF and the reference-recorded EKF trace)."""
    from oracle import oracle as O
    from mpc_amd.batched import LsimBatch, StepLQRBatch
    from mpc_amd.controllers import LOcp
    N, Tf, n = 50, 0.5, 60
    dt = Tf / N
    cfg = O.make_cfg("L", N, Tf)
    K0 = np.asarray(LOcp(N, Tf, _host_only=True).K, dtype=np.float64).reshape(5)
    rng = np.random.default_rng(11)
    B = 37
    starts = np.tile(np.array([-1.0, 0.0, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0]), (B, 1))
    starts[:, 1] = rng.uniform(-1.5, 1.5, B); starts[:, 7] = rng.uniform(-0.25, 0.25, B)
    gains = K0[None, :] * rng.uniform(0.3, 3.0, (B, 1))
    gains[0] = K0
    lqr = StepLQRBatch(dt, starts, gains)
    xs, us = lqr.run(n)
    xs = xs.cpu().numpy(); us = us.cpu().numpy()
    saturated = 0
    for b in range(B):
        x = starts[b].copy()
        for i in range(n):
            u = gains[b] @ (np.array([1 / 500, 0, 0, 0, 0]) - x[[1, 3, 5, 6, 7]])
            u0 = u
            u = np.clip(u, -0.9, 0.9); u = np.clip(u, (-0.3 - x[7]) / dt, (0.3 - x[7]) / dt)
            saturated += int(u != u0)
            assert abs(us[i, b] - u) <= 1e-9 * max(1.0, abs(u)), (b, i)
            x = orc64.plant_rk4(x, float(us[i, b]), dt, cfg[3:9])
            np.testing.assert_allclose(xs[i, b], x, rtol=1e-12, atol=1e-12)
    assert saturated > 50                                     # the clips are exercised
    np.testing.assert_array_equal(lqr.state.cpu().numpy(), xs[-1])
    # one shared gain == the same gain per scenario, bit for bit
    a = StepLQRBatch(dt, starts[:5], K0).run(20)[0].cpu().numpy()
    b_ = StepLQRBatch(dt, starts[:5], np.tile(K0, (5, 1))).run(20)[0].cpu().numpy()
    np.testing.assert_array_equal(a, b_)

    # lsim: scenario 0 replays the reference-recorded trace; the others carry their own input sequence
    g = golden["ekf"]
    nn = 40
    dt2 = float(g["ekf_dt"])
    Bl = 9
    true0 = np.tile(np.asarray(g["ekf_true0"], dtype=np.float64), (Bl, 1))
    est0 = np.tile(np.asarray(g["ekf_x0"], dtype=np.float64), (Bl, 1))
    true0[1:, 1] += rng.uniform(-0.5, 0.5, Bl - 1); est0[1:, 5] += rng.uniform(-0.2, 0.2, Bl - 1)
    u = np.tile(np.asarray(g["ekf_u"], dtype=np.float64)[:nn, None], (1, Bl))
    u[:, 1:] += rng.uniform(-0.2, 0.2, (nn, Bl - 1))
    sim = LsimBatch(dt2, true0, est0)
    st, est = sim.run(u)
    st = st.cpu().numpy(); est = est.cpu().numpy()
    np.testing.assert_allclose(st[:, 0], g["ekf_true"][:nn], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(est[:, 0], g["ekf_x"][:nn], rtol=1e-8, atol=1e-9)
    from mpc_amd.EKF import P0_DIAG, Q_DIAG
    mp = cfg[3:9]
    Rd = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])
    for b in range(Bl):
        x = true0[b].copy(); xe = est0[b].copy(); P = np.diag(np.asarray(P0_DIAG, dtype=np.float64))
        for i in range(nn):
            x = orc64.plant_rk4(x, float(u[i, b]), dt2, mp)
            xe, P = orc64.ekf_time_update(xe, P, float(u[i, b]), dt2, mp, np.asarray(Q_DIAG, dtype=np.float64))
            xe, P = orc64.ekf_meas_update(xe, P, x[[0, 1, 2, 3, 4, 6, 7]], Rd)[:2]
            np.testing.assert_allclose(st[i, b], x, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(est[i, b], xe, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sim.P.cpu().numpy()[Bl - 1], P, rtol=1e-7, atol=1e-9)
    # a shared (n,) sequence == the same sequence handed over per scenario
    s1 = LsimBatch(dt2, true0, est0).run(u[:, 0].copy())[1].cpu().numpy()
    s2 = LsimBatch(dt2, true0, est0).run(np.tile(u[:, :1], (1, Bl)))[1].cpu().numpy()
    np.testing.assert_array_equal(s1, s2)


def test_control_buffer_interp_matches_numpy():
    """archive/mpc_node.py:421-427 is np.interp over linspace(0, Tf, N+1); numpy is the reference's own arithmetic here."""
    import torch
    from mpc_amd.batched import control_buffer_interp
    rng = np.random.default_rng(5)
    for N, Tf in ((50, 0.5), (80, 2.0), (7, 0.35), (128, 1.28)):
        B = 300
        traj = rng.normal(size=(B, N + 1, 7))
        nodes = np.linspace(0, Tf, N + 1)
        t = rng.uniform(-0.1 * Tf, 1.1 * Tf, B)
        t[:N + 1] = nodes[:min(B, N + 1)]                       # exactly on the nodes
        t[N + 1:2 * N + 1] = np.nextafter(nodes[:N], 1.0)[:B - N - 1]
        want = np.array([np.interp(t[b], nodes, traj[b, :, 6]) for b in range(B)]) * np.pi / 2 / 0.4
        got = control_buffer_interp(torch.as_tensor(traj, device="cuda"), Tf, t).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-14)
        got0 = control_buffer_interp(torch.as_tensor(traj, device="cuda"), Tf, 0.013, col=2, scale=1.0).cpu().numpy()
        np.testing.assert_allclose(got0, [np.interp(0.013, nodes, traj[b, :, 2]) for b in range(B)], rtol=1e-13, atol=1e-14)


def test_invariance_check_and_bisection(golden):
    """stability.py's vertex test and bisection on the GPU (a batch of levels per launch) against the numpy restatement,
    on the reference's printed (A, B) of terminal_set.m with the DARE gain; saturated and linear closed loops."""
    import scipy.linalg
    from oracle import oracle as O
    from mpc_amd.stability import (ClosedLoopMap, binary_search, check_control_amissible_invariance,
                                   check_invariance_batched)
    g = golden["kat_terminal_set"]
    A = np.asarray(g["A"], dtype=np.float64); B = np.asarray(g["B"], dtype=np.float64).reshape(5, 1)
    Qw = np.diag([1e5, 1.0, 1e-3, 1e-3, 1.0]); R = np.array([[10.0]])
    P = scipy.linalg.solve_discrete_are(A, B, Qw, R)
    K = np.linalg.solve(B.T @ P @ B + R, B.T @ P @ A)
    seen = set()
    for umax in (0.9, np.inf, 0.05):
        f = ClosedLoopMap(A, B, K, umax)
        fref = lambda x, u=umax: A @ x + B[:, 0] * np.clip(-(K @ x)[0], -u, u)
        cs = np.concatenate([np.logspace(-6, 6, 61), [1e-12, 1e9]])
        ok = check_invariance_batched(P, f, cs)
        ref = np.array([O.invariance_check(P, fref, c) for c in cs])
        np.testing.assert_array_equal(ok, ref)
        seen |= set(ok.tolist())
        assert check_control_amissible_invariance(P, f, cs[7]) == ref[7]
        for c_u, eps in ((1e4, 1e-1), (50.0, 1e-2), (0.05, 1e-1)):
            assert binary_search(P, f, c_u, eps) == O.invariance_binary_search(P, fref, c_u, eps)
    # a contraction passes at every level (the printed system above never does: the box is an outer approximation), and a
    # saturated loop around an unstable plant passes only while the input stays unsaturated
    Au = np.diag([1.05, 0.9, 0.8, 0.7, 0.6]); Bu = np.array([[1.0], [0.0], [0.0], [0.0], [0.0]]); Ku = np.array([[0.6, 0, 0, 0, 0.0]])
    Pu = np.diag([1.0, 2.0, 3.0, 4.0, 5.0])
    for (Ax, Bx, Kx, ux) in ((0.5 * np.eye(5), B, np.zeros((1, 5)), np.inf), (Au, Bu, Ku, 0.5)):
        f = ClosedLoopMap(Ax, Bx, Kx, ux)
        fref = lambda x, A_=Ax, B_=Bx, K_=Kx, u=ux: A_ @ x + B_[:, 0] * np.clip(-(K_ @ x)[0], -u, u)
        cs = np.logspace(-4, 4, 41)
        ok = check_invariance_batched(Pu, f, cs)
        np.testing.assert_array_equal(ok, [O.invariance_check(Pu, fref, c) for c in cs])
        seen |= set(ok.tolist())
        assert binary_search(Pu, f, 1e4, 1e-1) == O.invariance_binary_search(Pu, fref, 1e4, 1e-1)
    assert seen == {True, False}
    with pytest.raises(TypeError):
        check_control_amissible_invariance(P, lambda x: x, 1.0)


def test_library_error_paths():
    """Argument checks of the C ABI: unsupported horizon, oversize batch, missing buffers raise with a message."""
    from mpc_amd.controllers import LOcp
    with pytest.raises(RuntimeError, match="horizon"):
        LOcp(129, 1.29)
    ctl = LOcp(10, 0.5, batch=2)
    with pytest.raises(ValueError, match="exceeds"):
        ctl.solve_batched(np.zeros((3, 7)), np.zeros((3, 11, 4)), np.full((3, 11), 8.0))
    with pytest.raises(ValueError):
        ctl.optimize(np.zeros(5), np.zeros((11, 4)), np.full(11, 8.0))
    with pytest.raises(RuntimeError, match="no solve"):
        ctl.stats()
