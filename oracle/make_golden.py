"""Generate tests/golden/*.npz by IMPORTING the reference's own modules (run in the build container only).

Importable reference modules: continuous_dynamics.Dynamics, EKF.CarEKF,
utils.path_planning.SkidpadPlanner, utils.step_planning.StepPlanner (numba is
only referenced by a dead spec list, so a stub module satisfies the import).
The controllers (LMPC2/LPVMPC/OFLMPC2/NLMPC) need acados/casadi/control and
cannot be imported: no solver output is recorded here.

The fixtures hold inputs and the reference's outputs only; no reference source
text is stored.  Usage:  python oracle/make_golden.py [/root/reference [fixture ...]]
(fixture names, e.g. "measurement", restrict the run to those files; default: all)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ONLY = set(sys.argv[2:])


def save(name, **rec):
    if not ONLY or name in ONLY:
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def main():
    import matplotlib
    matplotlib.use("Agg")
    nb = types.ModuleType("numba")
    nb.float32 = "float32"; nb.int16 = "int16"
    sys.modules.setdefault("numba", nb)
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from continuous_dynamics import Dynamics
        from EKF import CarEKF
        from utils.path_planning import SkidpadPlanner
        from utils.step_planning import StepPlanner
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20250523)

    # ---- plant: single_track_model and rk4_integraton, nx = 8 and 10, several dt ----
    rec = {}
    for nx, dist in ((8, False), (10, True)):
        xs, us, dts, fs, x1s = [], [], [], [], []
        for dt in (0.002, 0.01, 0.02, 0.05):
            with contextlib.redirect_stdout(io.StringIO()):
                dyn = Dynamics(dt, disturbance=dist)
            for _ in range(16):
                psi = rng.uniform(-np.pi, np.pi)
                x = np.array([rng.uniform(-5, 40), rng.uniform(-20, 20), np.cos(psi), np.sin(psi), rng.uniform(4, 16),
                              rng.uniform(-1, 1), rng.uniform(-1.5, 1.5), rng.uniform(-0.3, 0.3)])
                if dist:
                    x = np.concatenate([x, [rng.uniform(-0.05, 0.05), rng.uniform(-12, 2)]])
                u = rng.uniform(-0.9, 0.9)
                xs.append(x); us.append(u); dts.append(dt)
                fs.append(dyn.single_track_model(x, u)); x1s.append(dyn.rk4_integraton(x, u))
        rec[f"x{nx}"] = np.array(xs); rec[f"u{nx}"] = np.array(us); rec[f"dt{nx}"] = np.array(dts)
        rec[f"f{nx}"] = np.array(fs); rec[f"x1_{nx}"] = np.array(x1s)
    with contextlib.redirect_stdout(io.StringIO()):
        dyn = Dynamics(0.01)
    rec["Cf_Cr"] = np.array([dyn.Cf, dyn.Cr])
    # a known-answer case: dt=0.01, x=[15,0.1,1,0,8,0,0,0], u=0.5
    rec["kat_x1"] = dyn.rk4_integraton(np.array([15, 0.1, 1.0, 0, 8.0, 0, 0, 0]), 0.5)
    save("plant", **rec)

    # ---- EKF: Jacobians, and a full predict/update trace driven like StepSimulator.lsim ----
    rec = {}
    with contextlib.redirect_stdout(io.StringIO()):
        dyn = Dynamics(0.01, disturbance=True)
    xj, Fj = [], []
    for _ in range(24):
        psi = rng.uniform(-np.pi, np.pi)
        x = np.array([rng.uniform(-5, 40), rng.uniform(-20, 20), np.cos(psi), np.sin(psi), rng.uniform(4, 16),
                      rng.uniform(-1, 1), rng.uniform(-1.5, 1.5), rng.uniform(-0.3, 0.3), rng.uniform(-0.05, 0.05),
                      rng.uniform(-12, 2)])
        xj.append(x); Fj.append(dyn.jacobian_forward_euler(x)[2])
    rec["jac_x"] = np.array(xj); rec["jac_F"] = np.array(Fj); rec["jac_dt"] = np.array(0.01)
    n = 200
    dt = 0.01
    with contextlib.redirect_stdout(io.StringIO()):
        plant = Dynamics(dt, disturbance=True)
        ekf = CarEKF(dt, True, inital_state=[0.0, 0.0, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    x = np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0])
    u = 0.1 * np.ones(n); u[100:] = 0.0
    tr_x, tr_P, tr_xp, tr_Pp, tr_true, tr_y = [], [], [], [], [], []
    rec["ekf_P0"] = ekf.P.copy(); rec["ekf_Q"] = np.diag(ekf.Q).copy(); rec["ekf_R"] = np.diag(ekf.R).copy()
    rec["ekf_x0"] = ekf.x_est.copy(); rec["ekf_true0"] = x.copy()
    for i in range(n):
        x = plant.rk4_integraton(x, u[i])
        ekf.time_update(u[i])
        tr_xp.append(ekf.x_est.copy()); tr_Pp.append(ekf.P.copy())
        y = plant.measure_state_noiseless(x)
        ekf.measurement_update(y)
        tr_x.append(ekf.x_est.copy()); tr_P.append(ekf.P.copy()); tr_true.append(x.copy()); tr_y.append(y.copy())
    rec.update(ekf_u=u, ekf_dt=np.array(dt), ekf_x_pred=np.array(tr_xp), ekf_P_pred=np.array(tr_Pp),
               ekf_x=np.array(tr_x), ekf_P=np.array(tr_P), ekf_true=np.array(tr_true), ekf_y=np.array(tr_y),
               ekf_red=ekf.estimated_red_state())
    save("ekf", **rec)

    # ---- planners ----
    rec = {}
    # (a) the reference's own saved output utils/waypoints.npy and the call that makes it
    pl = SkidpadPlanner(target_vel=9.0, Nt=500, dt=0.01)
    wp, sp, p0, R = pl.request_waypoints(18, 0, 0, 5)
    saved = np.load(os.path.join(REF, "utils", "waypoints.npy"))
    assert np.array_equal(wp, saved), "reference planner no longer reproduces utils/waypoints.npy"
    rec["saved_waypoints"] = saved
    # (b) a synthetic drive around the whole figure-8 with the lap bookkeeping of SkidpadSimulator
    N, dt, v = 50, 0.01, 8.0
    pl = SkidpadPlanner(target_vel=v, Nt=N, dt=dt)
    poses, laps, prevs, wps, p0s, prev_after = [], [], [], [], [], []
    lap = 0
    center, r = 16.75, 9.125
    L = 2 * np.pi * r
    s = 14.0
    oldx = s
    k = 0
    while s < center + 4 * L + 12:
        # centre-line pose at arc length s plus a small wobble
        if s < center:
            px, py, hd = s, 0.0, 0.0
        elif s < center + 2 * L:
            a = (s - center) / r; px, py, hd = center + np.sin(a) * r, -r + np.cos(a) * r, -a
        elif s < center + 4 * L:
            a = (s - center - 2 * L) / r; px, py, hd = center + np.sin(a) * r, r - np.cos(a) * r, a
        else:
            px, py, hd = s - 4 * L, 0.0, 0.0
        px += 0.05 * np.sin(0.7 * k); py += 0.08 * np.cos(0.9 * k); hd += 0.03 * np.sin(0.5 * k)
        if px > center and oldx <= center:
            lap += 1
        oldx = px
        hdw = np.arctan2(np.sin(hd), np.cos(hd))
        poses.append([px, py, hdw]); laps.append(lap); prevs.append(pl.prev_progress)
        w, spd, p0, R = pl.request_waypoints(px, py, hdw, lap)
        wps.append(w.copy()); p0s.append(p0); prev_after.append(pl.prev_progress)
        s += 0.9 if k % 3 else 0.08
        k += 1
    rec.update(sk_pose=np.array(poses), sk_lap=np.array(laps), sk_prev=np.array(prevs), sk_wp=np.array(wps),
               sk_p0=np.array(p0s), sk_prev_after=np.array(prev_after), sk_N=np.array(N), sk_dt=np.array(dt),
               sk_v=np.array(v))
    # (c) step planner
    N, dt, v = 50, 0.01, 15.0
    spn = StepPlanner(target_vel=v, Nt=N, dt=dt)
    poses, wps, absw = [], [], []
    for _ in range(40):
        px, py, hd = rng.uniform(-6, 3), rng.uniform(-0.5, 1.5), rng.uniform(-0.3, 0.3)
        w, spd, p0, R, ab = spn.request_waypoints(px, py, hd)
        poses.append([px, py, hd]); wps.append(w.copy()); absw.append(ab.copy())
    rec.update(st_pose=np.array(poses), st_wp=np.array(wps), st_abs=np.array(absw), st_N=np.array(N),
               st_dt=np.array(dt), st_v=np.array(v))
    save("planners", **rec)

    # ---- planner speed profile (laps outside 0..7), progress2speed, request_desired_speed ----
    rec = {}
    prng = np.random.default_rng(77)
    cases = []
    for (v, acc, fac, start) in ((9.0, 10.0, 0.5, None), (8.0, 10.0, 0.5, 3.0), (12.0, 6.0, 0.3, None), (5.0, 10.0, 0.7, None)):
        pl = SkidpadPlanner(target_vel=v, max_accel=acc, Nt=40, dt=(0.05 if fac == 0.5 else 0.1), slowdown_speed_factor=fac)
        if start is not None:
            pl.set_starting_position(start)
        zones = np.array([pl.accel_zone_1_start, pl.accel_zone_1_end, pl.braking_zone_1_start, pl.braking_zone_1_end,
                          pl.braking_zone_2_start, pl.braking_zone_2_end, pl.slowdown_speed_factor], dtype=np.float64)
        qs = np.concatenate([np.linspace(-2.0, 30.0, 60), np.linspace(4 * 2 * 9.125 * np.pi + 10, 4 * 2 * 9.125 * np.pi + 45, 120),
                             zones[:6]])
        sp = np.array([pl.progress2speed(q) for q in qs], dtype=np.float64)
        ds = np.array([pl.request_desired_speed(q) for q in qs], dtype=np.float64)
        reqs = []
        for lap in (8, 9, -1, 8, 8, 8, 8, 8):
            px = prng.uniform(-1.0, 48.0); py = prng.uniform(-0.4, 0.4); hd = prng.uniform(-0.2, 0.2)
            N = int(prng.choice([10, 40, 80])); dtt = pl.dt     # an explicit dt leaves prev_progress unbound there (:223-228)
            prev = pl.prev_progress
            w, spd, p0, R = pl.request_waypoints(px, py, hd, lap, Nt=N)
            reqs.append(dict(pose=np.array([px, py, hd]), lap=lap, N=N, dt=dtt, prev=prev, wp=w.copy(), speeds=spd.copy(),
                             p0=p0, prev_after=pl.prev_progress))
        cases.append(dict(v=v, zones=zones, qs=qs, sp=sp, ds=ds, reqs=reqs))
    for ci, c in enumerate(cases):
        rec[f"c{ci}_v"] = np.array(c["v"]); rec[f"c{ci}_zones"] = c["zones"]; rec[f"c{ci}_qs"] = c["qs"]
        rec[f"c{ci}_sp"] = c["sp"]; rec[f"c{ci}_ds"] = c["ds"]; rec[f"c{ci}_nreq"] = np.array(len(c["reqs"]))
        for ri, q in enumerate(c["reqs"]):
            for k, val in q.items():
                rec[f"c{ci}_r{ri}_{k}"] = np.asarray(val)
    rec["ncases"] = np.array(len(cases))
    save("planner_speed_profile", **rec)

    # ---- step planner with a ramped step (ramp_length > 0, step_planning.py:75-79) ----
    rec = {}
    rrng = np.random.default_rng(4242)
    for ci, (v, N, dt, ramp) in enumerate(((15.0, 50, 0.01, 2.0), (8.0, 80, 0.0125, 5.5), (10.0, 30, 0.02, 0.75))):
        spn = StepPlanner(target_vel=v, Nt=N, dt=dt, ramp_length=ramp)
        poses, wps, absw, spds, p0s = [], [], [], [], []
        for _ in range(24):
            px, py, hd = rrng.uniform(-6, ramp + 2), rrng.uniform(-0.5, 1.5), rrng.uniform(-0.3, 0.3)
            w, spd, p0, R, ab = spn.request_waypoints(px, py, hd)
            poses.append([px, py, hd]); wps.append(w.copy()); absw.append(ab.copy()); spds.append(spd.copy()); p0s.append(p0)
        rec.update({f"c{ci}_pose": np.array(poses), f"c{ci}_wp": np.array(wps), f"c{ci}_abs": np.array(absw),
                    f"c{ci}_speeds": np.array(spds), f"c{ci}_p0": np.array(p0s), f"c{ci}_N": np.array(N),
                    f"c{ci}_dt": np.array(dt), f"c{ci}_v": np.array(v), f"c{ci}_ramp": np.array(ramp)})
    rec["ncases"] = np.array(3)
    save("step_ramp", **rec)

    # ---- measurement model: measure_state_noiseless and measure_state_noise (continuous_dynamics.py:212-223).  The noisy
    # one adds measurement_noises times ONE scalar drawn as rng.normal(len(x)), i.e. N(len(x), 1), from default_rng(1);
    # three consecutive calls per model size are recorded, each on its own state.
    rec = {}
    mrng = np.random.default_rng(77)
    for nx, dist in ((8, False), (10, True)):
        with contextlib.redirect_stdout(io.StringIO()):
            dyn = Dynamics(0.01, disturbance=dist)
        xs = mrng.uniform(-2, 2, (3, nx)); xs[:, 4] += 10
        rec[f"x{nx}"] = xs
        rec[f"y_noiseless{nx}"] = np.array([dyn.measure_state_noiseless(x) for x in xs])
        rec[f"y_noisy{nx}"] = np.array([dyn.measure_state_noise(x) for x in xs])
        rec[f"H{nx}"] = np.array(dyn.measurement_jacobian())
    rec["sigma"] = np.array(dyn.measurement_noises)
    save("measurement", **rec)

    # ---- known answers held by the reference: terminal_set.m:4-10 (A 5x5, B at vx=15, dt=0.02), values as printed ----
    A = np.array([[1.0, 0.3, 0.02, 0.0, 0.0],
                  [0.0, 1.0, 0.0, 0.02, 0.0],
                  [0.0, 0.0, 0.6802073, -0.00369231, -2.0977733],
                  [0.0, 0.0, -0.00219257, 0.8871941, -1.1200825],
                  [0.0, 0.0, 0.0, 0.0, 1.0]])
    B = np.array([0.0, 0.0, 0.0, 0.0, 0.02])
    save("kat_terminal_set", A=A, B=B, dt=np.array(0.02), vx=np.array(15.0))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
