"""The reference's own single-scenario configuration end to end: SkidpadSimulator(N=50, Tf=0.5, model="L") from its
default start (skidpad_simulator.py:50-60), 3400 closed-loop steps = the whole skidpad (entry straight, two right and two
left circles, exit), as one fused launch, against the checker's loop.  Gates: position within 1e-3 m and yaw within
1e-4 rad along the whole run (differences feed back through the loop, so this is looser than the per-step input bar),
every lap counted alike, and sampled solutions certified by their own KKT residuals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c1_full_skidpad_run_matches_checker(orc64):
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    N, Tf, nsteps = 50, 0.5, 3400
    start = np.array([[15.0, 0.1, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0]])
    cfg = O.make_cfg("L", N, Tf)
    ref = start.copy(); laps = np.zeros(1, np.int32); pp = np.zeros(1)
    # the checker's trajectory, recorded every 100 steps
    marks = []
    for blk in range(nsteps // 100):
        orc64.skidpad_run(cfg, ref, laps, pp, 8.0, 100, record=False)
        marks.append((ref[0].copy(), int(laps[0])))
    sim = SkidpadBatch(N, Tf, start, model="L")
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    xs = xs.cpu().numpy()[:, 0]
    assert (sim.status.cpu().numpy() == 0).all()
    for blk, (xr, lap) in enumerate(marks):
        x = xs[100 * (blk + 1) - 1]
        assert np.hypot(x[0] - xr[0], x[1] - xr[1]) <= 1e-3, (blk, x, xr)
        dyaw = np.arctan2(x[3], x[2]) - np.arctan2(xr[3], xr[2])
        assert abs((dyaw + np.pi) % (2 * np.pi) - np.pi) <= 1e-4, (blk, dyaw)
    assert int(sim.lap.item()) == marks[-1][1] == 5           # entry + 4 circles done
    assert abs(np.hypot(xs[:, 2], xs[:, 3]) - 1).max() < 1e-6  # the heading vector stays normalised (skidpad_simulator.py:134)


def _full_run(orc, model, N, Tf, nsteps=3400, ctl_kw=None, cfg_kw=None):
    """The checker's run from the default start (skidpad_simulator.py:50-60), recorded every 100 steps with the
    controller's memory (LPV prev_x / prev_u, NL iterate) carried from block to block, and the library's fused run."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    start = np.array([[15.0, 0.1, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0]])
    cfg = O.make_cfg(model, N, Tf, **(cfg_kw or {}))
    ref = start.copy(); laps = np.zeros(1, np.int32); pp = np.zeros(1)
    lx = np.zeros((1, 6, N + 1)); lu = np.zeros((1, N))
    marks = []
    for blk in range(nsteps // 100):
        orc.skidpad_run(cfg, ref, laps, pp, 8.0, 100, record=False, lin_x=lx, lin_u=lu)
        marks.append((ref[0].copy(), int(laps[0])))
    sim = SkidpadBatch(N, Tf, start, model=model, **(ctl_kw or {}))
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    return sim, xs.cpu().numpy()[:, 0], us.cpu().numpy()[:, 0], marks


def _c1_gates(sim, xs, marks, laps_expected=5):
    assert (sim.status.cpu().numpy() == 0).all()
    worst_pos = worst_yaw = 0.0
    for blk, (xr, lap) in enumerate(marks):
        x = xs[100 * (blk + 1) - 1]
        worst_pos = max(worst_pos, np.hypot(x[0] - xr[0], x[1] - xr[1]))
        dyaw = np.arctan2(x[3], x[2]) - np.arctan2(xr[3], xr[2])
        worst_yaw = max(worst_yaw, abs((dyaw + np.pi) % (2 * np.pi) - np.pi))
    assert worst_pos <= 1e-3 and worst_yaw <= 1e-4, (worst_pos, worst_yaw)
    assert int(sim.lap.item()) == marks[-1][1] == laps_expected
    assert abs(np.hypot(xs[:, 2], xs[:, 3]) - 1).max() < 1e-6
    return worst_pos, worst_yaw


@pytest.mark.parametrize("layout", ["wave", "lane"])
def test_default_controller_full_skidpad_run_matches_checker(orc64, layout):
    """The reference's default controller, SkidpadSimulator(model="LPV") (skidpad_simulator.py:24) at its canonical
    N = 80, Tf = 0.8 (plot_generator.py:604): the whole skidpad, through every circle change and lap bump, in both solver
    layouts, with the C1 gates (1e-3 m, 1e-4 rad, equal lap counts)."""
    sim, xs, us, marks = _full_run(orc64, "LPV", 80, 0.8, ctl_kw={"solver_layout": layout})
    _c1_gates(sim, xs, marks)
    assert np.abs(us).max() <= 0.9 + 1e-9 and np.abs(xs[:, 7]).max() <= 0.3 + 1e-6


def test_nonlinear_controller_full_skidpad_run_matches_checker(orc64):
    """SkidpadSimulator(model="NL"), N = 50: the non-convex programme's local solution depends on the iterate history,
    so the run also shows that library and checker stay on the same branch through the whole figure of eight."""
    kw = dict(sqp_tol=1e-10, sqp_max_iter=50)
    sim, xs, us, marks = _full_run(orc64, "NL", 50, 0.5, ctl_kw=kw, cfg_kw=dict(sqp_tol=1e-11, sqp_max_iter=50))
    _c1_gates(sim, xs, marks)


@pytest.mark.parametrize("kind", ["L", "OFL"])
def test_sampled_solutions_satisfy_kkt(orc64, kind):
    """Independent of the checker's solver: stationarity of H U + g over the active set by non-negative least squares,
    primal feasibility, on scenarios with active input and steering bounds."""
    import torch
    from scipy.optimize import nnls
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, OFLOcp
    N, Tf, B = 50, 0.5, 16
    rng = np.random.default_rng(3)
    cfg = O.make_cfg(kind, N, Tf)
    ctl = (LOcp if kind == "L" else OFLOcp)(N, Tf, batch=B)
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([15.0 + rng.uniform(0, 30, B), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.28, 0.28, B)], axis=1)
    x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc64.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
    U, traj, status, iters = ctl.solve_batched(x0, np.array(wps), np.array(ps), dh)
    torch.cuda.synchronize()
    U = U.cpu().numpy()
    nactive = 0
    for b in range(B):
        H, g, A, lo, up = O.condensed_qp(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b])
        v = A @ U[b]
        assert (v <= up + 1e-9).all() and (v >= lo - 1e-9).all()
        at_up = v >= up - 1e-7; at_lo = v <= lo + 1e-7
        G = np.hstack([A[at_up].T, -A[at_lo].T])              # H U + g + G lam = 0 with lam >= 0
        r = H @ U[b] + g
        if G.shape[1]:
            lam, rn = nnls(G, -r)
            nactive += 1
        else:
            rn = np.linalg.norm(r)
        assert rn <= 1e-8 * max(1.0, np.abs(g).max()), (b, rn, np.abs(g).max())
    assert nactive >= 4
