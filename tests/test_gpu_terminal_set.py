"""Terminal set C (x_N[1:] - x_ref) <= 1 of the linear controllers: the constraint the reference prepares
(LMPC2.py:444-449, stability.py:93-112) and leaves commented out (LMPC2.py:212-218), solved by the lane-per-scenario
layout and compared with the checker given the same rows.  Parity unpinned like every QP solve (no solver output in the
reference); a dense KKT certificate (non-negative least squares over the active rows) checks the answers independently
of both solvers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _utol(u_ref, umax=0.9):
    return 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)


def _cases(orc, kind, N, Tf, B, seed):
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    cfg = O.make_cfg(kind, N, Tf)
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([15.0 + rng.uniform(0, 30, B), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.28, 0.28, B)], axis=1)
    x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    return cfg, x0, np.array(wps), np.array(ps)


def _terminal_error(kind, traj, wp):
    """x_N[1:] - x_ref from a returned trajectory (columns x, y, cos psi, sin psi, vy, r, delta)."""
    N = traj.shape[0] - 1
    e = np.array([traj[N, 1], np.arcsin(traj[N, 3]), traj[N, 4], traj[N, 5], traj[N, 6]])
    if kind != "LPV":
        e[0] -= wp[N, 1]; e[1] -= wp[N, 3]
    return e


def _rows_cutting(errors, rng, cut=0.9, ncut=4, extra=6):
    """Rows that cut into the terminal errors of the problem without the set: a random direction over (psi, vy, r, delta)
    scaled so that the scenario with the third largest value has to give up 1 - cut of it (mild, so that the set stays
    reachable within the input bounds for most scenarios), plus loose rows that are never active."""
    rows = []
    for _ in range(ncut):
        d = np.zeros(5); d[1:] = rng.standard_normal(4) * np.array([1.0, 0.3, 0.3, 1.0])
        v = np.sort(errors @ d)
        if v[-3] <= 0:
            d, v = -d, np.sort(-(errors @ d))
        rows.append(d / (cut * max(v[-3], 0.2 * v[-1])))
    for _ in range(extra):
        d = rng.standard_normal(5)
        rows.append(d / (5.0 * np.abs(errors @ d).max()))
    return np.array(rows)


@pytest.mark.parametrize("kind,N,Tf", [("L", 50, 0.5), ("OFL", 50, 0.5), ("LPV", 80, 0.8), ("L", 25, 0.5)])
def test_terminal_rows_solve_matches_checker(orc80, kind, N, Tf):
    import torch
    from scipy.optimize import nnls
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, LPVOcp, OFLOcp
    B = 12
    cfg, x0, wps, ps = _cases(orc80, kind, N, Tf, B, 11)
    rng = np.random.default_rng(5)
    dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
    lin_x = np.zeros((B, 6, N + 1)); lin_u = np.zeros((B, N))

    def checker(b):
        lx, lu = (lin_x[b].copy(), lin_u[b].copy()) if kind == "LPV" else (None, None)
        return orc80.optimize(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b], lx, lu)

    free = [checker(b) for b in range(B)]
    errors = np.array([_terminal_error(kind, free[b][1], wps[b]) for b in range(B)])
    C = _rows_cutting(errors, rng)
    ctl = {"L": LOcp, "OFL": OFLOcp, "LPV": LPVOcp}[kind](N, Tf, batch=B, terminal_set=C)
    def lin():      # fresh linearisation memory on the device (the solve updates it in place)
        return dict(lin_x=torch.zeros((B, 6, N + 1), dtype=torch.float64, device=ctl.device),
                    lin_u=torch.zeros((B, N), dtype=torch.float64, device=ctl.device)) if kind == "LPV" else {}

    U, traj, status, iters = ctl.solve_batched(x0, wps, ps, dh, **lin())
    torch.cuda.synchronize()
    U = U.cpu().numpy(); traj = traj.cpu().numpy(); status = status.cpu().numpy()
    orc80.set_terminal_set(C)
    try:
        ref = [checker(b) for b in range(B)]
    finally:
        orc80.set_terminal_set(None)
    nactive = nmoved = 0
    for b in range(B):
        st_r, tr_r, U_r = ref[b][0], ref[b][1], ref[b][2]
        # a set that cannot be reached within the input bounds leaves no minimiser: both sides must say so (status 2)
        assert status[b] == st_r and st_r in (0, 2), (b, st_r, status[b])
        if st_r == 2:
            continue
        err = np.abs(U[b] - U_r)
        assert (err <= _utol(U_r)).all(), (kind, b, err.max())
        np.testing.assert_allclose(traj[b], tr_r, rtol=0, atol=1e-7)
        s = C @ _terminal_error(kind, traj[b], wps[b])
        assert (s <= 1 + 1e-9).all()
        nactive += int((s >= 1 - 1e-7).any())
        nmoved += int(np.abs(U_r - free[b][2]).max() > 1e-3)
        if kind != "LPV":
            H, g, A, lo, up = O.condensed_qp(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b], terminal_C=C)
            v = A @ U[b]
            assert (v <= up + 1e-9).all() and (v >= lo - 1e-9).all()
            at_up = v >= up - 1e-7; at_lo = v <= lo + 1e-7
            G = np.hstack([A[at_up].T, -A[at_lo].T])
            r = H @ U[b] + g
            rn = nnls(G, -r)[1] if G.shape[1] else np.linalg.norm(r)
            assert rn <= 1e-8 * max(1.0, np.abs(g).max()), (b, rn)
    assert nactive >= 3 and nmoved >= 3, (nactive, nmoved)
    # removing the set gives back the problem without it
    ctl.set_terminal_set(None)
    U0, _, status0, _ = ctl.solve_batched(x0, wps, ps, dh, **lin())
    U0 = U0.cpu().numpy()
    for b in range(B):
        assert (np.abs(U0[b] - free[b][2]) <= _utol(free[b][2])).all()


def test_closed_loop_with_terminal_set_matches_checker(orc80):
    """Skidpad closed loop of the L controller with a terminal set that is active over the first steps (warm starts
    carry the terminal multipliers from step to step), against the checker's loop with the same rows."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    import bench
    N, Tf, B, nsteps = 50, 0.5, 6, 30
    cfg = O.make_cfg("L", N, Tf)
    st = bench.initial_states(B, 4)
    # rows from the terminal errors of the first solve without the set: each direction scaled on the scenario that is
    # largest in it, which then has to give up a tenth
    errors = []
    for b in range(B):
        wp, sp, _, _ = orc80.skidpad_waypoints(st[b, 0], st[b, 1], np.arctan2(st[b, 3], st[b, 2]), 0, 0.0, st[b, 4], N, Tf / N)
        _, traj, _, _, _ = orc80.optimize(cfg, st[b, [0, 1, 2, 3, 5, 6, 7]], wp, sp)
        errors.append(_terminal_error("L", traj, wp))
    errors = np.array(errors)
    rows = []
    for j in (1, 2, 3, 4):
        for sgn in (1.0, -1.0):
            d = np.zeros(5); d[j] = sgn
            rows.append(d / (0.9 * (errors @ d).max()))
    C = np.array(rows)
    sim = SkidpadBatch(N, Tf, st, model="L", solver_layout="lane", terminal_set=C)
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    us = us.cpu().numpy(); xs = xs.cpu().numpy()
    assert (sim.status.cpu().numpy() == 0).all()
    sim0 = SkidpadBatch(N, Tf, st, model="L", solver_layout="lane")
    _, us0 = sim0.run(nsteps, record=True)
    us0 = us0.cpu().numpy()
    orc80.set_terminal_set(C)
    try:
        moved = 0
        for b in range(B):
            ref = st[b:b + 1].copy(); laps = np.zeros(1, np.int32); pp = np.zeros(1)
            uh, _ = orc80.skidpad_run(cfg, ref, laps, pp, float(st[b, 4]), nsteps)
            err = np.abs(us[:, b] - uh[:, 0])
            assert (err <= 10 * _utol(uh[:, 0])).all(), (b, int(np.argmax(err)), err.max())
            np.testing.assert_allclose(xs[-1, b], ref[0], rtol=0, atol=1e-7)       # the plant state after the last step
            moved += int(np.abs(us[:, b] - us0[:, b]).max() > 1e-3)
    finally:
        orc80.set_terminal_set(None)
    assert moved >= 2, moved           # the set changed the applied inputs of several scenarios


def test_terminal_set_argument_checks():
    from mpc_amd.controllers import LOcp
    C = np.zeros((3, 5)); C[:, 4] = 1.0
    with pytest.raises(RuntimeError, match="MPC_LAYOUT_LANE"):
        LOcp(50, 0.5, batch=4, solver_layout="wave", terminal_set=C)
    w = LOcp(50, 0.5, batch=4, solver_layout="wave")
    with pytest.raises(RuntimeError, match="MPC_LAYOUT_LANE"):
        w.set_terminal_set(C)
    ctl = LOcp(50, 0.5, batch=4, solver_layout="lane")
    with pytest.raises(RuntimeError, match="MPC_MAX_TERMINAL_ROWS"):
        ctl.set_terminal_set(np.zeros((257, 5)))
    with pytest.raises(ValueError):
        ctl.set_terminal_set(np.zeros((3, 4)))
    ctl.set_terminal_set(C); ctl.set_terminal_set(None)          # set, then removed


def test_unreachable_terminal_set_is_reported(orc80):
    """A terminal set the input bounds cannot reach leaves the problem without a minimiser: status 2 on both sides."""
    import torch
    from mpc_amd.controllers import LOcp
    N, Tf, B = 50, 0.5, 4
    cfg, x0, wps, ps = _cases(orc80, "L", N, Tf, B, 3)
    C = np.array([[1e4, 0, 0, 0, 0], [-1e4, 0, 0, 0, 0]])        # |y_N - y_ref| <= 1e-4
    ctl = LOcp(N, Tf, batch=B, terminal_set=C)
    _, _, status, _ = ctl.solve_batched(x0, wps, ps)
    torch.cuda.synchronize()
    orc80.set_terminal_set(C)
    try:
        ref = [orc80.optimize(cfg, x0[b], wps[b], ps[b])[0] for b in range(B)]
    finally:
        orc80.set_terminal_set(None)
    assert ref == [2] * B and (status.cpu().numpy() == 2).all()


def test_auto_layout_handle_moves_to_the_stage_wise_solver(orc64):
    """A controller whose layout was left to the library (it starts on the wavefront-per-scenario kernels) takes a
    terminal set later on: mpc_set_terminal_set moves the handle to the lane-per-scenario solver, and its answers are
    those of a controller created there."""
    import torch
    from mpc_amd.controllers import LOcp
    N, Tf, B = 50, 0.5, 8
    rng = np.random.default_rng(4)
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([np.full(B, 15.0), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.05, 0.05, B)], axis=1)
    wps, ps = [], []
    for s_ in st:
        wp, sp, _, _ = orc64.skidpad_waypoints(s_[0], s_[1], np.arctan2(s_[3], s_[2]), 0, 0.0, s_[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]; wps = np.array(wps); ps = np.array(ps)
    a = LOcp(N, Tf, batch=B)                                   # auto: wave layout
    U0, traj0, _, _ = a.solve_batched(x0, wps, ps)
    torch.cuda.synchronize()
    err = np.array([_terminal_error("L", traj0[b].cpu().numpy(), wps[b]) for b in range(B)])
    d = np.array([0.0, 1.0, 0.3, -0.2, 0.5])
    C = (d / (0.85 * np.abs(err @ d).max()))[None, :] * np.array([[1.0], [-1.0]])
    a.set_terminal_set(C)
    b = LOcp(N, Tf, batch=B, solver_layout="lane", terminal_set=C)
    Ua, ta, sa, _ = a.solve_batched(x0, wps, ps)
    Ub, tb, sb, _ = b.solve_batched(x0, wps, ps)
    torch.cuda.synchronize()
    assert torch.equal(Ua, Ub) and torch.equal(ta, tb) and torch.equal(sa, sb)
    assert not torch.equal(Ua, U0)                             # the rows are active somewhere


def test_batched_linear_programmes_match_highs():
    """mpc_lp_maximize against scipy's HiGHS on random bounded programmes (1 to 6 unknowns, 8 to 600 rows, the origin
    inside or outside the feasible set) and on an unbounded one."""
    from scipy.optimize import linprog
    from mpc_amd.stability import lp_maximize_batched
    rng = np.random.default_rng(17)
    for d, m in [(1, 8), (2, 30), (3, 64), (5, 84), (5, 600), (6, 257)]:
        A = rng.standard_normal((m, d))
        x_in = rng.standard_normal(d) * (0.0 if m % 2 else 1.0)              # a point strictly inside
        b = A @ x_in + rng.uniform(0.1, 2.0, m)
        A = np.vstack([A, np.eye(d), -np.eye(d)]); b = np.concatenate([b, x_in + 50.0, 50.0 - x_in])   # bounded
        C = rng.standard_normal((24, d))
        opt, xs, st = lp_maximize_batched(A, b, C)
        assert (st == 0).all(), (d, m, st)
        for q in range(C.shape[0]):
            r = linprog(-C[q], A_ub=A, b_ub=b, bounds=(None, None), method="highs")
            assert r.status == 0
            assert opt[q] == pytest.approx(-r.fun, rel=1e-8, abs=1e-8), (d, m, q)
            assert (A @ xs[q] <= b + 1e-8).all() and C[q] @ xs[q] == pytest.approx(opt[q], rel=1e-12, abs=1e-12)
    A = np.array([[1.0, 0.0], [-1.0, 0.0], [0.0, -1.0]]); b = np.ones(3)           # open towards +y
    opt, xs, st = lp_maximize_batched(A, b, np.array([[0.0, 1.0], [1.0, 0.0]]))
    assert st[0] == 1 and st[1] == 0 and opt[1] == pytest.approx(1.0, abs=1e-9)


def test_invariant_set_loop_on_the_gpu_matches_checker_loop():
    """estimate_control_amissible_invariant_set (stability.py:7-90) with its linear programmes on the GPU.  For a gain of
    ordinary size the loop stops at the same index with the same rows as the checker's loop with scipy's LP solver.  With
    the reference's own gain (LMPC2.py:433-449: R' = 0 makes K ~ 1e4, the rows' condition number 1e17) the outcome of the
    loop is decided by which programmes the LP solver gives up on (the reference skips those, stability.py:43-76): the
    optima are compared where both solvers finish, and the rows LOcp.terminal_set_estimate() hands out are checked for
    what they are, the set whose points keep the state and input bounds over the next n steps."""
    from scipy.optimize import linprog
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp
    from mpc_amd import stability as S
    ctl = LOcp(50, 0.5)
    A, B, K = ctl.A_stability.astype(np.float64), ctl.B_stability.astype(np.float64).reshape(5, 1), np.atleast_2d(ctl.K)
    ubx = np.array([100, 100, 100, 100, ctl.max_steering]); ubu = np.array([ctl.max_steering_rate])
    import scipy.linalg
    P2 = scipy.linalg.solve_discrete_are(A, B, np.diag([1e3, 1.0, 0.1, 0.1, 1.0]), np.array([[1.0]]))
    K2 = np.linalg.solve(B.T @ P2 @ B + 1.0, B.T @ P2 @ A)
    ubx2 = np.array([2.0, 0.5, 3.0, 3.0, 0.3])
    C2_ref, n2, _ = O.estimate_invariant_set(K2, A, B, ubx2, ubu, n_limit=200)
    C2 = S.estimate_control_amissible_invariant_set(K2, A, B, ubx2, ubu, n_limit=200)
    assert C2.shape == C2_ref.shape and np.array_equal(C2, C2_ref) and n2 > 3
    # the reference's gain: first two rounds of programmes, which both solvers finish
    phi = A - B @ K; p1 = phi.copy(); rows, rhs = [], []
    for n in (2, 3):
        rows += [p1, -p1, K @ p1, -(K @ p1)]; rhs += [ubx, ubx, ubu, ubu]
        p1 = p1 @ phi
        Cn = np.array([p1[j] / ubx[j] for j in range(5)] + [(K @ p1)[0] / ubu[0]])
        Am, bm = np.concatenate(rows), np.concatenate(rhs)
        opt, xs, st = S.lp_maximize_batched(Am, bm, Cn)
        assert (st == 0).all()
        for q in range(6):
            r = linprog(-Cn[q], A_ub=Am, b_ub=bm, bounds=(None, None), method="highs")
            assert r.status == 0 and opt[q] == pytest.approx(-r.fun, rel=1e-5, abs=1e-9), (n, q)
    C = ctl.terminal_set_estimate()
    assert C is ctl.terminal_set_estimate()                  # computed once
    n = C.shape[0] // 12 - 1
    assert C.shape[1] == 5 and C.shape[0] % 12 == 0 and n >= 2
    np.testing.assert_array_equal(C, S.reconstruct_control_amissible_invariant_set(n, K, A, B, ubx, ubu))
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((200, 5)) * np.array([0.5, 0.1, 1.0, 1.0, 0.05])
    pts *= (rng.uniform(0.2, 1.0, 200) / (pts @ C.T).max(axis=1))[:, None]
    x = pts.T
    for _ in range(min(n, 8)):
        assert (np.abs(x) <= ubx[:, None] * (1 + 1e-9)).all() and (np.abs(K @ x) <= ubu[:, None] * (1 + 1e-9)).all()
        x = phi @ x
