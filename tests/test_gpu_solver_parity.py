"""GPU parity of the QP solve (mpc_solve_only through the C ABI) against the CPU checker.

Bar (north-star): control inputs within 1e-5 relative per step.  The test states it as
|u_gpu - u_ref| <= 1e-5 * max(|u_ref|, 1e-3 * u_max) for the applied input u_0 and for every entry of the
predicted input sequence; the reference value is the 80-bit checker driven to the unique minimiser."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenarios(rng, B, vx=8.0):
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([np.full(B, 15.0), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, vx),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.05, 0.05, B)], axis=1)
    return st


def _tol(u_ref, umax=0.9):
    return 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)


# horizons 10, 20, 30, 40, 50 and 64 run register-resident; 54, 80, 100 and 128 take the any-horizon kernel
# (one and two stages per lane, and its largest size)
@pytest.mark.parametrize("kind,N,Tf", [("L", 50, 0.5), ("L", 10, 0.5), ("OFL", 50, 0.5), ("L", 64, 0.64),
                                       ("L", 54, 0.54), ("OFL", 80, 0.8), ("L", 100, 1.0), ("L", 128, 1.28),
                                       ("L", 7, 0.07), ("L", 2, 0.02), ("OFL", 3, 0.03),
                                       ("L", 20, 0.2), ("OFL", 30, 0.3), ("L", 40, 0.4)])
def test_solve_only_matches_checker(orc80, kind, N, Tf):
    import torch
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, OFLOcp
    rng = np.random.default_rng(7)
    B = 48 if N <= 64 else 24
    st = _scenarios(rng, B)
    st[0] = [15.0, 0.1, 1, 0, 8.0, 0, 0, 0]
    st[1] = [15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28]       # steering bound becomes active
    st[2] = [15.0, 0.3, 1, 0, 8.0, 0, 0, 0.28]
    st[3] = [12.0, 0.0, 1, 0, 8.0, 0, 0, 0.0]          # on the line: unconstrained
    cfg = O.make_cfg(kind, N, Tf)
    ctl = (LOcp if kind == "L" else OFLOcp)(N, Tf, batch=B)
    x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc80.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    wps = np.array(wps); ps = np.array(ps)
    dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
    refs = [orc80.optimize(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b]) for b in range(B)]
    # per-scenario condensing first, then the fixed-model cache (every scenario runs at vx = 8)
    for cached in (False, True):
        ctl.condense_fixed(8.0 if cached else None)
        U, traj, status, iters = ctl.solve_batched(x0, wps, ps, dh)
        torch.cuda.synchronize()
        U = U.cpu().numpy(); traj = traj.cpu().numpy(); status = status.cpu().numpy(); iters = iters.cpu().numpy()
        assert (status == 0).all()
        nconstrained = 0
        for b in range(B):
            stt, traj_ref, U_ref, lam, info = refs[b]
            assert stt == 0
            nconstrained += int(np.abs(lam).max() > 1e-9)
            err = np.abs(U[b] - U_ref)
            assert (err <= _tol(U_ref)).all(), (cached, b, err.max(), iters[b])
            np.testing.assert_allclose(traj[b], traj_ref, rtol=1e-7, atol=1e-8)
        assert nconstrained >= 3        # the active-bound branch was exercised
        if kind == "L" and Tf <= 0.55:   # longer previews reach the first circle, where bounds bite
            assert iters[3] == 0        # and so was the unconstrained shortcut


@pytest.mark.parametrize("cached,N,Tf", [(False, 50, 0.5), (True, 50, 0.5), (False, 72, 0.72)])
def test_closed_loop_steps_match_checker(orc80, cached, N, Tf):
    """40 fused closed-loop steps (plan -> solve -> RK4 -> lap) against the checker's own loop."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    rng = np.random.default_rng(11)
    B, nsteps = 32, 40
    st = _scenarios(rng, B)
    cfg = O.make_cfg("L", N, Tf)
    sim = SkidpadBatch(N, Tf, st, model="L", fixed_model_cache=cached)
    ref_states = st.copy(); laps = np.zeros(B, np.int32); pp = np.zeros(B)
    uh_ref, _ = orc80.skidpad_run(cfg, ref_states, laps, pp, 8.0, nsteps)
    us = []
    for t in range(nsteps):
        us.append(sim.step().cpu().numpy().copy())
    torch.cuda.synchronize()
    us = np.array(us)
    # per-step inputs: the closed loop feeds differences back, so compare step by step with the stated bar
    assert (np.abs(us - uh_ref) <= 10 * _tol(uh_ref)).all(), np.abs(us - uh_ref).max()
    np.testing.assert_allclose(sim.state.cpu().numpy(), ref_states, rtol=0, atol=1e-7)
    assert (sim.lap.cpu().numpy() == laps).all()


@pytest.mark.parametrize("N,Tf", [(50, 0.5), (90, 0.9)])
def test_fused_run_equals_step_by_step(N, Tf):
    """mpc_closed_loop_run (all steps in one launch) reproduces one-launch-per-step bit for bit."""
    import torch
    from mpc_amd.batched import SkidpadBatch
    rng = np.random.default_rng(5)
    B, nsteps = 64, 60
    st = _scenarios(rng, B)
    a = SkidpadBatch(N, Tf, st, model="L")
    b = SkidpadBatch(N, Tf, st, model="L")
    us_a = torch.stack([a.step().clone() for _ in range(nsteps)])
    xs_b, us_b = b.run(nsteps, record=True)
    torch.cuda.synchronize()
    assert torch.equal(us_a, us_b)
    assert torch.equal(a.state, b.state) and torch.equal(a.state, xs_b[-1])
    assert torch.equal(a.lap, b.lap) and torch.equal(a.prev_progress, b.prev_progress)


@pytest.mark.parametrize("model,N,Tf", [("L", 50, 0.5), ("LPV", 20, 0.4), ("NL", 20, 0.4)])
def test_chained_rebalanced_run_equals_step_by_step(model, N, Tf):
    """Runs of a large batch map waves to scenarios in the order of the solver work of the previous launch, and a run
    longer than 1024 steps goes out as a chain of launches: every scenario's inputs, states, totals and statuses equal the
    one-launch-per-step run bit for bit."""
    import torch
    from mpc_amd.batched import SkidpadBatch
    rng = np.random.default_rng(8)
    B, nsteps = 1100, 150                      # >= 1024 scenarios: the second run below is ordered by the work of this one
    st = _scenarios(rng, B)
    st[::3, 1] *= 4.0                          # a third of the scenarios start far off the line: uneven solver work
    a = SkidpadBatch(N, Tf, st, model=model)
    b = SkidpadBatch(N, Tf, st, model=model)
    us_a, it_a, st_a = [], torch.zeros(B, dtype=torch.int64, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(nsteps):
        us_a.append(a.step().clone()); it_a += a.iters; st_a = torch.maximum(st_a, a.status)
    us_a = torch.stack(us_a)
    xs_b, us_b = b.run(nsteps, record=True)
    torch.cuda.synchronize()
    assert torch.equal(us_a, us_b)
    assert torch.equal(a.state, b.state) and torch.equal(a.state, xs_b[-1])
    assert torch.equal(a.lap, b.lap) and torch.equal(a.prev_progress, b.prev_progress)
    assert torch.equal(it_a, b.iters.to(torch.int64)) and torch.equal(st_a, b.status)
    if model == "L":       # the interior-point work is uneven, so the order is not the identity (the LPV and NL kernels
        assert int(it_a.max()) > 4 * max(1, int(it_a.min()))      # finish nearly every QP in the active-set attempt)
    # the second call starts from the order the first one left
    n2 = 1100 if model == "L" else 130         # L: longer than one chunk, so a chain of two launches
    us_a2 = torch.stack([a.step().clone() for _ in range(n2)])
    xs_b2, us_b2 = b.run(n2, record=True)
    torch.cuda.synchronize()
    assert torch.equal(us_a2, us_b2) and torch.equal(a.state, b.state)


def test_out_of_reach_steering_start_is_relaxed_and_flagged(orc80):
    """A start steering angle beyond the bound by more than one step of full rate makes the reference's QP infeasible
    (acados raises there).  Library and checker both solve the problem with the bound widened along the way back inside
    and flag it with status 4; angles that one step can still bring back are solved as posed (status 0)."""
    import torch
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, OFLOcp
    N, Tf = 50, 0.5
    deltas = [0.33, -0.36, 0.305, 0.3, -0.3087]
    for kind, cls in (("L", LOcp), ("OFL", OFLOcp)):
        cfg = O.make_cfg(kind, N, Tf)
        ctl = cls(N, Tf, batch=len(deltas))
        st = np.array([[15.0, 0.1 * (-1) ** i, 1, 0, 8.0, 0.05, -0.05, d] for i, d in enumerate(deltas)])
        x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
        wps, ps = [], []
        for s in st:
            wp, sp, _, _ = orc80.skidpad_waypoints(s[0], s[1], 0.0, 0, 0.0, s[4], N, Tf / N)
            wps.append(wp); ps.append(sp)
        dh = np.full(len(deltas), -9.0) if kind == "OFL" else None
        U, traj, status, iters = ctl.solve_batched(x0, np.array(wps), np.array(ps), dh)
        torch.cuda.synchronize()
        U = U.cpu().numpy(); status = status.cpu().numpy()
        assert status.tolist() == [4, 4, 0, 0, 0]
        for b in range(len(deltas)):
            stt, traj_ref, U_ref, lam, info = orc80.optimize(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b])
            assert stt == status[b]
            assert (np.abs(U[b] - U_ref) <= _tol(U_ref)).all(), (kind, b, np.abs(U[b] - U_ref).max())
        assert np.isfinite(U).all()


def test_lpv_steps_that_once_broke_the_interior_point(orc80, golden):
    """Closed-loop steps of the LPV N=80 workload on which the interior point used to end without a minimiser (the
    Newton matrix loses its definiteness to rounding when the complementarity passes 1e-15 with a steering-angle bound
    active; about one solve in 1e5).  The inputs of those steps were recorded on the GPU (tools/lpv_fail_scan.py); they
    must solve with status 0 to the input bar, started cold."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    d = golden["lpv_hard_steps"]
    N, Tf = 80, 0.8
    n = d["state"].shape[0]
    cfg = O.make_cfg("LPV", N, Tf)
    sim = SkidpadBatch(N, Tf, d["state"], starting_laps=d["lap"], model="LPV")
    sim.prev_progress.copy_(torch.as_tensor(d["prev_progress"]))
    sim.lin_x.copy_(torch.as_tensor(d["lin_x"])); sim.lin_u.copy_(torch.as_tensor(d["lin_u"]))
    u0 = sim.step().cpu().numpy()
    status = sim.status.cpu().numpy(); iters = sim.iters.cpu().numpy()
    assert (status == 0).all(), status
    assert (iters > 0).all()                                            # every one of them is a constrained solve
    for b in range(n):
        lx = np.ascontiguousarray(d["lin_x"][b]).copy(); lu = np.ascontiguousarray(d["lin_u"][b]).copy()
        st, _, _, _, u_ref, _ = orc80.skidpad_step(cfg, d["state"][b], int(d["lap"][b]), float(d["prev_progress"][b]),
                                                   float(d["state"][b, 4]), lx, lu)
        assert st == 0
        assert abs(u0[b] - u_ref) <= _tol(np.array([u_ref]))[0], (b, u0[b], u_ref, iters[b])


@pytest.mark.parametrize("kind,N,Tf,px_weight", [("L", 50, 0.5, 0.0), ("OFL", 80, 0.8, 0.0), ("L", 7, 0.07, 0.0), ("L", 128, 1.28, 0.0),
                                                   ("L", 50, 0.5, 2.0), ("OFL", 30, 0.3, 0.5)])
def test_lane_layout_solve_matches_checker(orc80, kind, N, Tf, px_weight):
    """The lane-per-scenario kernel (stage-wise Riccati interior point, no condensing) on the L / OFL problems, against
    the checker: a second GPU algorithm for the same minimiser, next to the condensed wave-per-scenario kernels.  With
    the parameter files' weights nothing weighs the longitudinal position and the solver build without it runs; a
    weight on it (px_weight, entry Q[0][0] of the parameter set) takes the general build."""
    import copy
    import torch
    from oracle import oracle as O
    from mpc_amd.controllers import LOcp, OFLOcp, read_parameters
    params = copy.deepcopy(read_parameters("parameters_L.yaml"))
    params["controller"]["Q"][0][0] = px_weight
    rng = np.random.default_rng(17)
    B = 40
    st = _scenarios(rng, B)
    st[1] = [15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28]       # steering bound becomes active
    st[2] = [15.0, 0.3, 1, 0, 8.0, 0, 0, 0.28]
    st[3] = [12.0, 0.0, 1, 0, 8.0, 0, 0, 0.0]          # on the line: unconstrained
    st[4] = [15.0, 0.2, 1, 0, 8.0, 0, 0, 0.32]         # start outside the steering bound: status 4 on both sides
    cfg = O.make_cfg(kind, N, Tf, params=params)
    ctl = (LOcp if kind == "L" else OFLOcp)(N, Tf, batch=B, solver_layout="lane", params=params)
    assert (cfg[11] != 0.0) == (px_weight != 0.0)
    x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc80.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    wps = np.array(wps); ps = np.array(ps)
    dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
    U, traj, status, iters = ctl.solve_batched(x0, wps, ps, dh)
    torch.cuda.synchronize()
    U = U.cpu().numpy(); traj = traj.cpu().numpy(); status = status.cpu().numpy(); iters = iters.cpu().numpy()
    nconstrained = 0
    for b in range(B):
        stt, traj_ref, U_ref, lam, info = orc80.optimize(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b])
        assert status[b] == stt and stt in (0, 4), (b, status[b], stt)
        nconstrained += int(np.abs(lam).max() > 1e-9)
        err = np.abs(U[b] - U_ref)
        assert (err <= _tol(U_ref)).all(), (b, err.max(), iters[b])
        np.testing.assert_allclose(traj[b], traj_ref, rtol=1e-7, atol=1e-8)
    assert nconstrained >= 3
    if kind == "L" and Tf <= 0.55:
        assert iters[3] == 0                       # the unconstrained minimiser was taken as it is


@pytest.mark.parametrize("model,layout,N,Tf", [("L", "lane", 50, 0.5), ("LPV", "wave", 40, 0.4), ("LPV", "lane", 40, 0.4)])
def test_closed_loop_in_both_layouts_matches_checker(orc80, model, layout, N, Tf):
    """40 fused closed-loop steps in the layout that is not the controller's default (L on lanes, LPV on waves), and LPV
    on lanes at a short horizon, against the checker's loop."""
    import torch
    from oracle import oracle as O
    from mpc_amd.batched import SkidpadBatch
    rng = np.random.default_rng(19)
    B, nsteps = 24, 40
    st = _scenarios(rng, B)
    cfg = O.make_cfg(model, N, Tf)
    sim = SkidpadBatch(N, Tf, st, model=model, solver_layout=layout)
    ref_states = st.copy(); laps = np.zeros(B, np.int32); pp = np.zeros(B)
    uh_ref, _ = orc80.skidpad_run(cfg, ref_states, laps, pp, 8.0, nsteps)
    xs, us = sim.run(nsteps, record=True)
    torch.cuda.synchronize()
    us = us.cpu().numpy()
    assert (sim.status.cpu().numpy() == 0).all()
    assert (np.abs(us - uh_ref) <= 10 * _tol(uh_ref)).all(), np.abs(us - uh_ref).max()
    np.testing.assert_allclose(sim.state.cpu().numpy(), ref_states, rtol=0, atol=1e-7)
    assert (sim.lap.cpu().numpy() == laps).all()
