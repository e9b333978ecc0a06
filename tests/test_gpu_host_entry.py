"""mpc_optimize_host (host arrays in and out, one launch and one synchronisation per call: what LOcp / LPVOcp / OFLOcp /
NLOcp.optimize go through) against mpc_solve_only on device arrays: the same kernels on the same numbers, so the results
are bit-identical; and mpc_set_lpv_lin_mode against handles created with either linearisation rule."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(orc, rng, B, N, Tf):
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([15.0 + rng.uniform(0, 30, B), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.25, 0.25, B)], axis=1)
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    return np.ascontiguousarray(st[:, [0, 1, 2, 3, 5, 6, 7]]), np.array(wps), np.array(ps)


def _host_call(ctl, B, x0, wp, p, dh=None, lin_x=None, lin_u=None):
    from mpc_amd import _dev, _lib
    N = ctl.N
    U = np.full((B, N), np.nan); traj = np.full((B, N + 1, 7), np.nan)
    status = np.full(B, -1, np.int32); iters = np.full(B, -1, np.int32)
    hp = lambda a: None if a is None else a.ctypes.data
    rc = ctl.solver._lib.mpc_optimize_host(ctl.solver.h, _dev.stream_ptr(ctl.device), B, hp(x0), hp(wp), hp(p), hp(dh), hp(lin_x),
                                           hp(lin_u), hp(U), hp(traj), hp(status), hp(iters))
    _lib.check(rc, ctl.solver.h, "mpc_optimize_host")
    return U, traj, status, iters


@pytest.mark.parametrize("kind,N,Tf", [("L", 50, 0.5), ("OFL", 50, 0.5), ("LPV", 80, 0.8), ("LPV", 50, 0.5), ("NL", 50, 0.5), ("L", 25, 0.5)])
def test_host_entry_equals_device_entry(orc64, kind, N, Tf):
    import torch
    from mpc_amd import _dev
    from mpc_amd.controllers import LOcp, LPVOcp, NLOcp, OFLOcp
    cls = {"L": LOcp, "OFL": OFLOcp, "LPV": LPVOcp, "NL": NLOcp}[kind]
    rng = np.random.default_rng(31)
    B = 12
    a = cls(N, Tf, batch=B); b = cls(N, Tf, batch=B)
    for call in range(3):                                   # carried memory (LPV prev_x / prev_u, NL iterate) travels alike
        x0, wp, p = _problems(orc64, rng, B, N, Tf)
        dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
        if call == 0:
            lx_h = np.zeros((B, 6, N + 1)) if kind == "LPV" else None
            lu_h = np.zeros((B, N)) if kind in ("LPV", "NL") else None
            lx_d = None if lx_h is None else _dev.f64(lx_h, a.device).clone()
            lu_d = None if lu_h is None else _dev.f64(lu_h, a.device).clone()
        U_d, traj_d, st_d, it_d = a.solve_batched(x0, wp, p, dh, lx_d, lu_d)
        U_h, traj_h, st_h, it_h = _host_call(b, B, x0, wp, p, dh, lx_h, lu_h)
        torch.cuda.synchronize()
        assert np.array_equal(U_h, U_d.cpu().numpy()) and np.array_equal(traj_h, traj_d.cpu().numpy())
        assert np.array_equal(st_h, st_d.cpu().numpy()) and np.array_equal(it_h, it_d.cpu().numpy())
        if lx_h is not None:
            assert np.array_equal(lx_h, lx_d.cpu().numpy())
        if lu_h is not None:
            assert np.array_equal(lu_h, lu_d.cpu().numpy())
    assert (it_h > 0).any()                                 # constrained solves among them


def test_host_entry_refuses_bad_arguments():
    from mpc_amd import _dev
    from mpc_amd.controllers import LOcp, LPVOcp
    ctl = LOcp(50, 0.5, batch=2)
    z = np.zeros(3 * 51 * 7); st = np.zeros(3, np.int32)
    f = ctl.solver._lib.mpc_optimize_host
    args = [z.ctypes.data] * 3 + [None, None, None, z.ctypes.data, None, st.ctypes.data, st.ctypes.data]
    assert f(ctl.solver.h, _dev.stream_ptr(ctl.device), 3, *args) == -1                       # more than the handle's capacity
    assert b"outside 1.." in ctl.solver._lib.mpc_last_error(ctl.solver.h)
    assert f(ctl.solver.h, _dev.stream_ptr(ctl.device), 1, None, *args[1:]) == -1
    lpv = LPVOcp(50, 0.5, batch=1)
    assert f(lpv.solver.h, _dev.stream_ptr(lpv.device), 1, *args) == -1                       # LPV without lin_x / lin_u
    assert ctl.solver._lib.mpc_set_lpv_lin_mode(ctl.solver.h, 1) == -1                        # not an LPV handle


def test_lin_mode_switch_equals_a_handle_created_that_way(orc64):
    """LPVOcp.optimize(..., lin_mode=...) switches the linearisation rule of the handle it has (mpc_set_lpv_lin_mode)."""
    from mpc_amd.controllers import LPVOcp
    N, Tf = 50, 0.5
    rng = np.random.default_rng(7)
    x0, wp, p = _problems(orc64, rng, 3, N, Tf)
    a = LPVOcp(N, Tf)
    outs = []
    for i, mode in enumerate(["reference", "prev_iter", "prev_iter"]):
        outs.append(a.optimize(x0[i], wp[i], p[i], lin_mode=mode))
    b = LPVOcp(N, Tf)
    ref0 = b.optimize(x0[0], wp[0], p[0], lin_mode="reference")
    from oracle import oracle as O
    cfg = O.make_cfg("LPV", N, Tf, lin_mode="prev_iter")
    lx = b.prev_x.copy(); lu = np.ascontiguousarray(b.prev_u[0, :N]).copy()
    assert np.array_equal(outs[0][2], ref0[2])
    for i in (1, 2):
        _, traj_ref, U_ref, _, _ = orc64.optimize(cfg, x0[i], wp[i], p[i], 0.0, lx, lu)
        err = np.abs(outs[i][2][:, 0] - U_ref)
        assert (err <= 1e-5 * np.maximum(np.abs(U_ref), 1e-3 * 0.9)).all(), (i, err.max())
