"""The allocation-free entry point mpc_solve_only_noalloc (static device scratch array instead of a handle's workspace)
against mpc_solve_only on a handle, through the C ABI: same kernels, so the results must agree bit for bit; two calls in a
row, the second with a smaller batch, so that what the first call left in the shared array is seen not to matter."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(orc, rng, B, N, Tf):
    psi = rng.uniform(-0.1, 0.1, B)
    st = np.stack([np.full(B, 15.0), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
                   rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.05, 0.05, B)], axis=1)
    st[0] = [12.0, 0.0, 1, 0, 8.0, 0, 0, 0.0]                       # on the line: no bound active
    st[1] = [15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28]                     # steering bound active
    wps, ps = [], []
    for s in st:
        wp, sp, _, _ = orc.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
        wps.append(wp); ps.append(sp)
    return st[:, [0, 1, 2, 3, 5, 6, 7]].copy(), np.array(wps), np.array(ps)


@pytest.mark.parametrize("kind,N,Tf", [("L", 50, 0.5), ("OFL", 64, 0.64)])
def test_noalloc_solve_equals_handle_solve(orc64, kind, N, Tf):
    import torch
    from mpc_amd import _dev, _lib
    from mpc_amd.controllers import LOcp, OFLOcp
    lib = _lib.load()
    rng = np.random.default_rng(23)
    ctl = (LOcp if kind == "L" else OFLOcp)(N, Tf, batch=64)       # wave layout, no fixed-model cache
    dev = ctl.device
    for B in (64, 23):                                              # second call: smaller batch, other problems
        x0, wps, ps = _problems(orc64, rng, B, N, Tf)
        dh = rng.uniform(-12, -8, B) if kind == "OFL" else None
        U_ref, traj_ref, st_ref, it_ref = ctl.solve_batched(x0, wps, ps, dh)
        x0d, wpd, pd = _dev.f64(x0, dev), _dev.f64(wps, dev), _dev.f64(ps, dev)
        dhd = _dev.f64(dh, dev) if dh is not None else None
        U = torch.full((B, N), float("nan"), dtype=torch.float64, device=dev)
        traj = torch.full((B, N + 1, 7), float("nan"), dtype=torch.float64, device=dev)
        status = torch.full((B,), -1, dtype=torch.int32, device=dev); iters = torch.full((B,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(dev)[0]
        cfg_nob = C.pointer(type(ctl.solver.cfg).from_buffer_copy(ctl.solver.cfg))
        cfg_nob.contents.B = 0                                       # the handle's capacity is not read by this entry point
        rc = lib.mpc_solve_only_noalloc(cfg_nob, _dev.stream_ptr(dev), B, _dev.ptr(x0d), _dev.ptr(wpd), _dev.ptr(pd),
                                        _dev.ptr(dhd), _dev.ptr(U), _dev.ptr(traj), _dev.ptr(status), _dev.ptr(iters))
        _lib.check(rc, None, "mpc_solve_only_noalloc")
        torch.cuda.synchronize()
        if B != 64:                                                  # past the first call's one-time address look-up:
            assert torch.cuda.mem_get_info(dev)[0] == free_before    # the device's free memory did not move
        assert torch.equal(U, U_ref) and torch.equal(traj, traj_ref)
        assert torch.equal(status, st_ref) and torch.equal(iters, it_ref)
        assert int((iters > 0).sum()) >= 1                                       # the interior point ran ...
        if kind == "L":
            assert int((iters == 0).sum()) >= 1                                  # ... and so did the unconstrained shortcut
        # and against the checker, to the input bar
        from oracle import oracle as O
        cfg = O.make_cfg(kind, N, Tf)
        for b in (0, 1, B - 1):
            _, _, U_chk, _, _ = orc64.optimize(cfg, x0[b], wps[b], ps[b], 0.0 if dh is None else dh[b])
            assert (np.abs(U[b].cpu().numpy() - U_chk) <= 1e-5 * np.maximum(np.abs(U_chk), 1e-3 * 0.9)).all()


def test_noalloc_solve_refuses_what_does_not_fit():
    import torch
    from mpc_amd import _dev, _lib
    from mpc_amd.controllers import LOcp, LPVOcp
    lib = _lib.load()
    ctl = LOcp(50, 0.5, batch=1)
    dev = ctl.device
    z = torch.zeros(65 * 51 * 7, dtype=torch.float64, device=dev)
    st = torch.zeros(65, dtype=torch.int32, device=dev)
    args = (_dev.stream_ptr(dev), 65, _dev.ptr(z), _dev.ptr(z), _dev.ptr(z), None, _dev.ptr(z), None, _dev.ptr(st), _dev.ptr(st))
    assert lib.mpc_solve_only_noalloc(C.byref(ctl.solver.cfg), *args) == -1                   # MPC_EINVAL: 65 scenarios
    assert b"outside 1..64" in lib.mpc_last_error(None)
    big = LOcp(100, 1.0, batch=1)
    assert lib.mpc_solve_only_noalloc(C.byref(big.solver.cfg), args[0], 4, *args[2:]) == -1   # horizon without a register kernel
    assert b"horizon" in lib.mpc_last_error(None)
    lpv = LPVOcp(50, 0.5, batch=1)
    assert lib.mpc_solve_only_noalloc(C.byref(lpv.solver.cfg), args[0], 4, *args[2:]) == -1
    assert b"kinds L and OFL" in lib.mpc_last_error(None)
