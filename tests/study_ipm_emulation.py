"""numpy (float64) emulation of the library's condensed interior point (mpc_ipm.hpp: interior_point) on a dumped
closed-loop step of the LPV workload, for studying its iterates on the CPU.  Development tool, not product code.

  python tests/study_ipm_emulation.py gpurun_out/lpv_fail/cold_0.npz      (uses the checker: lives with the tests)
"""
import sys

import numpy as np

sys.path.insert(0, '.')
from oracle import oracle as O


def lpv_qp(cfg, state8, lap, pp, lin_x, orc):
    """(H, g, up_d, lo_d) of the LPV step problem as the kernels condense it (mpc_wave.hpp: stage_model)."""
    N, dt = int(cfg[1]), cfg[2]
    mp = cfg[3:9]; m, Iz, lf, lr, Cf, Cr = mp
    vx = state8[4]
    wp, sp, _, _ = orc.skidpad_waypoints(state8[0], state8[1], np.arctan2(state8[3], state8[2]), int(lap), float(pp), vx, N, dt)
    wq = dt * cfg[11:17]; wr = dt * cfg[17]; QN = cfg[18:54].reshape(6, 6)
    x = np.array([0, 0, 0, state8[5], state8[6], state8[7]])
    G = np.zeros((6, N)); H = wr * np.eye(N); g = np.zeros(N)
    eps = cfg[54]
    for k in range(N):
        v = sp[k]
        a11 = -(Cf + Cr) / (m * v + eps); a12 = (-v + (Cr * lr - Cf * lf)) / (m * v + eps)
        a21 = (lr * Cr - lf * Cf) / (Iz * v + eps); a22 = -(lf * lf * Cf + lr * lr * Cr) / (Iz * v + eps)
        psib = lin_x[2, k + 1] if cfg[55] else 0.0
        vyb = lin_x[3, k + 1]
        sn, co = np.sin(psib), np.cos(psib)
        J02, J03, J12, J13 = -v * sn - vyb * co, -sn, v * co - vyb * sn, co
        A = np.eye(6)
        A[0, 2] = dt * J02; A[0, 3] = dt * J03; A[1, 2] = dt * J12; A[1, 3] = dt * J13; A[2, 4] = dt
        A[3, 3] = 1 + dt * a11; A[3, 4] = dt * a12; A[3, 5] = dt * (-Cf / m)
        A[4, 3] = dt * a21; A[4, 4] = 1 + dt * a22; A[4, 5] = dt * (-lf * Cf / Iz)
        c = np.zeros(6)
        c[0] = dt * ((v * co - vyb * sn) - (J02 * psib + J03 * vyb)); c[1] = dt * ((v * sn + vyb * co) - (J12 * psib + J13 * vyb))
        x = A @ x + c
        G = A @ G; G[5, k] = dt
        ref = np.array([wp[k + 1, 0], wp[k + 1, 1], wp[k + 1, 3], 0, 0, 0])
        if k + 1 < N:
            H += G.T @ (wq[:, None] * G); g += G.T @ (wq * (x - ref))
        else:
            H += G.T @ QN @ G; g += G.T @ QN @ x          # LPV: terminal reference stays zero
    d0, dmax, umax = state8[7], cfg[9], cfg[10]
    reach = 0.98 * umax * dt * (np.arange(N) + 1)
    up = np.maximum(dmax, d0 - reach) - d0; lo = np.minimum(-dmax, d0 + reach) - d0
    return H, g, up, lo


def ipm(H, g, up_d, lo_d, dt, umax, dmax, tol_mu=1e-15, tol_res=1e-12, max_iter=80, safe_after=25, verbose=True, variant=0):
    N = len(g)
    vd_mask = np.arange(N) < N - 1
    T = np.tril(np.ones((N, N))) * dt
    gscale = max(1.0, np.abs(g).max()); lam0 = max(1.0, 0.01 * gscale)
    U = np.zeros(N)
    tpu = np.full(N, umax); tmu = np.full(N, umax); lpu = np.full(N, lam0); lmu = np.full(N, lam0)
    tpd = np.maximum(up_d, 0.1 * dmax); tmd = np.maximum(-lo_d, 0.1 * dmax); lpd = np.full(N, lam0); lmd = np.full(N, lam0)
    tpd[~vd_mask] = 1; tmd[~vd_mask] = 1; lpd[~vd_mask] = 0; lmd[~vd_mask] = 0
    inv_nc = 1.0 / (2.0 * (2 * N - 1))
    res_prev, status, iters = 1e300, 0, 0
    while True:
        hu = H @ U
        vd = T @ U
        lamd = np.where(vd_mask, lpd - lmd, 0.0)
        rd = hu + g + (lpu - lmu) + T.T @ lamd
        rppu = U + tpu - umax; rpmu = -U + tmu - umax
        rppd = np.where(vd_mask, vd + tpd - up_d, 0.0); rpmd = np.where(vd_mask, -vd + tmd + lo_d, 0.0)
        mu = (tpu * lpu + tmu * lmu + tpd * lpd + tmd * lmd).sum() * inv_nc
        resd = np.abs(rd).max(); resp = max(np.abs(rppu).max(), np.abs(rpmu).max(), np.abs(rppd).max(), np.abs(rpmd).max())
        res = max(resd / gscale, resp)
        if verbose:
            print(f"it {iters:3d} mu {mu:.3e} resd/gs {resd / gscale:.3e} resp {resp:.3e}", end="")
        if mu <= tol_mu and (res <= tol_res or res >= 0.5 * res_prev):
            break
        if mu <= 1e-3 * tol_mu:
            status = 0 if res <= 1e3 * tol_res else 2
            break
        if not np.isfinite(mu):
            status = 2; break
        if iters >= max_iter:
            status = 2; break
        res_prev = res
        Du = lpu / tpu + lmu / tmu
        Dd = np.where(vd_mask, lpd / tpd + lmd / tmd, 0.0)
        K = H + np.diag(Du) + T.T @ np.diag(Dd) @ T
        Lc = np.linalg.cholesky(K)
        safe = iters >= safe_after
        sigmu = 0.25 * mu if safe else 0.0
        dtpu = dtmu = dlpu = dlmu = dtpd = dtmd = dlpd = dlmd = np.zeros(N)
        for ps in range(2):
            rcpu, rcmu, rcpd, rcmd = tpu * lpu, tmu * lmu, tpd * lpd, tmd * lmd
            if safe:
                rcpu, rcmu, rcpd, rcmd = rcpu - sigmu, rcmu - sigmu, rcpd - sigmu, rcmd - sigmu
            elif ps:
                rcpu = rcpu + dtpu * dlpu - sigmu; rcmu = rcmu + dtmu * dlmu - sigmu
                rcpd = rcpd + dtpd * dlpd - sigmu; rcmd = rcmd + dtmd * dlmd - sigmu
            wu = (lpu * rppu - rcpu) / tpu - (lmu * rpmu - rcmu) / tmu
            wd = np.where(vd_mask, (lpd * rppd - rcpd) / tpd - (lmd * rpmd - rcmd) / tmd, 0.0)
            rhs = -rd - wu - T.T @ wd
            dU = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
            dvd = T @ dU
            dtpu = -rppu - dU; dtmu = -rpmu + dU
            dlpu = -(rcpu + lpu * dtpu) / tpu; dlmu = -(rcmu + lmu * dtmu) / tmu
            dtpd = np.where(vd_mask, -rppd - dvd, 0.0); dtmd = np.where(vd_mask, -rpmd + dvd, 0.0)
            dlpd = np.where(vd_mask, -(rcpd + lpd * dtpd) / tpd, 0.0); dlmd = np.where(vd_mask, -(rcmd + lmd * dtmd) / tmd, 0.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                rmax = max((-dtpu / tpu).max(), (-dtmu / tmu).max(), (-dlpu / lpu).max(), (-dlmu / lmu).max(),
                           (-dtpd / tpd)[vd_mask].max(), (-dtmd / tmd)[vd_mask].max(), (-dlpd / lpd)[vd_mask].max(), (-dlmd / lmd)[vd_mask].max())
            alpha = 1.0 / rmax if rmax > 1.0 else 1.0
            if safe:
                break
            if ps == 0:
                mua = ((tpu + alpha * dtpu) * (lpu + alpha * dlpu) + (tmu + alpha * dtmu) * (lmu + alpha * dlmu) +
                       (tpd + alpha * dtpd) * (lpd + alpha * dlpd) + (tmd + alpha * dtmd) * (lmd + alpha * dlmd)).sum() * inv_nc
                sg = mua / mu
                if variant == 0:
                    sigmu = max(sg ** 3 * mu, 1e-3 * min(mu, res))
                else:
                    sigmu = max(sg ** 3 * mu, 1e-3 * min(mu, res), min(mu, 0.1 * res * gscale * 0 + mu * min(1.0, 10 * res)))
        if not safe:
            alpha *= min(max(0.995, 1.0 - mu), 0.999999)
        else:
            prods = np.concatenate([tpu * lpu, tmu * lmu, (tpd * lpd)[vd_mask], (tmd * lmd)[vd_mask]])
            gam = min(0.01, 0.5 * prods.min() / mu)
            alpha *= 0.99
            for bt in range(20):
                p = np.concatenate([(tpu + alpha * dtpu) * (lpu + alpha * dlpu), (tmu + alpha * dtmu) * (lmu + alpha * dlmu),
                                    ((tpd + alpha * dtpd) * (lpd + alpha * dlpd))[vd_mask], ((tmd + alpha * dtmd) * (lmd + alpha * dlmd))[vd_mask]])
                if p.min() >= gam * p.sum() * inv_nc:
                    break
                alpha *= 0.6
        if verbose:
            print(f" | alpha {alpha:.4f} sigmu {sigmu:.2e} {'safe' if safe else ''}")
        U = U + alpha * dU
        tpu = tpu + alpha * dtpu; tmu = tmu + alpha * dtmu; lpu = lpu + alpha * dlpu; lmu = lmu + alpha * dlmu
        tpd = tpd + alpha * dtpd; tmd = tmd + alpha * dtmd; lpd = lpd + alpha * dlpd; lmd = lmd + alpha * dlmd
        iters += 1
    if verbose:
        print(f"\nstatus {status} iters {iters}")
    return U, status, iters, mu, res


if __name__ == "__main__":
    orc = O.Oracle("f64")
    cfg = O.make_cfg("LPV", 80, 0.8)
    for f in sys.argv[1:]:
        d = np.load(f)
        H, g, up, lo = lpv_qp(cfg, d["state"], d["lap"], d["pp"], d["lin_x"], orc)
        print(f, "cond(H) %.2e" % np.linalg.cond(H), "u0 gpu", float(d["u0"]))
        U, st, it, mu, res = ipm(H, g, up, lo, cfg[2], cfg[10], cfg[9])
        lx = np.ascontiguousarray(d["lin_x"]).copy(); lu = np.ascontiguousarray(d["lin_u"]).copy()
        s8 = d["state"]
        wp, sp, _, _ = orc.skidpad_waypoints(s8[0], s8[1], np.arctan2(s8[3], s8[2]), int(d["lap"]), float(d["pp"]), s8[4], 80, cfg[2])
        stt, traj, Uref, lam, info = O.Oracle("f80").optimize(cfg, s8[[0, 1, 2, 3, 5, 6, 7]], wp, sp, 0.0, lx, lu)
        print("emul u0", U[0], "checker u0", Uref[0], "max|dU|", np.abs(U - Uref).max())
