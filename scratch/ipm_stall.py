"""numpy replica of the GPU interior-point loop on QPs where the GPU hit its iteration cap."""
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle as O

def ipm_gpu(H, g, lo_d, up_d, umax, dmax, dt, N, tol_mu=1e-15, tol_res=1e-12, maxit=60, verbose=False, variant=0):
    nd = N - 1
    U = np.zeros(N)
    tpu = np.full(N, umax); tmu = np.full(N, umax); lpu = np.ones(N); lmu = np.ones(N)
    tpd = np.full(nd, max(up_d, 0.1 * dmax)); tmd = np.full(nd, max(-lo_d, 0.1 * dmax)); lpd = np.ones(nd); lmd = np.ones(nd)
    T = np.tril(np.ones((N, N)))[:nd] * dt
    gscale = max(1.0, np.abs(g).max())
    res_prev = 1e300
    for it in range(maxit + 1):
        vd = T @ U
        rd = H @ U + g + (lpu - lmu) + T.T @ (lpd - lmd)
        rppu = U + tpu - umax; rpmu = -U + tmu - umax
        rppd = vd + tpd - up_d; rpmd = -vd + tmd + lo_d
        mu = (tpu @ lpu + tmu @ lmu + tpd @ lpd + tmd @ lmd) / (2 * (2 * N - 1))
        res = max(np.abs(rd).max() / gscale, np.abs(rppu).max(), np.abs(rpmu).max(), np.abs(rppd).max(), np.abs(rpmd).max())
        if verbose: print(it, "mu %.3e res %.3e" % (mu, res), end=' ')
        if mu <= tol_mu and (res <= tol_res or res >= 0.5 * res_prev): return U, it, 0
        if it >= maxit: return U, it, 2
        res_prev = res
        Du = lpu / tpu + lmu / tmu; Dd = lpd / tpd + lmd / tmd
        K = H + np.diag(Du) + T.T @ (Dd[:, None] * T)
        Kinv = np.linalg.inv(K)
        sigmu = 0.0
        dtpu = dtmu = dlpu = dlmu = np.zeros(N); dtpd = dtmd = dlpd = dlmd = np.zeros(nd)
        for p in range(2):
            rcpu = tpu * lpu; rcmu = tmu * lmu; rcpd = tpd * lpd; rcmd = tmd * lmd
            if p:
                rcpu = rcpu + dtpu * dlpu - sigmu; rcmu = rcmu + dtmu * dlmu - sigmu
                rcpd = rcpd + dtpd * dlpd - sigmu; rcmd = rcmd + dtmd * dlmd - sigmu
            wu = (lpu * rppu - rcpu) / tpu - (lmu * rpmu - rcmu) / tmu
            wd = (lpd * rppd - rcpd) / tpd - (lmd * rpmd - rcmd) / tmd
            dU = Kinv @ (-rd - wu - T.T @ wd)
            dvd = T @ dU
            dtpu = -rppu - dU; dtmu = -rpmu + dU
            dlpu = -(rcpu + lpu * dtpu) / tpu; dlmu = -(rcmu + lmu * dtmu) / tmu
            dtpd = -rppd - dvd; dtmd = -rpmd + dvd
            dlpd = -(rcpd + lpd * dtpd) / tpd; dlmd = -(rcmd + lmd * dtmd) / tmd
            def ms(x, dx):
                neg = dx < 0
                return min(1.0, (-x[neg] / dx[neg]).min()) if neg.any() else 1.0
            ap = min(ms(tpu, dtpu), ms(tmu, dtmu), ms(tpd, dtpd), ms(tmd, dtmd))
            ad = min(ms(lpu, dlpu), ms(lmu, dlmu), ms(lpd, dlpd), ms(lmd, dlmd))
            alpha = min(ap, ad)
            if p == 0:
                mua = ((tpu + alpha * dtpu) @ (lpu + alpha * dlpu) + (tmu + alpha * dtmu) @ (lmu + alpha * dlmu) +
                       (tpd + alpha * dtpd) @ (lpd + alpha * dlpd) + (tmd + alpha * dtmd) @ (lmd + alpha * dlmd)) / (2 * (2 * N - 1))
                sg = mua / mu
                sigmu = sg ** 3 * mu
                if variant == 1: sigmu = max(sigmu, 0.0)
        if verbose: print("alpha %.3e sig %.2e" % (alpha, sigmu / mu))
        alpha *= 0.995
        U = U + alpha * dU
        tpu = tpu + alpha * dtpu; tmu = tmu + alpha * dtmu; lpu = lpu + alpha * dlpu; lmu = lmu + alpha * dlmu
        tpd = tpd + alpha * dtpd; tmd = tmd + alpha * dtmd; lpd = lpd + alpha * dlpd; lmd = lmd + alpha * dlmd
    return U, it, 2

if __name__ == '__main__':
    orc = O.Oracle('f64')
    N, Tf = 50, 0.5; dt = Tf / N
    cfg = O.make_cfg("OFL", N, Tf)
    for i in range(3):
        xe = np.load(f"gpurun_out/bad_est_{i}.npy")
        hd = np.arctan2(xe[3], xe[2])
        wp, sp, p0, ab = orc.step_waypoints(xe[0], xe[1], hd, 15.0, N, dt)
        x0 = xe[[0, 1, 2, 3, 5, 6, 7]]
        H, g, A, lo, up = O.condensed_qp(cfg, x0, wp, sp, xe[9])
        print("case", i, "delta0", xe[7], "gmax", np.abs(g).max())
        U, it, st = ipm_gpu(H, g, lo[N], up[N], cfg[10], cfg[9], dt, N, verbose=True)
        print("-> it", it, "status", st)
        r = orc.optimize(cfg, x0, wp, sp, xe[9]); print("oracle", r[0], r[4], np.abs(r[2] - U).max())
