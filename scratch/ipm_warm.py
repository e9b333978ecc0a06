"""Warm-started IPM over a closed-loop sequence of L-controller QPs (numpy prototype)."""
import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from oracle import oracle as O

def ipm3(H, g, lo_d, up_d, umax, dmax, dt, N, warm=None, delta=1e-2, tol_mu=1e-15, tol_res=1e-12, maxit=100, shift=True):
    nd = N - 1
    T = np.tril(np.ones((N, N)))[:nd] * dt
    A = np.vstack([np.eye(N), T])
    up = np.concatenate([np.full(N, umax), np.full(nd, up_d)]); lo = np.concatenate([np.full(N, -umax), np.full(nd, lo_d)])
    m = len(up)
    gscale = max(1.0, np.abs(g).max())
    if warm is None:
        U = np.zeros(N)
        tp = np.concatenate([np.full(N, umax), np.full(nd, max(up_d, 0.1 * dmax))])
        tm = np.concatenate([np.full(N, umax), np.full(nd, max(-lo_d, 0.1 * dmax))])
        l0 = max(1.0, gscale / 100)
        lp = np.full(m, l0); lm = np.full(m, l0)
    else:
        U, lp, lm = (w.copy() for w in warm)
        if shift:
            U = np.concatenate([U[1:], U[-1:]])
            lp = np.concatenate([lp[1:N], lp[N - 1:N], lp[N + 1:], lp[-1:]]); lm = np.concatenate([lm[1:N], lm[N - 1:N], lm[N + 1:], lm[-1:]])
        v = A @ U
        rng_ = up - lo
        tp = np.maximum(up - v, delta * rng_); tm = np.maximum(v - lo, delta * rng_)
        # target complementarity: keep active multipliers, lift inactive ones to a common small product
        mu0 = delta * 0.01 * gscale
        lp = np.maximum(lp, mu0 / tp); lm = np.maximum(lm, mu0 / tm)
    res_prev = 1e300
    def ms(x, dx):
        neg = dx < 0
        return min(1.0, (-x[neg] / dx[neg]).min()) if neg.any() else 1.0
    for it in range(maxit + 1):
        v = A @ U
        rd = H @ U + g + A.T @ (lp - lm)
        rpp = v + tp - up; rpm = -v + tm + lo
        mu = (tp @ lp + tm @ lm) / (2 * m)
        res = max(np.abs(rd).max() / gscale, np.abs(rpp).max(), np.abs(rpm).max())
        if mu <= tol_mu and (res <= tol_res or res >= 0.5 * res_prev): return U, lp, lm, it, 0
        if mu <= 1e-3 * tol_mu: return U, lp, lm, it, 0 if res < 1e-9 else 2
        if it >= maxit: return U, lp, lm, it, 2
        res_prev = res
        D = lp / tp + lm / tm
        Kinv = np.linalg.inv(H + A.T @ (D[:, None] * A))
        def direction(rcp, rcm):
            rhs = -rd - A.T @ ((lp * rpp - rcp) / tp - (lm * rpm - rcm) / tm)
            dU = Kinv @ rhs; dv = A @ dU
            dtp = -rpp - dv; dtm = -rpm + dv
            dlp = -(rcp + lp * dtp) / tp; dlm = -(rcm + lm * dtm) / tm
            return dU, dtp, dtm, dlp, dlm
        dU, dtp, dtm, dlp, dlm = direction(tp * lp, tm * lm)
        a = min(ms(tp, dtp), ms(tm, dtm), ms(lp, dlp), ms(lm, dlm))
        mua = ((tp + a * dtp) @ (lp + a * dlp) + (tm + a * dtm) @ (lm + a * dlm)) / (2 * m)
        sm = max((mua / mu) ** 3 * mu, 1e-3 * min(mu, res))
        dU, dtp, dtm, dlp, dlm = direction(tp * lp + dtp * dlp - sm, tm * lm + dtm * dlm - sm)
        a = min(ms(tp, dtp), ms(tm, dtm), ms(lp, dlp), ms(lm, dlm)) * min(max(0.995, 1 - mu), 0.999999)
        U = U + a * dU; tp = tp + a * dtp; tm = tm + a * dtm; lp = lp + a * dlp; lm = lm + a * dlm
    return U, lp, lm, it, 2

if __name__ == '__main__':
    orc = O.Oracle('f64')
    N, Tf = 50, 0.5; dt = Tf / N
    cfg = O.make_cfg("L", N, Tf)
    rng = np.random.default_rng(0)
    for delta in (None, 1e-1, 3e-2, 1e-2, 3e-3, 1e-3):
      for shift in ((True, False) if delta else (True,)):
        its = []; errs = []; fails = 0
        rng = np.random.default_rng(0)
        for sc in range(6):
            psi = rng.uniform(-0.1, 0.1)
            st8 = np.array([15.0, rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), 8.0, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)])
            lap, pp = 0, 0.0; warm = None
            for t in range(60):
                hd = np.arctan2(st8[3], st8[2])
                wp, sp, p0, pp = orc.skidpad_waypoints(st8[0], st8[1], hd, lap, pp, 8.0, N, dt)
                x0 = st8[[0, 1, 2, 3, 5, 6, 7]]
                H, g, A, lo, up = O.condensed_qp(cfg, x0, wp, sp)
                Uu = -np.linalg.solve(H, g); v = A @ Uu
                if (v <= up).all() and (v >= lo).all():
                    U = Uu; warm = None
                else:
                    U, lp, lm, it, stt = ipm3(H, g, lo[N], up[N], 0.9, 0.3, dt, N, warm=warm if delta else None, delta=delta or 0, shift=shift)
                    its.append(it); fails += stt != 0
                    warm = (U, lp, lm)
                st8 = orc.plant_rk4(st8, U[0], dt, cfg[3:9])
        its = np.array(its)
        print("delta", delta, "shift", shift, "solves", len(its), "mean it %.2f" % its.mean(), "max", its.max(), "fails", fails)
