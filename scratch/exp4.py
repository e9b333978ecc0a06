"""Condensed Mehrotra IPM prototype (fp64) vs long-double truth, closed loop."""
import numpy as np, sys
from exp1 import build, grad, cons, Planner, rk4, center

def ldl_solve_factory(K):
    n = K.shape[0]; L = K.copy(); dt_ = K.dtype
    d = np.zeros(n, dt_)
    for k in range(n):
        d[k] = L[k, k]
        c = L[k+1:, k].copy()
        l = c / d[k]
        L[k+1:, k+1:] -= np.outer(l, c)
        L[k+1:, k] = l
    def solve(b):
        w = b.copy()
        for k in range(n):
            w[k+1:] -= L[k+1:, k] * w[k]
        x = w / d
        for k in range(n-1, -1, -1):
            x[:k] -= L[k, :k] * x[k]
        return x
    return solve

def ipm(H, g, A, lo, up, dtype=np.float64, tol_mu=1e-13, tol_r=1e-10, maxit=60, U0=None, verbose=False):
    H = H.astype(dtype); g = g.astype(dtype); A = A.astype(dtype); lo = lo.astype(dtype); up = up.astype(dtype)
    n = len(g); m = len(lo)
    U = np.zeros(n, dtype) if U0 is None else U0.astype(dtype)
    v = A @ U
    rng_ = up - lo
    tp = np.maximum(up - v, 0.05 * rng_); tm = np.maximum(v - lo, 0.05 * rng_)
    # scale for multipliers
    gs = max(np.abs(H @ U + g).max(), 1.0)
    lp = np.full(m, 1.0, dtype) * gs * 0 + 1.0; lm = lp.copy()
    mu0 = (tp @ lp + tm @ lm) / (2 * m)
    one = dtype(1.0)
    for it in range(maxit):
        v = A @ U
        rd = H @ U + g + A.T @ (lp - lm)
        rpp = v + tp - up; rpm = -v + tm + lo
        mu = (tp @ lp + tm @ lm) / (2 * m)
        res = max(np.abs(rd).max(), np.abs(rpp).max(), np.abs(rpm).max())
        if verbose: print(it, float(mu), float(res))
        if mu <= tol_mu and res <= tol_r: break
        D = lp / tp + lm / tm
        K = H + A.T @ (D[:, None] * A)
        solve = ldl_solve_factory(K)
        def direction(rcp, rcm):
            rhs = -rd - A.T @ ((lp * rpp - rcp) / tp - (lm * rpm - rcm) / tm)
            dU = solve(rhs); dv = A @ dU
            dtp = -rpp - dv; dtm = -rpm + dv
            dlp = -(rcp + lp * dtp) / tp; dlm = -(rcm + lm * dtm) / tm
            return dU, dtp, dtm, dlp, dlm
        def maxstep(x, dx):
            neg = dx < 0
            return min(one, (-x[neg] / dx[neg]).min()) if neg.any() else one
        dU, dtp, dtm, dlp, dlm = direction(tp * lp, tm * lm)
        a = min(maxstep(tp, dtp), maxstep(tm, dtm), maxstep(lp, dlp), maxstep(lm, dlm))
        mu_aff = ((tp + a * dtp) @ (lp + a * dlp) + (tm + a * dtm) @ (lm + a * dlm)) / (2 * m)
        sig = (mu_aff / mu) ** 3
        dU, dtp, dtm, dlp, dlm = direction(tp * lp + dtp * dlp - sig * mu, tm * lm + dtm * dlm - sig * mu)
        ap = min(maxstep(tp, dtp), maxstep(tm, dtm)); ad = min(maxstep(lp, dlp), maxstep(lm, dlm))
        a = min(ap, ad) * dtype(0.995)
        U = U + a * dU; tp = tp + a * dtp; tm = tm + a * dtm; lp = lp + a * dlp; lm = lm + a * dlm
    return U, lp - lm, it

if __name__ == '__main__':
    N, dt, vx = 50, 0.01, 8.0
    md = build(N, dt, vx); H = md['H']; A, _, _ = cons(md, 0.0)
    Hinv = np.linalg.inv(H)
    rng = np.random.default_rng(0); nsteps = int(sys.argv[1]); nsc = int(sys.argv[2]); chk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    its = []; errs = []
    for scen in range(nsc):
        psi = rng.uniform(-0.1, 0.1)
        x = np.array([15.0, rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)])
        if scen == 0: x = np.array([15.0, 0.1, 1, 0, 8.0, 0, 0, 0])
        if scen == 1: x = np.array([15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28])
        if scen == 2: x = np.array([15.0, 0.3, 1, 0, 8.0, 0, 0, 0.28])
        lap = 0; pl = Planner(vx, N, dt)
        for t in range(nsteps):
            hd = np.arctan2(x[3], x[2]); w = pl.wp(x[0], x[1], hd, lap)
            yref = np.zeros((N + 1, 6)); yref[:, 0] = w[:, 0]; yref[:, 1] = w[:, 1]; yref[:, 2] = w[:, 3]
            xhat = np.array([0, 0, 0, x[5], x[6], x[7]])
            g, xf = grad(md, xhat, yref); _, lo, up = cons(md, x[7])
            Uunc = -Hinv @ g; v = A @ Uunc
            if (v <= up).all() and (v >= lo).all():
                u = Uunc; it = 0
            else:
                u, lam, it = ipm(H, g, A, lo, up)
                if t % chk == 0:
                    ut, lamt, itt = ipm(H, g, A, lo, up, dtype=np.longdouble, tol_mu=1e-17, tol_r=1e-15, maxit=80)
                    e = float(np.abs(u - ut).max()); errs.append(e)
                    print(scen, t, 'it', it, 'it_ld', itt, 'err', e, 'u0', u[0], 'nact', int((np.abs(lamt) > 1e-8).sum()), 'nd', int((np.abs(lamt[N:]) > 1e-8).sum()))
            its.append(it)
            xn = rk4(x, u[0], dt)
            if xn[0] > center and x[0] <= center: lap += 1
            x = xn
    its = np.array(its); c = its[its > 0]
    print('steps', len(its), 'constrained', len(c), 'iters mean', c.mean() if len(c) else 0, 'max', c.max() if len(c) else 0, 'max err', max(errs) if errs else None)
