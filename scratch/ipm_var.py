import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from oracle import oracle as O

def ipm(H, g, lo_d, up_d, umax, dmax, dt, N, tol_mu=1e-15, tol_res=1e-12, maxit=100, variant="base", verbose=False):
    nd = N - 1
    T = np.tril(np.ones((N, N)))[:nd] * dt
    A = np.vstack([np.eye(N), T])
    up = np.concatenate([np.full(N, umax), np.full(nd, up_d)]); lo = np.concatenate([np.full(N, -umax), np.full(nd, lo_d)])
    m = len(up)
    U = np.zeros(N)
    tp = np.concatenate([np.full(N, umax), np.full(nd, max(up_d, 0.1 * dmax))])
    tm = np.concatenate([np.full(N, umax), np.full(nd, max(-lo_d, 0.1 * dmax))])
    lp = np.ones(m); lm = np.ones(m)
    gscale = max(1.0, np.abs(g).max())
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
        if verbose: print(it, "mu %.3e res %.3e" % (mu, res), "minprod %.2e" % min((tp * lp).min(), (tm * lm).min()), end=' ')
        if mu <= tol_mu and (res <= tol_res or res >= 0.5 * res_prev): return U, it, 0
        if it >= maxit: return U, it, 2
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
        a_aff = min(ms(tp, dtp), ms(tm, dtm), ms(lp, dlp), ms(lm, dlm))
        mua = ((tp + a_aff * dtp) @ (lp + a_aff * dlp) + (tm + a_aff * dtm) @ (lm + a_aff * dlm)) / (2 * m)
        sig = (mua / mu) ** 3
        sm = sig * mu
        d2 = direction(tp * lp + dtp * dlp - sm, tm * lm + dtm * dlm - sm)
        a2 = min(ms(tp, d2[1]), ms(tm, d2[2]), ms(lp, d2[3]), ms(lm, d2[4]))
        if variant == "safeguard" and a2 < 0.5 * a_aff and a2 < 0.9:
            # corrector made it worse: centering without the second-order term
            d3 = direction(tp * lp - sm, tm * lm - sm)
            a3 = min(ms(tp, d3[1]), ms(tm, d3[2]), ms(lp, d3[3]), ms(lm, d3[4]))
            if a3 > a2: d2, a2 = d3, a3
        if variant == "split":
            ap = min(ms(tp, d2[1]), ms(tm, d2[2])) * 0.995; ad = min(ms(lp, d2[3]), ms(lm, d2[4])) * 0.995
        else:
            ap = ad = a2 * 0.995
        if variant == "mstep":   # Mehrotra's step heuristic replaced by fixed fraction to boundary 0.9 early
            ap = ad = a2 * (0.995 if mu < 1e-6 else 0.9)
        if verbose: print("a_aff %.3f a %.3f sig %.2e" % (a_aff, a2, sig))
        dU, dtp, dtm, dlp, dlm = d2
        U = U + ap * dU; tp = tp + ap * dtp; tm = tm + ap * dtm; lp = lp + ad * dlp; lm = lm + ad * dlm
        if variant == "floor":
            # keep pairwise products from collapsing far below the average (central-path neighbourhood)
            mun = (tp @ lp + tm @ lm) / (2 * m)
            lp = np.maximum(lp, 1e-3 * mun / tp); lm = np.maximum(lm, 1e-3 * mun / tm)
    return U, it, 2

if __name__ == '__main__':
    orc = O.Oracle('f64')
    N, Tf = 50, 0.5; dt = Tf / N
    cfg = O.make_cfg("OFL", N, Tf)
    probs = []
    for i in range(2):
        xe = np.load(f"gpurun_out/bad_est_{i}.npy")
        wp, sp, p0, ab = orc.step_waypoints(xe[0], xe[1], np.arctan2(xe[3], xe[2]), 15.0, N, dt)
        probs.append(O.condensed_qp(cfg, xe[[0, 1, 2, 3, 5, 6, 7]], wp, sp, xe[9]))
    # ordinary problems: random skidpad L starts + OFL step starts
    rng = np.random.default_rng(0)
    cfgL = O.make_cfg("L", N, Tf)
    for i in range(60):
        psi = rng.uniform(-0.1, 0.1)
        s = [15.0 + rng.uniform(0, 40), rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), 8.0, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.25, 0.25)]
        wp, sp, _, _ = orc.skidpad_waypoints(s[0], s[1], psi, 0, 0.0, 8.0, N, dt)
        probs.append(O.condensed_qp(cfgL, np.array(s)[[0, 1, 2, 3, 5, 6, 7]], wp, sp))
    for i in range(60):
        xe = np.array([-5 + rng.uniform(0, 10), rng.normal(0, 0.3), 1, rng.normal(0, 0.05), 15, rng.normal(0, 0.2), rng.normal(0, 0.2), rng.uniform(-0.29, 0.29), 0, rng.uniform(-12, 2)])
        wp, sp, p0, ab = orc.step_waypoints(xe[0], xe[1], np.arctan2(xe[3], xe[2]), 15.0, N, dt)
        probs.append(O.condensed_qp(cfg, xe[[0, 1, 2, 3, 5, 6, 7]], wp, sp, xe[9]))
    for variant in ("base", "safeguard", "split", "mstep", "floor"):
        its = []; fails = 0; errs = []
        for (H, g, A, lo, up) in probs:
            U, it, st = ipm(H, g, lo[N], up[N], 0.9, 0.3, dt, N, variant=variant)
            its.append(it); fails += st != 0
        its = np.array(its)
        print(variant, "fails", fails, "first two", its[:2], "mean", its[2:].mean(), "max", its[2:].max())
    if len(sys.argv) > 1:
        H, g, A, lo, up = probs[int(sys.argv[2])]
        ipm(H, g, lo[N], up[N], 0.9, 0.3, dt, N, variant=sys.argv[1], verbose=True, maxit=40)

def ipm2(H, g, lo_d, up_d, umax, dmax, dt, N, tol_mu=1e-15, tol_res=1e-12, maxit=100, split=True, lam0=1.0, frac=0.995, t0=None, sigpow=3):
    nd = N - 1
    T = np.tril(np.ones((N, N)))[:nd] * dt
    A = np.vstack([np.eye(N), T])
    up = np.concatenate([np.full(N, umax), np.full(nd, up_d)]); lo = np.concatenate([np.full(N, -umax), np.full(nd, lo_d)])
    m = len(up)
    U = np.zeros(N)
    tp = np.concatenate([np.full(N, umax), np.full(nd, max(up_d, 0.1 * dmax))])
    tm = np.concatenate([np.full(N, umax), np.full(nd, max(-lo_d, 0.1 * dmax))])
    gscale = max(1.0, np.abs(g).max())
    l0 = lam0(gscale) if callable(lam0) else lam0
    lp = np.full(m, l0); lm = np.full(m, l0)
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
        if mu <= tol_mu * max(1.0, l0 if False else 1.0) and (res <= tol_res or res >= 0.5 * res_prev): return U, it, 0
        if it >= maxit: return U, it, 2
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
        if split:
            app = min(ms(tp, dtp), ms(tm, dtm)); adp = min(ms(lp, dlp), ms(lm, dlm))
        else:
            app = adp = min(ms(tp, dtp), ms(tm, dtm), ms(lp, dlp), ms(lm, dlm))
        mua = ((tp + app * dtp) @ (lp + adp * dlp) + (tm + app * dtm) @ (lm + adp * dlm)) / (2 * m)
        sm = (mua / mu) ** sigpow * mu
        dU, dtp, dtm, dlp, dlm = direction(tp * lp + dtp * dlp - sm, tm * lm + dtm * dlm - sm)
        ap = min(ms(tp, dtp), ms(tm, dtm)); ad = min(ms(lp, dlp), ms(lm, dlm))
        if not split: ap = ad = min(ap, ad)
        f = frac(mu) if callable(frac) else frac
        ap *= f; ad *= f
        U = U + ap * dU; tp = tp + ap * dtp; tm = tm + ap * dtm; lp = lp + ad * dlp; lm = lm + ad * dlm
    return U, it, 2

if __name__ == '__main__':
    truth = [ipm2(H, g, lo[N], up[N], 0.9, 0.3, dt, N, tol_mu=1e-18)[0] for (H, g, A, lo, up) in probs]
    for name, kw in [("split", {}), ("split lam sqrt", dict(lam0=lambda s: np.sqrt(s))), ("split lam s/100", dict(lam0=lambda s: max(1, s / 100))),
                     ("split lam 10", dict(lam0=10.0)), ("split lam 100", dict(lam0=100.0)), ("split frac .99", dict(frac=0.99)),
                     ("split frac .999", dict(frac=0.999)), ("split tol 1e-13", dict(tol_mu=1e-13)), ("nosplit lam 100", dict(split=False, lam0=100.0)),
                     ("split frac adaptive", dict(frac=lambda mu: max(0.995, 1 - mu))), ("split sig2", dict(sigpow=2)), ("split sig4", dict(sigpow=4))]:
        its = []; fails = 0; err = 0
        for (H, g, A, lo, up), Ut in zip(probs, truth):
            U, it, st = ipm2(H, g, lo[N], up[N], 0.9, 0.3, dt, N, **kw)
            its.append(it); fails += st != 0; err = max(err, np.abs(U - Ut).max())
        its = np.array(its)
        print("%-22s fails %d first two %s mean %.2f max %d err %.1e" % (name, fails, its[:2], its[2:].mean(), its[2:].max(), err))
    print("---- lam0 sweep")
    for name, kw in [("s/10", dict(lam0=lambda s: max(1, s / 10))), ("s/30", dict(lam0=lambda s: max(1, s / 30))), ("s/100", dict(lam0=lambda s: max(1, s / 100))),
                     ("s/300", dict(lam0=lambda s: max(1, s / 300))), ("s/1000", dict(lam0=lambda s: max(1, s / 1000))),
                     ("s/100 nosplit", dict(lam0=lambda s: max(1, s / 100), split=False)), ("s/100 adaptive", dict(lam0=lambda s: max(1, s / 100), frac=lambda mu: max(0.995, 1 - mu))),
                     ("s/100 tol13", dict(lam0=lambda s: max(1, s / 100), tol_mu=1e-13)), ("s/30 nosplit", dict(lam0=lambda s: max(1, s / 30), split=False))]:
        its = []; fails = 0; err = 0
        for (H, g, A, lo, up), Ut in zip(probs, truth):
            U, it, st = ipm2(H, g, lo[N], up[N], 0.9, 0.3, dt, N, **kw)
            its.append(it); fails += st != 0; err = max(err, np.abs(U - Ut).max())
        its = np.array(its)
        print("%-22s fails %d first two %s mean %.2f max %d err %.1e" % (name, fails, its[:2], its[2:].mean(), its[2:].max(), err))
