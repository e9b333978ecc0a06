"""Scratch experiment 2: warm-started ADMM + active-set polish + KKT verify in closed loop."""
import numpy as np, scipy.linalg as sla, sys
from exp1 import *

def polish(H, Hinv, G0, HiAt, g, A, lo, up, act):
    idx = np.nonzero(act)[0]
    Uunc = -Hinv @ g
    lam = np.zeros(len(lo))
    if len(idx):
        b = np.where(act[idx] > 0, up[idx], lo[idx])
        S = G0[np.ix_(idx, idx)]
        try:
            c = sla.cho_factor(S)
        except Exception:
            return None, None, False
        lam[idx] = sla.cho_solve(c, (A @ Uunc)[idx] - b)
    u = Uunc - HiAt @ lam
    # one refinement step on stationarity
    v = A @ u
    feas = (v <= up + 1e-9).all() and (v >= lo - 1e-9).all()
    sign = ((lam[act > 0] >= -1e-9).all() if (act > 0).any() else True) and ((lam[act < 0] <= 1e-9).all() if (act < 0).any() else True)
    return u, lam, bool(feas and sign)

def truth(H, Hinv, G0, HiAt, g, A, lo, up):
    # ADMM long run then polish; verified
    rv = np.full(len(lo), 1.0)
    x, z, y, hist = admm(H, g, A, lo, up, rv, iters=2000)
    act = hist[-1]
    u, lam, ok = polish(H, Hinv, G0, HiAt, g, A, lo, up, act)
    if not ok:
        x, z, y, hist = admm(H, g, A, lo, up, rv, iters=20000)
        u, lam, ok = polish(H, Hinv, G0, HiAt, g, A, lo, up, hist[-1])
    return u, lam, ok

class Hybrid:
    def __init__(s, H, A, rho_u=1.0, rho_d=1.0, sigma=1e-6, alpha=1.6):
        n = H.shape[0]; s.n = n; s.A = A; s.H = H
        s.rv = np.concatenate([np.full(n, rho_u), np.full(A.shape[0] - n, rho_d)])
        s.sigma, s.alpha = sigma, alpha
        s.M = np.linalg.inv(H + sigma * np.eye(n) + A.T @ (s.rv[:, None] * A))
        s.Hinv = np.linalg.inv(H); s.G0 = A @ s.Hinv @ A.T; s.HiAt = s.Hinv @ A.T
        s.x = np.zeros(n); s.y = np.zeros(A.shape[0])
    def solve(s, g, lo, up, check_every=1, stable=3, maxit=2000):
        A, rv, sg, al = s.A, s.rv, s.sigma, s.alpha
        Uunc = -s.Hinv @ g; v = A @ Uunc
        if (v <= up).all() and (v >= lo).all():
            s.x = Uunc; s.y[:] = 0; return Uunc, 0, 0
        x = s.x; y = s.y; z = np.clip(A @ x, lo, up)
        last = None; cnt = 0; npol = 0
        for it in range(1, maxit + 1):
            xt = s.M @ (sg * x - g + A.T @ (rv * z - y)); zt = A @ xt
            x = al * xt + (1 - al) * x; zh = al * zt + (1 - al) * z
            zn = np.clip(zh + y / rv, lo, up); y = y + rv * (zh - zn); z = zn
            act = np.where((z >= up) & (y > 0), 1, np.where((z <= lo) & (y < 0), -1, 0))
            if last is not None and (act == last).all(): cnt += 1
            else: cnt = 0
            last = act
            if cnt >= stable:
                u, lam, ok = polish(s.H, s.Hinv, s.G0, s.HiAt, g, A, lo, up, act); npol += 1
                cnt = 0
                if ok:
                    s.x = u; s.y = lam; return u, it, npol
        s.x = x; s.y = y
        return x, it, npol
    def shift(s):
        n = s.n
        s.x = np.concatenate([s.x[1:], s.x[-1:]])
        yu = s.y[:n]; yd = s.y[n:]
        s.y = np.concatenate([yu[1:], yu[-1:], yd[1:], yd[-1:]])

if __name__ == '__main__':
    N, dt, vx = 50, 0.01, 8.0
    md = build(N, dt, vx); H = md['H']
    A, _, _ = cons(md, 0.0)
    rho_u = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rho_d = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    stable = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    hy = Hybrid(H, A, rho_u, rho_d)
    rng = np.random.default_rng(0)
    nsteps = int(sys.argv[1]); nsc = int(sys.argv[2])
    its = []; pols = []; errs = []; nacts = []
    for scen in range(nsc):
        psi = rng.uniform(-0.1, 0.1)
        x = np.array([15.0, rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)])
        if scen == 0: x = np.array([15.0, 0.1, 1, 0, 8.0, 0, 0, 0])
        if scen == 1: x = np.array([15.0, -0.3, 1, 0, 8.0, 0, 0, 0.28])
        lap = 0; pl = Planner(vx, N, dt); hy.x[:] = 0; hy.y[:] = 0
        for t in range(nsteps):
            hd = np.arctan2(x[3], x[2]); w = pl.wp(x[0], x[1], hd, lap)
            yref = np.zeros((N + 1, 6)); yref[:, 0] = w[:, 0]; yref[:, 1] = w[:, 1]; yref[:, 2] = w[:, 3]
            xhat = np.array([0, 0, 0, x[5], x[6], x[7]])
            g, xf = grad(md, xhat, yref); _, lo, up = cons(md, x[7])
            u, it, npol = hy.solve(g, lo, up, stable=stable)
            ut, lamt, okt = truth(H, hy.Hinv, hy.G0, hy.HiAt, g, A, lo, up)
            e = np.abs(u - ut).max(); nact = int((np.abs(lamt) > 1e-10).sum())
            nd = int((np.abs(lamt[N:]) > 1e-10).sum())
            its.append(it); pols.append(npol); errs.append(e); nacts.append(nact)
            if not okt or e > 1e-8 or (t < 40 and t % 4 == 0):
                print(scen, t, 'it', it, 'npol', npol, 'err', e, 'truth_ok', okt, 'nact', nact, 'ndelta', nd, 'u0', ut[0])
            hy.shift()
            xn = rk4(x, ut[0], dt)
            if xn[0] > center and x[0] <= center: lap += 1
            x = xn
    its = np.array(its); print('steps', len(its), 'mean it', its.mean(), 'p50', np.median(its), 'p99', np.percentile(its, 99), 'max', its.max(), 'mean pol', np.mean(pols), 'max err', np.max(errs), 'constrained frac', (np.array(nacts) > 0).mean())
