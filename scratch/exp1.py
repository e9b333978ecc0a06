"""Scratch experiment: condensed L-model QP, exact IPM truth, PDAS and ADMM behaviour in closed loop."""
import numpy as np, scipy.linalg as sla, sys, time

m, Iz, wb, xcg = 180.0, 294.0, 1.53, 0.57
lf, lr = xcg * wb, (1 - xcg) * wb
Cy = np.array([1.537405752168591e04, 2.417765976460659e04, 3.121158998819641e04, 3.636055041362088e04])
Cx = [300, 500, 700, 900]
Cf = float(np.interp((9.81 * m / 2) * (1 - xcg), Cx, Cy)); Cr = float(np.interp((9.81 * m / 2) * xcg, Cx, Cy))
print(Cf, Cr)

def Ac_L(vx):
    A = np.zeros((6, 6))
    A[1, 2] = vx; A[1, 3] = 1; A[2, 4] = 1
    A[3, 3] = -(Cf + Cr) / (m * vx + 1e-3); A[3, 4] = (-vx + (Cr * lr - Cf * lf)) / (m * vx + 1e-3); A[3, 5] = -Cf / m
    A[4, 3] = (lr * Cr - lf * Cf) / (Iz * vx + 1e-3); A[4, 4] = -(lf * lf * Cf + lr * lr * Cr) / (Iz * vx + 1e-3); A[4, 5] = -lf * Cf / Iz
    return A

def dare_P(dt):
    A = (np.eye(6) + dt * Ac_L(15.0)).astype(np.float32)[1:, 1:]
    B = (dt * np.eye(6)[:, 5:6]).astype(np.float32)[1:, :]
    Q = np.diag([1e5, 1, 0, 0, 0.0]); R = np.zeros((1, 1))
    P = sla.solve_discrete_are(A.astype(float), B.astype(float), Q, R)
    return P

def build(N, dt, vx, beta=1.0):
    A = np.eye(6) + dt * Ac_L(vx); B = np.zeros(6); B[5] = dt
    W = np.diag([0, 1e5, 1, 0, 0, 1.0]); Rr = 10.0
    We = np.zeros((6, 6)); We[1:, 1:] = beta * 0.5 * dare_P(dt)
    # Phi_k, Gamma
    Phi = [np.eye(6)]
    for k in range(N): Phi.append(A @ Phi[-1])
    Gam = np.zeros((N + 1, 6, N))
    for j in range(N):
        v = B.copy()
        for k in range(j + 1, N + 1):
            Gam[k, :, j] = v; v = A @ v
    H = np.zeros((N, N))
    for k in range(1, N):
        H += dt * Gam[k].T @ W @ Gam[k]
    H += Gam[N].T @ We @ Gam[N]
    H += dt * Rr * np.eye(N)
    return dict(A=A, B=B, W=W, We=We, Phi=np.array(Phi), Gam=Gam, H=H, N=N, dt=dt, Rr=Rr)

def grad(md, xhat, yref):
    """yref (N+1,6) state refs; returns g and constant free response"""
    N, dt = md['N'], md['dt']
    g = np.zeros(N)
    xf = np.einsum('kij,j->ki', md['Phi'], xhat)  # (px drift constant ignored: zero weight)
    for k in range(1, N):
        g += dt * md['Gam'][k].T @ md['W'] @ (xf[k] - yref[k])
    g += md['Gam'][N].T @ md['We'] @ (xf[N] - yref[N])
    return g, xf

def cons(md, dhat):
    N, dt = md['N'], md['dt']
    T = np.tril(np.ones((N, N)), -1)[1:] * dt  # rows k=1..N-1: sum_{j<k} u_j dt
    Acon = np.vstack([np.eye(N), T])
    lo = np.concatenate([-0.9 * np.ones(N), (-0.3 - dhat) * np.ones(N - 1)])
    up = np.concatenate([0.9 * np.ones(N), (0.3 - dhat) * np.ones(N - 1)])
    return Acon, lo, up

def ipm(H, g, A, lo, up, tol=1e-12, maxit=100):
    """dense Mehrotra PD-IPM truth: min .5u'Hu+g'u, lo<=Au<=up"""
    n = len(g); G = np.vstack([A, -A]); h = np.concatenate([up, -lo]); mc = len(h)
    u = np.zeros(n); s = np.maximum(h - G @ u, 1e-2) ; s[:] = np.maximum(s, 1.0); z = np.ones(mc)
    for it in range(maxit):
        rd = H @ u + g + G.T @ z; rp = G @ u + s - h; mu = s @ z / mc
        if max(np.abs(rd).max(), np.abs(rp).max()) < tol and mu < tol: break
        Wd = z / s
        K = H + G.T @ (Wd[:, None] * G); c = sla.cho_factor(K)
        def solve(rc):
            # rc = s*z target residual: s dz + z ds = -rc
            rhs = -rd - G.T @ ((-rc + z * rp) / s) * 1.0
            # derive: ds = -rp - G du ; dz = (-rc - z ds)/s = (-rc + z rp + z G du)/s
            du = sla.cho_solve(c, -rd - G.T @ ((-rc + z * rp) / s))
            ds = -rp - G @ du; dz = (-rc - z * ds) / s
            return du, ds, dz
        du, ds, dz = solve(s * z)
        def step(v, dv):
            neg = dv < 0
            return min(1.0, (-v[neg] / dv[neg]).min()) if neg.any() else 1.0
        a = min(step(s, ds), step(z, dz)); mu_aff = (s + a * ds) @ (z + a * dz) / mc
        sig = (mu_aff / mu) ** 3
        du, ds, dz = solve(s * z + ds * dz - sig * mu)
        a = 0.995 * min(step(s, ds), step(z, dz)); a = min(a, 1.0)
        u += a * du; s += a * ds; z += a * dz
    lam = z[:mc // 2] - z[mc // 2:]
    return u, lam, it

def pdas(H, Hinv, g, A, lo, up, act0=None, maxit=30, c=1.0):
    """primal-dual active set. returns u, lam, iters, ok"""
    n = len(g); mcon = len(lo)
    Uunc = -Hinv @ g
    G0 = A @ Hinv @ A.T
    act = np.zeros(mcon, int) if act0 is None else act0.copy()  # +1 upper, -1 lower
    if act0 is None:
        v = A @ Uunc
        act = np.where(v > up, 1, np.where(v < lo, -1, 0))
    seen = []
    for it in range(maxit):
        idx = np.nonzero(act)[0]
        lam = np.zeros(mcon)
        if len(idx):
            b = np.where(act[idx] > 0, up[idx], lo[idx])
            S = G0[np.ix_(idx, idx)]
            try:
                lam[idx] = sla.solve(S, (A @ Uunc)[idx] - b, assume_a='pos')
            except Exception:
                return None, None, it, False
        u = Uunc - Hinv @ (A.T @ lam)
        v = A @ u
        new = np.where((lam + c * (v - up) > 0) & ((act > 0) | (v > up + 1e-12)), 1, 0)
        new = np.where((lam + c * (v - lo) < 0) & ((act < 0) | (v < lo - 1e-12)), -1, new)
        # standard PDAS rule:
        new = np.where(lam + c * (v - up) > 0, 1, np.where(lam + c * (v - lo) < 0, -1, 0))
        if (new == act).all():
            return u, lam, it + 1, True
        key = new.tobytes()
        if key in seen: return u, lam, it + 1, False
        seen.append(key); act = new
    return u, lam, maxit, False

def admm(H, g, A, lo, up, rho_vec, sigma=1e-6, alpha=1.6, iters=50, x0=None, z0=None, y0=None):
    n = len(g)
    K = H + sigma * np.eye(n) + A.T @ (rho_vec[:, None] * A); M = np.linalg.inv(K)
    x = np.zeros(n) if x0 is None else x0.copy(); z = np.clip(A @ x, lo, up) if z0 is None else z0.copy(); y = np.zeros(len(lo)) if y0 is None else y0.copy()
    hist = []
    for it in range(iters):
        xt = M @ (sigma * x - g + A.T @ (rho_vec * z - y))
        zt = A @ xt
        x = alpha * xt + (1 - alpha) * x
        zh = alpha * zt + (1 - alpha) * z
        zn = np.clip(zh + y / rho_vec, lo, up)
        y = y + rho_vec * (zh - zn); z = zn
        act = np.where((z >= up) & (y > 0), 1, np.where((z <= lo) & (y < 0), -1, 0))
        hist.append(act)
    return x, z, y, hist

# ---------------- planner / plant (numpy restatement for the experiment) -------------
r_, center = 9.125, 16.75; lapL = 2 * r_ * np.pi
def pos2progress(x, y, lap):
    if lap == 0: return x
    if 0 < lap < 3: return (-np.arctan2(x - center, -r_ - y) + np.pi) * r_ + (lap - 1) * lapL + center
    if 2 < lap < 5: return (np.arctan2(center - x, y - r_) + np.pi) * r_ + (lap - 1) * lapL + center
    return x + 4 * lapL
def p2pose(p):
    out = np.zeros((len(p), 4))
    for i, q in enumerate(p):
        if (q - center) // (4 * lapL) > 0: out[i] = [q - 4 * lapL, 0, 1, 0]
        elif (q - center) // (2 * lapL) > 0:
            a = (q - center - 2 * lapL) / r_; b = (q - center) / r_
            out[i] = [center + np.sin(a) * r_, r_ - np.cos(a) * r_, np.cos(b), np.sin(b)]
        elif q - center > 0:
            b = (q - center) / r_; out[i] = [center + np.sin(b) * r_, -r_ + np.cos(b) * r_, np.cos(b), -np.sin(b)]
        elif q < center: out[i] = [q, 0, 1, 0]
    return out
class Planner:
    def __init__(s, v, N, dt): s.v, s.N, s.dt, s.prev = v, N, dt, 0.0
    def wp(s, x, y, hd, lap):
        p0 = pos2progress(x, y, lap)
        if s.prev > p0 + 2 * r_: lap += 1; p0 = pos2progress(x, y, lap)
        pr = p0 + np.concatenate([[0], np.cumsum(np.full(s.N, s.v))]) * s.dt
        w = p2pose(pr); s.prev = p0
        w[:, 0] -= x; w[:, 1] -= y
        R = np.array([[np.cos(hd), -np.sin(hd)], [np.sin(hd), np.cos(hd)]])
        w[:, 2:] = w[:, 2:] @ R; w[:, :2] = w[:, :2] @ R
        return w
def plant_f(x, u):
    d = np.zeros(8)
    d[0] = x[2] * x[4] - x[3] * x[5]; d[1] = x[3] * x[4] + x[2] * x[5]; d[2] = -x[6] * x[3]; d[3] = x[6] * x[2]
    d[5] = (-(Cf + Cr) / (m * x[4]) * x[5] + (-x[4] + (Cr * lr - Cf * lf) / (m * x[4])) * x[6]) - Cf / m * x[7]
    d[6] = (lr * Cr - lf * Cf) / (Iz * x[4]) * x[5] - (lf * lf * Cf + lr * lr * Cr) / (Iz * x[4]) * x[6] - Cf * lf / Iz * x[7]
    d[7] = u
    return d
def rk4(x, u, dt):
    k1 = plant_f(x, u); k2 = plant_f(x + dt / 2 * k1, u); k3 = plant_f(x + dt / 2 * k2, u); k4 = plant_f(x + dt * k3, u)
    return x + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6

if __name__ == '__main__':
    N, dt, vx = 50, 0.01, 8.0
    md = build(N, dt, vx)
    H = md['H']; Hinv = np.linalg.inv(H)
    ev = np.linalg.eigvalsh(H); print('eig H', ev[0], ev[-1], 'cond', ev[-1] / ev[0])
    rng = np.random.default_rng(0)
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    stats = dict(nact=[], pd_it=[], pd_fail=0, admm_ident=[])
    for scen in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        psi = rng.uniform(-0.1, 0.1)
        x = np.array([15.0, rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)])
        if scen == 0: x = np.array([15.0, 0.1, 1, 0, 8.0, 0, 0, 0])
        lap = 0; pl = Planner(vx, N, dt)
        for t in range(nsteps):
            hd = np.arctan2(x[3], x[2])
            w = pl.wp(x[0], x[1], hd, lap)
            yref = np.zeros((N + 1, 6)); yref[:, 0] = w[:, 0]; yref[:, 1] = w[:, 1]; yref[:, 2] = w[:, 3]
            xhat = np.array([0, 0, 0, x[5], x[6], x[7]])
            g, xf = grad(md, xhat, yref)
            A, lo, up = cons(md, x[7])
            u, lam, it = ipm(H, g, A, lo, up)
            nact = int((np.abs(lam) > 1e-9).sum())
            up_, lam_p, pit, ok = pdas(H, Hinv, g, A, lo, up)
            if not ok or np.abs(up_ - u).max() > 1e-7:
                stats['pd_fail'] += 1
                print('PDAS fail', scen, t, ok, None if up_ is None else np.abs(up_ - u).max(), 'nact', nact)
            stats['nact'].append(nact); stats['pd_it'].append(pit)
            if nact > 0 and t % 5 == 0:
                # admm identification speed from cold start
                true_act = np.where(lam > 1e-9, 1, np.where(lam < -1e-9, -1, 0))
                res = []
                for rho in (0.1, 1.0, 10.0, 100.0):
                    rv = np.full(len(lo), rho); rv[N:] *= 1.0
                    xa, za, ya, hist = admm(H, g, A, lo, up, rv, iters=300)
                    first = next((i for i in range(len(hist)) if all((h == true_act).all() for h in hist[i:i + 5])), -1)
                    res.append((rho, first, float(np.abs(xa - u).max())))
                stats['admm_ident'].append(res)
                if t % 20 == 0: print(scen, t, 'nact', nact, 'u0', u[0], 'admm', res)
            xn = rk4(x, u[0], dt)
            if xn[0] > center and x[0] <= center: lap += 1
            x = xn
        print('scen', scen, 'final', x, 'lap', lap)
    na = np.array(stats['nact']); pi = np.array(stats['pd_it'])
    print('steps', len(na), 'frac constrained', (na > 0).mean(), 'max nact', na.max(), 'pdas iters mean/max', pi.mean(), pi.max(), 'fails', stats['pd_fail'])
