import numpy as np, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from oracle import oracle as O
from ipm_var import ipm2
orc = O.Oracle('f64')
N, Tf = 80, 0.8; dt = Tf / N
cfg = O.make_cfg("LPV", N, Tf); cfg0 = cfg.copy(); cfg0[0] = 0
print("W", cfg[11:18], "We diag", np.diag(cfg[18:54].reshape(6, 6)))
rng = np.random.default_rng(1)
probs = []
for i in range(80):
    psi = rng.uniform(-0.1, 0.1); vx = rng.uniform(6, 12)
    s = [15.0 + rng.uniform(0, 60), rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.25, 0.25)]
    wp, sp, _, _ = orc.skidpad_waypoints(s[0], s[1], psi, 0, 0.0, vx, N, dt)
    probs.append(O.condensed_qp(cfg0, np.array(s)[[0, 1, 2, 3, 5, 6, 7]], wp, sp))
print("gscale", [float(np.abs(p[1]).max()) for p in probs[:5]], "cond", [float(np.linalg.cond(p[0])) for p in probs[:3]])
truth = [ipm2(H, g, lo[N], up[N], 0.9, 0.3, dt, N, tol_mu=1e-16, lam0=lambda s: max(1, s / 100), frac=0.995, maxit=300)[0] for (H, g, A, lo, up) in probs]
for name, kw in [("old: nosplit lam1", dict(split=False, lam0=1.0)), ("split lam1", dict()), ("new: s/100 adaptive", dict(lam0=lambda s: max(1, s / 100), frac=lambda mu: min(max(0.995, 1 - mu), 0.999999))),
                 ("s/100", dict(lam0=lambda s: max(1, s / 100))), ("s/10 adaptive", dict(lam0=lambda s: max(1, s / 10), frac=lambda mu: min(max(0.995, 1 - mu), 0.999999))),
                 ("s/1000 adaptive", dict(lam0=lambda s: max(1, s / 1000), frac=lambda mu: min(max(0.995, 1 - mu), 0.999999))),
                 ("nosplit s/100 adaptive", dict(split=False, lam0=lambda s: max(1, s / 100), frac=lambda mu: min(max(0.995, 1 - mu), 0.999999))),
                 ("lam1 adaptive", dict(frac=lambda mu: min(max(0.995, 1 - mu), 0.999999)))]:
    its = []; fails = 0; err = 0
    for (H, g, A, lo, up), Ut in zip(probs, truth):
        try:
            U, it, st = ipm2(H, g, lo[N], up[N], 0.9, 0.3, dt, N, maxit=100, **kw)
        except np.linalg.LinAlgError:
            U, it, st = Ut * np.nan, 100, 9
        its.append(it); fails += st != 0; err = max(err, np.abs(U - Ut).max())
    its = np.array(its)
    print("%-24s fails %d mean %.2f max %d err %.1e" % (name, fails, its.mean(), its.max(), err))
