import numpy as np, sys, torch
sys.path.insert(0, '.')
from oracle import oracle as O
from mpc_amd.controllers import NLOcp
orc80 = O.Oracle('f80')
N, Tf, B = 50, 0.5, 24
rng = np.random.default_rng(21)
psi = rng.uniform(-0.1, 0.1, B)
st = np.stack([np.full(B, 15.0), rng.uniform(-0.3, 0.3, B), np.cos(psi), np.sin(psi), np.full(B, 8.0),
               rng.uniform(-0.2, 0.2, B), rng.uniform(-0.2, 0.2, B), rng.uniform(-0.05, 0.05, B)], axis=1)
x0 = st[:, [0, 1, 2, 3, 5, 6, 7]]
wps, ps = [], []
for s in st:
    wp, sp, _, _ = orc80.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), 0, 0.0, s[4], N, Tf / N)
    wps.append(wp); ps.append(sp)
wps = np.array(wps); ps = np.array(ps)
cfg = O.make_cfg("NL", N, Tf, sqp_tol=1e-11, sqp_max_iter=50)
refs = [orc80.nl_optimize(cfg, x0[b], wps[b], ps[b]) for b in range(B)]
for tol, mx, tmu in ((1e-11, 50, 1e-13), (1e-9, 50, 1e-13), (1e-8, 50, 1e-13), (1e-11, 50, 1e-16), (1e-10, 1, 1e-13), (1e-10, 2, 1e-13), (1e-10, 3, 1e-13), (1e-10, 5, 1e-13)):
    ctl = NLOcp(N, Tf, batch=B, sqp_tol=tol, sqp_max_iter=mx, tol_mu=tmu)
    U, traj, status, iters = ctl.solve_batched(x0, wps, ps)
    torch.cuda.synchronize()
    U = U.cpu().numpy()
    err = np.array([np.abs(U[b] - refs[b][2]).max() for b in range(B)])
    print(tol, mx, tmu, "status", status.cpu().numpy().tolist(), "sqp", ctl.stats()["sum_sqp_iterations"], "ipm", iters.cpu().numpy().sum(), "maxerr %.2e" % err.max(), np.array2string(err, precision=1))
