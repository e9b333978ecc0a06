import numpy as np, sys, torch
sys.path.insert(0, '.')
from oracle import oracle as O
from mpc_amd.controllers import LPVOcp
orc = O.Oracle('f64')
N, Tf = 80, 0.8
cfg = O.make_cfg("LPV", N, Tf)
d = np.load("tools/lpv_bad_0.npz")
st8 = d["state"]; vx = st8[4]
hd = np.arctan2(st8[3], st8[2])
wp, sp, p0, pp = orc.skidpad_waypoints(st8[0], st8[1], hd, 0, 0.0, vx, N, Tf / N)
ref = orc.optimize(cfg, st8[[0, 1, 2, 3, 5, 6, 7]], wp, sp, 0.0, np.zeros((6, N + 1)), np.zeros(N))
print("oracle", ref[0], ref[4])
for tmu, mi in ((1e-15, 80), (1e-13, 80), (1e-10, 80), (1e-15, 9), (1e-15, 10), (1e-15, 11), (1e-15, 12), (1e-15, 13), (1e-15, 14)):
    ctl = LPVOcp(N, Tf, batch=1, tol_mu=tmu, max_iter=mi)
    lx = torch.zeros((1, 6, N + 1), dtype=torch.float64, device="cuda"); lu = torch.zeros((1, N), dtype=torch.float64, device="cuda")
    U, traj, status, iters = ctl.solve_batched(st8[None, [0, 1, 2, 3, 5, 6, 7]], wp[None], sp[None], None, lx, lu)
    torch.cuda.synchronize()
    s = ctl.stats()
    print(tmu, mi, "status", status.item(), "iters", iters.item(), "mu %.3e res %.3e" % (s["max_complementarity"], s["max_residual"]), "err", np.abs(U.cpu().numpy()[0] - ref[2]).max())
