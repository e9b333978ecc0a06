import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle as O
orc = O.Oracle('f64')
N, Tf = 80, 0.8
cfg = O.make_cfg("LPV", N, Tf)
rng = np.random.default_rng(1)
allit = []; sts = {}
for sc in range(12):
    psi = rng.uniform(-0.1, 0.1); vx = rng.uniform(6, 12)
    st8 = np.array([15.0, rng.uniform(-0.3, 0.3), np.cos(psi), np.sin(psi), vx, rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)])
    lap, pp = 0, 0.0
    lx = np.zeros((6, N + 1)); lu = np.zeros(N)
    for t in range(100):
        st, st8, lap, pp, u0, info = orc.skidpad_step(cfg, st8, lap, pp, vx, lx, lu)
        allit.append(info[0]); sts[st] = sts.get(st, 0) + 1
        if st not in (0,) and sts[st] <= 3:
            print("scenario", sc, "t", t, "status", st, "info", info, "state", st8)
    print(sc, "vx %.2f" % vx, "y %.3f" % st8[1], "delta %.3f" % st8[7])
allit = np.array(allit)
print("mean it", allit.mean(), "max", allit.max(), "status counts", sts, "hist", np.bincount(allit.astype(int)).tolist())
