import sys, numpy as np
sys.path.insert(0,'/root/repo')
from scipy.optimize import linprog
from mpc_amd.controllers import LOcp
from mpc_amd.stability import lp_maximize_batched
ctl = LOcp(50, 0.5)
A, B, K = ctl.A_stability.astype(np.float64), ctl.B_stability.astype(np.float64).reshape(5,1), np.atleast_2d(ctl.K)
ubx = np.array([100, 100, 100, 100, ctl.max_steering]); ubu = np.array([ctl.max_steering_rate])
print("K", K)
phi = A - B @ K; p1 = phi.copy(); rows=[]; rhs=[]
for n in range(2, 8):
    rows += [p1, -p1, K@p1, -(K@p1)]; rhs += [ubx, ubx, ubu, ubu]
    p1 = p1 @ phi
    C = np.array([p1[j]/ubx[j] for j in range(5)] + [(K@p1)[0]/ubu[0]] + [-p1[j]/ubx[j] for j in range(5)] + [-(K@p1)[0]/ubu[0]])
    Am = np.concatenate(rows); bm = np.concatenate(rhs)
    opt, xs, st = lp_maximize_batched(Am, bm, C)
    ref = []
    for c in C:
        r = linprog(-c, A_ub=Am, b_ub=bm, bounds=(None,None), method="highs")
        ref.append(-r.fun if r.status==0 else (np.inf if r.status==3 else np.nan))
    print(n, "rows", Am.shape[0], "status", st.tolist())
    print("  gpu", np.array2string(opt, precision=4)); print("  ref", np.array2string(np.array(ref), precision=4))
    print("  |A|max %.3g  cond rows %.3g" % (np.abs(Am).max(), np.linalg.cond(Am)))
