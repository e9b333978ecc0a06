"""Offline study (numpy + the CPU checker, no GPU) of the exchange rules of the active-set solve in
mpc_amd/csrc/mpc_aset.hpp on the c2 closed loop: from the previous step's active set moved one stage ahead, how many
solves settle, and in how many iterations, when every wrong-signed and every violated bound is exchanged at once ("pdas")
or one bound leaves and one enters per iteration ("single").  Not collected by pytest.

  python tests/study_active_set_rules.py <scenarios> <steps> <iteration cap> pdas|single|pertype|hybrid [moved|union|stay]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
from bench import initial_states
from collections import Counter
orc = O.Oracle("f64")
N, Tf = 50, 0.5; dt = Tf / N
cfg = O.make_cfg("L", N, Tf)
B = int(sys.argv[1]); nst = int(sys.argv[2]); MAXIT = int(sys.argv[3]); MODE = sys.argv[4]
START = sys.argv[5] if len(sys.argv) > 5 else "moved"     # starting set: the previous one moved a stage, "union" with where it was, or "stay"
st = initial_states(4096, 0)[:B]
umax, dmax = cfg[10], cfg[9]
wid = np.concatenate([np.full(N, umax), np.full(N - 1, dmax)])
def eqp(G, A, Uu, lo, up, au, al):
    idx = np.where(au | al)[0]
    lamv = np.zeros(A.shape[0]); skipped = np.zeros(A.shape[0], bool)
    if len(idx) == 0: return Uu, lamv, skipped
    E = A[idx]; bb = np.where(au[idx], up[idx], lo[idx])
    S = E @ G @ E.T; d = E @ Uu - bb
    Sw = S.copy(); skip = np.zeros(len(idx), bool)
    for K in range(len(idx)):
        if Sw[K, K] > 1e-9 * S[K, K]:
            l = Sw[:, K] / Sw[K, K]; l[:K + 1] = 0
            Sw -= np.outer(l, Sw[K, :])
        else: skip[K] = True
    ii = np.where(~skip)[0]
    nu = np.zeros(len(idx)); nu[ii] = np.linalg.solve(S[np.ix_(ii, ii)], d[ii])
    lamv[idx] = nu; skipped[idx[skip]] = True
    return Uu - G @ E.T @ nu, lamv, skipped
stats = {}
for b in range(B):
    s = st[b].copy(); lap = 0; pp = 0.0
    prev = None
    for k in range(nst):
        wp, sp, p0, _ = orc.skidpad_waypoints(s[0], s[1], np.arctan2(s[3], s[2]), lap, pp, s[4], N, dt)
        x0 = s[[0, 1, 2, 3, 5, 6, 7]]
        stt, traj, U, lam, info = orc.optimize(cfg, x0, wp, sp)
        H, g, A, lo, up = O.condensed_qp(cfg, x0, wp, sp)
        if k == 0 and b == 0: G = np.linalg.inv(H)
        v = A @ U
        aup = v >= up - 1e-9; alo = v <= lo + 1e-9
        Uu = -G @ g; vu = A @ Uu
        if (vu > up).any() or (vu < lo).any():
            phase = "0-50" if k < 50 else "50-70" if k < 70 else "70+"
            t = stats.setdefault(phase, dict(con=0, ok=0, wrong=0, big=0, its=Counter()))
            t["con"] += 1
            if prev is not None:
                pu, pl = prev
                au = np.concatenate([pu[1:N], pu[N-1:N], pu[N+1:], pu[-1:]])
                al = np.concatenate([pl[1:N], pl[N-1:N], pl[N+1:], pl[-1:]])
                if START == "union":        # the moved set and the set where it was: runs keep their far end
                    au, al = (au | pu) & ~pl, (al | pl) & ~pu
                elif START == "stay":
                    au, al = pu.copy(), pl.copy()
                cold = False
            else:
                au = vu > up; al = vu < lo; cold = True
            ok = False
            for it in range(MAXIT):
                if (au | al).sum() > 64: t["big"] += 1; break
                Un, lamv, skipped = eqp(G, A, Uu, lo, up, au, al)
                vv = A @ Un
                bad = (au & ((lamv < 0) | skipped)) | (al & ((lamv > 0) | skipped))
                viol = np.where(~au & ~al, np.maximum(vv - up - 1e-12, lo - 1e-12 - vv), -1.0) / wid
                if not bad.any() and not (viol > 0).any(): ok = True; break
                nau, nal = au.copy(), al.copy()
                if MODE == "pdas" or (MODE == "hybrid" and cold and it == 0):
                    nau &= ~bad; nal &= ~bad
                    nau |= (~au & ~al & (vv > up + 1e-12)); nal |= (~au & ~al & (vv < lo - 1e-12))
                elif MODE == "pertype":
                    # one bound out and one in per kind of bound (input bounds, steering bounds)
                    for sl in (slice(0, N), slice(N, 2 * N - 1)):
                        if bad[sl].any():
                            score = np.where(bad[sl], np.where(skipped[sl], 1e300, np.abs(lamv[sl]) * wid[sl]), -1)
                            j = sl.start + int(np.argmax(score)); nau[j] = False; nal[j] = False
                        if (viol[sl] > 0).any():
                            j = sl.start + int(np.argmax(viol[sl]))
                            if vv[j] > up[j]: nau[j] = True
                            else: nal[j] = True
                else:
                    if bad.any():
                        score = np.where(bad, np.where(skipped, 1e300, np.abs(lamv) * wid), -1)
                        j = int(np.argmax(score)); nau[j] = False; nal[j] = False
                    if (viol > 0).any():
                        j = int(np.argmax(viol))
                        if vv[j] > up[j]: nau[j] = True
                        else: nal[j] = True
                au, al = nau, nal
            if ok:
                if np.abs(Un - U).max() < 1e-6: t["ok"] += 1; t["its"][it + 1] += 1
                else: t["wrong"] += 1
            prev = (aup, alo)
        else:
            prev = None
        stx, s8, lap, pp, u0, _ = orc.skidpad_step(cfg, s, lap, pp, 8.0)
        s = s8
for ph, t in stats.items():
    print(ph, {k: (sorted(v.items()) if isinstance(v, Counter) else v) for k, v in t.items()})
