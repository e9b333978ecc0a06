"""Offline study (numpy + the CPU checker, no GPU) of the set the active-set solve of mpc_amd/csrc/mpc_aset.hpp starts
from, on the c2 closed loop (steps 50..70 of bench scenarios): exchange iterations per solve from
  shift   the previous step's set moved one stage ahead,
  best2   per row whichever of the two frames (row i+1 / row i of the previous step) is right, where either is,
  and     aset_predict's rule: a row at the end of a run is active when both frames, continued linearly from the last
          two steps (multiplier of an active row, bound distance of an inactive one), say so.
Not collected by pytest.

  python tests/study_active_set_prediction.py <scenarios> [modes, comma separated] [seed]
"""
import sys, collections
import numpy as np
sys.path.insert(0, ".")
from bench import initial_states
from oracle import oracle as O

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["shift", "best2", "and"]
N, Tf = 50, 0.5
orc = O.Oracle("f64")
cfg = O.make_cfg("L", N, Tf)
dt, umax, dmax = cfg[2], cfg[10], cfg[9]
scale = np.concatenate([np.full(N, umax), np.full(N - 1, dmax)])
st = initial_states(4096, 0)
pick = np.sort(np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1).choice(4096, NS, replace=False))


def qp_at(state8, lap, pp):
    wp, sp, _, _ = orc.skidpad_waypoints(state8[0], state8[1], np.arctan2(state8[3], state8[2]), lap, pp, 8.0, N, dt)
    red = np.array([state8[0], state8[1], state8[2], state8[3], state8[5], state8[6], state8[7]])
    return O.condensed_qp(cfg, red, wp, sp)


def solve_set(T, Uu, vu, lo, up, act):
    rows = sorted(act)
    if not rows:
        return Uu.copy(), {}
    b = np.array([up[r] if act[r] > 0 else lo[r] for r in rows])
    nu = np.linalg.lstsq(T[np.ix_(rows, rows)], vu[rows] - b, rcond=1e-13)[0]
    return Uu - T[rows][:, :N].T @ nu, dict(zip(rows, nu))


def in_run_rule(act):
    """a steering bound whose predecessor is active on the same side fixes that stage's input: its rate bound leaves"""
    for r in list(act):
        if r >= N and r - N >= 1 and act.get(r - 1) == act[r] and (r - N) in act:
            del act[r - N]
    return act


def exchange(T, A, Uu, lo, up, act, cap=30):
    """one bound leaves (largest wrong-signed multiplier in cost units), one enters (most violated), per iteration"""
    vu = A @ Uu
    for it in range(cap):
        U, nu = solve_set(T, Uu, vu, lo, up, act)
        v = A @ U
        bad = {r: abs(x) * scale[r] for r, x in nu.items() if not (x * act[r] >= 0)}
        viol = np.maximum(v - up, lo - v) / scale
        viol[list(act)] = -1
        cand = np.where(viol > 1e-12)[0]
        if not bad and len(cand) == 0:
            return U, act, it + 1
        if bad:
            del act[max(bad, key=bad.get)]
        if len(cand):
            r = cand[np.argmax(viol[cand])]
            act[r] = 1 if v[r] > up[r] else -1
    return U, act, cap


def sh(x, fill):
    """row i+1 of each kind at row i"""
    out = np.empty_like(x)
    out[:N - 1] = x[1:N]; out[N - 1] = fill
    out[N:2 * N - 2] = x[N + 1:]; out[2 * N - 2] = fill
    return out


def continue_row(a_t, nu_t, s_t, a_p, nu_p, s_p):
    if a_t != 0:
        return a_t if (a_p != a_t or (2 * nu_t - nu_p) * a_t > 0) else 0
    return 7 if (a_p == 0 and not (2 * s_t - s_p > 0)) else 0       # 7: becomes active


def predict(mode, h1, h0, truth):
    a1, nu1, s1 = h1
    A_, B_ = sh(a1, 0), a1
    pred = A_.copy()
    if mode == "shift" or (mode == "and" and h0 is None):
        return pred
    if mode == "best2":
        for i in range(2 * N - 1):
            if truth[i] in (A_[i], B_[i]):
                pred[i] = truth[i]
        return pred
    a0, nu0, s0 = h0
    nuA1, sA1 = sh(nu1, 0.0), sh(s1, 1.0)
    aA0, nuA0, sA0 = sh(sh(a0, 0), 0), sh(sh(nu0, 0.0), 0.0), sh(sh(s0, 1.0), 1.0)
    for i in range(2 * N - 1):
        if A_[i] == B_[i]:
            continue
        pa = continue_row(A_[i], nuA1[i], sA1[i], aA0[i], nuA0[i], sA0[i])
        pb = continue_row(B_[i], nu1[i], s1[i], a0[i], nu0[i], s0[i])
        p = pa if (pa != 0 and pb != 0) else 0
        pred[i] = (A_[i] if A_[i] != 0 else B_[i]) if p == 7 else p
    return pred


res = {m: collections.Counter() for m in MODES}
G = T = None
for si in pick:
    s8 = st[si:si + 1].copy(); laps = np.zeros(1, np.int32); pp = np.zeros(1)
    orc.skidpad_run(cfg, s8, laps, pp, 8.0, 40, record=False)
    state, lap, prevp = s8[0].copy(), int(laps[0]), float(pp[0])
    hist = []
    for step in range(40, 70):
        H, g, A, lo, up = qp_at(state, lap, prevp)
        if G is None:
            G = np.linalg.inv(H); T = A @ G @ A.T
        Uu = -G @ g; vu = A @ Uu
        if not ((vu > up + 1e-12).any() or (vu < lo - 1e-12).any()):
            hist = []
        else:
            cold = {int(r): (1 if vu[r] > up[r] else -1) for r in np.where((vu > up) | (vu < lo))[0]}
            start = cold if not hist else in_run_rule({int(i): int(v) for i, v in enumerate(sh(hist[-1][0], 0)) if v != 0})
            U, act, its0 = exchange(T, A, Uu, lo, up, dict(start))
            truth = np.zeros(2 * N - 1, int)
            for r, s in act.items():
                truth[r] = s
            for m in MODES:
                its = its0
                if hist:
                    p = predict(m, hist[-1], hist[-2] if len(hist) > 1 else None, truth)
                    _, _, its = exchange(T, A, Uu, lo, up, in_run_rule({int(i): int(v) for i, v in enumerate(p) if v != 0}))
                if step >= 50:
                    res[m][its] += 1
            _, nud = solve_set(T, Uu, vu, lo, up, act)
            nu = np.zeros(2 * N - 1)
            for r, x in nud.items():
                nu[r] = x * scale[r]
            v = A @ U
            hist.append((truth, nu, np.minimum(up - v, v - lo) / scale))
        _, state, lap, prevp, _, _ = orc.skidpad_step(cfg, state, lap, prevp, 8.0)
for m in MODES:
    c = res[m]; tot = sum(c.values())
    print(f"{m:6s} solves {tot}  mean iterations {sum(k * v for k, v in c.items()) / max(tot, 1):.2f}  {dict(sorted(c.items()))}")
