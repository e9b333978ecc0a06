"""How far the reference's own solver answer can sit from the unique minimiser of the QP it poses: the acados emulation
of the CPU checker (oracle.AcadosEmulation: SQP with the MIRROR-regularised Hessian, stop at KKT 1e-4, iterate kept
between calls) against the exact minimiser, (a) per step on the same problems along the C1 closed loop, (b) as two closed
loops, one driven by each.  Study aid: it pins nothing (acados itself cannot run here).

  python tests/report_acados_emulation.py [steps=3400] [N=50] [Tf=0.5]      (uses the checker: lives with the tests)
"""
import sys

import numpy as np

sys.path.insert(0, '.')
from oracle import oracle as O


def run(steps=3400, N=50, Tf=0.5, kind="L", reg_eps=1e-4):
    orc = O.Oracle("f64")
    dt = Tf / N
    cfg = O.make_cfg(kind, N, Tf); mp = cfg[3:9]
    em_same = O.AcadosEmulation(orc, cfg, reg_eps)        # emulated solver on the problems of the exact-driven loop
    em_loop = O.AcadosEmulation(orc, cfg, reg_eps)        # emulated solver driving its own loop
    xa = np.array([15.0, 0.1, 1.0, 0.0, 8.0, 0, 0, 0]); la, pa = 0, 0.0          # skidpad_simulator.py:50-60
    xb = xa.copy(); lb, pb = 0, 0.0
    out = {"max_abs_du0": 0.0, "max_rel_du0": 0.0, "max_abs_dU": 0.0, "max_pos_div": 0.0, "max_yaw_div": 0.0}
    lin = (np.zeros((6, N + 1)), np.zeros(N)) if kind == "LPV" else (None, None)
    for t in range(steps):
        wp, sp, _, pa = orc.skidpad_waypoints(xa[0], xa[1], np.arctan2(xa[3], xa[2]), la, pa, xa[4], N, dt)
        x07 = xa[[0, 1, 2, 3, 5, 6, 7]]
        lx = None if lin[0] is None else lin[0].copy(); lu = None if lin[1] is None else lin[1].copy()
        _, _, Ue = em_same.optimize(x07, wp, sp, 0.0, lx, lu)
        _, _, U, _, _ = orc.optimize(cfg, x07, wp, sp, 0.0, lin[0], lin[1])
        d = abs(U[0] - Ue[0])
        out["max_abs_du0"] = max(out["max_abs_du0"], d)
        out["max_rel_du0"] = max(out["max_rel_du0"], d / max(abs(U[0]), 1e-3 * 0.9))
        out["max_abs_dU"] = max(out["max_abs_dU"], np.abs(U - Ue).max())
        old = xa[0]; xa = orc.plant_rk4(xa, U[0], dt, mp)
        la += int(xa[0] > 16.75 and old <= 16.75)
        if kind != "LPV":
            wpb, spb, _, pb = orc.skidpad_waypoints(xb[0], xb[1], np.arctan2(xb[3], xb[2]), lb, pb, xb[4], N, dt)
            _, _, Ub = em_loop.optimize(xb[[0, 1, 2, 3, 5, 6, 7]], wpb, spb)
            old = xb[0]; xb = orc.plant_rk4(xb, Ub[0], dt, mp)
            lb += int(xb[0] > 16.75 and old <= 16.75)
            out["max_pos_div"] = max(out["max_pos_div"], float(np.hypot(xa[0] - xb[0], xa[1] - xb[1])))
            out["max_yaw_div"] = max(out["max_yaw_div"], abs(float(np.arctan2(xa[3], xa[2]) - np.arctan2(xb[3], xb[2]))))
    out["sqp_iterations_histogram"] = np.bincount(em_same.sqp_iterations).tolist()
    out["final_lap"] = la
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    steps = int(a[0]) if a else 3400
    N = int(a[1]) if len(a) > 1 else 50
    Tf = float(a[2]) if len(a) > 2 else 0.5
    for kind in ("L", "LPV"):
        print(kind, N, Tf, run(steps if kind == "L" else min(steps, 600), N if kind == "L" else 80, Tf if kind == "L" else 0.8, kind))
