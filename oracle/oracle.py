"""ctypes front-end of the CPU checker (oracle/mpc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under mpc_amd/ imports this module.

The controller set-up restated here (tyre stiffness, cost matrices, DARE
terminal weight) follows LMPC2.py:98-112,220-260,349-440, OFLMPC2.py:274-279
and LPVMPC.py:191-226 of the reference.  QP parity is unpinned (see the header
of mpc_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.linalg
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
KIND = {"L": 0, "OFL": 1, "LPV": 2, "NL": 3}
CFG_LEN = 74

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> None:
    """Compile both precisions of the checker with gcc (a few seconds)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))      # make rebuilds only what is stale


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Oracle:
    """One precision of the checker: 'f64' (double) or 'f80' (x87 long double)."""

    def __init__(self, precision: str = "f64"):
        path = os.path.join(_BUILD, f"liboracle_{precision}.so")
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.orc_real_bytes.restype = C.c_int
        L.orc_tyre_stiffness.argtypes = [C.c_double, C.c_double, _dp]
        L.orc_plant_f.argtypes = [_dp, C.c_double, C.c_int, _dp, _dp]
        L.orc_plant_rk4.argtypes = [_dp, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_ekf_jacobian.argtypes = [_dp, C.c_double, _dp, _dp]
        L.orc_ekf_time_update.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp, _dp]
        L.orc_ekf_meas_update.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_skidpad_waypoints.argtypes = [C.c_double] * 3 + [C.c_int, _dp, C.c_double, C.c_int, C.c_double, _dp, _dp]
        L.orc_skidpad_waypoints.restype = C.c_double
        L.orc_skidpad_waypoints_profile.argtypes = ([C.c_double] * 3 + [C.c_int, _dp, C.c_double, C.c_int, C.c_double, _dp, _dp,
                                                                        _dp])
        L.orc_skidpad_waypoints_profile.restype = C.c_double
        L.orc_progress2speed.argtypes = [_dp, C.c_double, C.c_double]
        L.orc_progress2speed.restype = C.c_double
        L.orc_step_waypoints.argtypes = [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double, _dp, _dp, _dp]
        L.orc_step_waypoints.restype = C.c_double
        L.orc_mpc_optimize.argtypes = [_dp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                       C.c_double, C.c_int]
        L.orc_mpc_optimize.restype = C.c_int
        L.orc_nl_optimize.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int]
        L.orc_nl_optimize.restype = C.c_int
        L.orc_skidpad_step.argtypes = [_dp, _dp, _ip, _dp, C.c_double, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
                                       C.c_int]
        L.orc_skidpad_step.restype = C.c_int
        L.orc_skidpad_run.argtypes = [_dp, _dp, _ip, _dp, C.c_double, C.c_int, C.c_int, _dp, C.c_double, C.c_double,
                                      C.c_int]
        L.orc_skidpad_run.restype = C.c_long
        L.orc_skidpad_run_cl.argtypes = L.orc_skidpad_run.argtypes + [_dp, _dp, _dp]
        L.orc_skidpad_run_cl.restype = C.c_long
        L.orc_ofl_run.argtypes = [_dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp,
                                  _ip, C.c_double, C.c_double, C.c_int]
        L.orc_ofl_run.restype = C.c_long
        L.orc_ofl_run_cl.argtypes = L.orc_ofl_run.argtypes + [_dp]
        L.orc_ofl_run_cl.restype = C.c_long
        L.orc_set_unconstrained_first.argtypes = [C.c_int]
        L.orc_set_terminal_set.argtypes = [C.c_int, _dp]
        L.orc_set_terminal_set.restype = C.c_int
        L.orc_mpc_optimize_sqp_step.argtypes = [_dp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp,
                                                C.c_double, C.c_double, C.c_int]
        L.orc_mpc_optimize_sqp_step.restype = C.c_int
        self.precision = precision
        # default solver tolerances: as tight as the arithmetic allows
        self.tol = (1e-13, 1e-15, 100) if precision == "f64" else (1e-17, 1e-18, 200)

    # -- set-up ---------------------------------------------------------
    def tyre_stiffness(self, m, x_cg):
        out = np.zeros(2)
        self.lib.orc_tyre_stiffness(float(m), float(x_cg), _ptr(out))
        return float(out[0]), float(out[1])

    # -- plant / EKF ----------------------------------------------------
    def plant_f(self, x, u, mp):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros_like(x)
        self.lib.orc_plant_f(_ptr(x), float(u), len(x), _ptr(np.ascontiguousarray(mp, dtype=np.float64)), _ptr(out))
        return out

    def plant_rk4(self, x, u, dt, mp):
        x = np.array(x, dtype=np.float64)
        self.lib.orc_plant_rk4(_ptr(x), float(u), float(dt), len(x), _ptr(np.ascontiguousarray(mp, dtype=np.float64)))
        return x

    def ekf_jacobian(self, x, dt, mp):
        F = np.zeros((10, 10))
        self.lib.orc_ekf_jacobian(_ptr(np.ascontiguousarray(x, dtype=np.float64)), float(dt),
                                  _ptr(np.ascontiguousarray(mp, dtype=np.float64)), _ptr(F))
        return F

    def ekf_time_update(self, x, P, u, dt, mp, Qd):
        x = np.array(x, dtype=np.float64); P = np.array(P, dtype=np.float64)
        self.lib.orc_ekf_time_update(_ptr(x), _ptr(P), float(u), float(dt),
                                     _ptr(np.ascontiguousarray(mp, dtype=np.float64)),
                                     _ptr(np.ascontiguousarray(Qd, dtype=np.float64)))
        return x, P

    def ekf_meas_update(self, x, P, y, Rd):
        x = np.array(x, dtype=np.float64); P = np.array(P, dtype=np.float64)
        self.lib.orc_ekf_meas_update(_ptr(x), _ptr(P), _ptr(np.ascontiguousarray(y, dtype=np.float64)),
                                     _ptr(np.ascontiguousarray(Rd, dtype=np.float64)))
        return x, P

    # -- planners ---------------------------------------------------------
    def skidpad_waypoints(self, x, y, heading, lap, prev_progress, v, N, dt):
        wp = np.zeros((N + 1, 4)); sp = np.zeros(N + 1); pp = np.array([prev_progress], dtype=np.float64)
        p0 = self.lib.orc_skidpad_waypoints(float(x), float(y), float(heading), int(lap), _ptr(pp), float(v), int(N),
                                            float(dt), _ptr(wp), _ptr(sp))
        return wp, sp, p0, float(pp[0])

    def skidpad_waypoints_profile(self, x, y, heading, lap, prev_progress, v, N, dt, zones):
        """request_waypoints with the speed profile for laps outside 0..7; zones = planner_zones(...)."""
        wp = np.zeros((N + 1, 4)); sp = np.zeros(N + 1); pp = np.array([prev_progress], dtype=np.float64)
        z = np.ascontiguousarray(zones, dtype=np.float64)
        p0 = self.lib.orc_skidpad_waypoints_profile(float(x), float(y), float(heading), int(lap), _ptr(pp), float(v),
                                                    int(N), float(dt), _ptr(z), _ptr(wp), _ptr(sp))
        return wp, sp, p0, float(pp[0])

    def progress2speed(self, zones, v, q):
        return self.lib.orc_progress2speed(_ptr(np.ascontiguousarray(zones, dtype=np.float64)), float(v), float(q))

    def step_waypoints(self, x, y, heading, v, N, dt, ramp_length=0.0):
        wp = np.zeros((N + 1, 4)); sp = np.zeros(N + 1); ab = np.zeros((N + 1, 4))
        p0 = self.lib.orc_step_waypoints(float(x), float(y), float(heading), float(v), int(N), float(dt),
                                         float(ramp_length), _ptr(wp), _ptr(sp), _ptr(ab))
        return wp, sp, p0, ab

    # -- controllers ------------------------------------------------------
    def optimize(self, cfg, x0_7, wp, p, d_hat=0.0, lin_x=None, lin_u=None, tol=None):
        N = int(cfg[1])
        traj = np.zeros((N + 1, 7)); U = np.zeros(N); lam = np.zeros(2 * N); info = np.zeros(4)
        tol = tol or self.tol
        st = self.lib.orc_mpc_optimize(_ptr(cfg), _ptr(np.ascontiguousarray(x0_7, dtype=np.float64)),
                                       _ptr(np.ascontiguousarray(wp, dtype=np.float64)),
                                       _ptr(np.ascontiguousarray(p, dtype=np.float64)), float(d_hat), _ptr(lin_x),
                                       _ptr(lin_u), _ptr(traj), _ptr(U), _ptr(lam), _ptr(info), tol[0], tol[1], tol[2])
        return st, traj, U, lam, info

    def optimize_sqp_step(self, cfg, x0_7, wp, p, x_prev, reg_eps=1e-4, d_hat=0.0, lin_x=None, lin_u=None, tol=(1e-8, 1e-8, 50)):
        """One SQP iteration of the reference's acados set-up from the iterate x_prev ((N+1,6) states): the QP with the
        MIRROR-regularised Hessian, solved to HPIPM's default accuracy.  Returns (status, traj, U, x (N+1,6), info)."""
        N = int(cfg[1])
        traj = np.zeros((N + 1, 7)); U = np.zeros(N); xo = np.zeros((N + 1, 6)); info = np.zeros(4)
        st = self.lib.orc_mpc_optimize_sqp_step(_ptr(cfg), _ptr(np.ascontiguousarray(x0_7, dtype=np.float64)),
                                                _ptr(np.ascontiguousarray(wp, dtype=np.float64)),
                                                _ptr(np.ascontiguousarray(p, dtype=np.float64)), float(d_hat), _ptr(lin_x),
                                                _ptr(lin_u), _ptr(np.ascontiguousarray(x_prev, dtype=np.float64)), float(reg_eps),
                                                _ptr(xo), _ptr(traj), _ptr(U), _ptr(info), tol[0], tol[1], int(tol[2]))
        return st, traj, U, xo, info

    def nl_optimize(self, cfg, x0_7, wp, p, U_warm=None, tol=None):
        """NLOcp.optimize; U_warm is the iterate kept from the previous call (None = zeros).  Returns
        (status, traj (N+1,7), U (N,), lam (2N,), info [qp iterations, sqp iterations, last step, defect])."""
        N = int(cfg[1])
        traj = np.zeros((N + 1, 7)); lam = np.zeros(2 * N); info = np.zeros(4)
        U = np.zeros(N) if U_warm is None else np.array(U_warm, dtype=np.float64).reshape(N)
        tol = tol or self.tol
        st = self.lib.orc_nl_optimize(_ptr(cfg), _ptr(np.ascontiguousarray(x0_7, dtype=np.float64)),
                                      _ptr(np.ascontiguousarray(wp, dtype=np.float64)),
                                      _ptr(np.ascontiguousarray(p, dtype=np.float64)), _ptr(U), _ptr(traj), _ptr(lam),
                                      _ptr(info), tol[0], tol[1], tol[2])
        return st, traj, U, lam, info

    def skidpad_step(self, cfg, state8, lap, prev_progress, target_vel, lin_x=None, lin_u=None, tol=None):
        tol = tol or self.tol
        st8 = np.array(state8, dtype=np.float64); lp = C.c_int(int(lap)); pp = np.array([prev_progress], dtype=np.float64)
        u0 = np.zeros(1); info = np.zeros(4)
        st = self.lib.orc_skidpad_step(_ptr(cfg), _ptr(st8), C.byref(lp), _ptr(pp), float(target_vel), _ptr(lin_x),
                                       _ptr(lin_u), _ptr(u0), _ptr(info), tol[0], tol[1], tol[2])
        return st, st8, lp.value, float(pp[0]), float(u0[0]), info

    def skidpad_run(self, cfg, states, laps, prev_progress, target_vel, nsteps, record=True, tol=None, lin_x=None, lin_u=None,
                    cl_stats=None):
        """states (B,8) / laps (B,) int32 / prev_progress (B,) are updated in place.  lin_x (B,6,N+1) / lin_u (B,N): the
        controllers' carried memory (LPV prev_x / prev_u, NL iterate in lin_u), read and updated, so that a run can be
        continued by the next call; cl_stats (B,8): receives the closed-loop statistics of this call (CL_KEYS)."""
        tol = tol or self.tol
        B = states.shape[0]
        assert states.flags.c_contiguous and laps.dtype == np.int32
        for a in (lin_x, lin_u, cl_stats):
            assert a is None or (a.flags.c_contiguous and a.dtype == np.float64)
        uh = np.zeros((nsteps, B)) if record else None
        iters = self.lib.orc_skidpad_run_cl(_ptr(cfg), _ptr(states), laps.ctypes.data_as(_ip), _ptr(prev_progress),
                                            float(target_vel), B, int(nsteps), _ptr(uh), tol[0], tol[1], tol[2], _ptr(lin_x),
                                            _ptr(lin_u), _ptr(cl_stats))
        return uh, int(iters)


    def ofl_run(self, cfg, x10, est, P, target_vel, nsteps, noise=None, ramp_length=0.0, record=True, tol=None, cl_stats=None):
        """simulator.py:206-281 for B scenarios: x10 / est (B,10) and P (B,10,10) are updated in place; noise (n,B,7)
        is added to the measurements; cl_stats (B,8) receives the closed-loop statistics (CL_KEYS).
        Returns (u (n,B), x (n,B,10), est (n,B,10), status (B,), iterations)."""
        tol = tol or self.tol
        B = x10.shape[0]
        assert x10.flags.c_contiguous and est.flags.c_contiguous and P.flags.c_contiguous
        uh = np.zeros((nsteps, B)) if record else None
        xh = np.zeros((nsteps, B, 10)) if record else None
        eh = np.zeros((nsteps, B, 10)) if record else None
        status = np.zeros(B, np.int32)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float64)
        assert nz is None or nz.shape == (nsteps, B, 7)
        assert cl_stats is None or (cl_stats.flags.c_contiguous and cl_stats.shape == (B, 8))
        iters = self.lib.orc_ofl_run_cl(_ptr(cfg), _ptr(x10), _ptr(est), _ptr(P), float(target_vel), B, int(nsteps), _ptr(nz),
                                        _ptr(EKF_Q_DIAG), _ptr(EKF_R_DIAG), float(ramp_length), _ptr(uh), _ptr(xh), _ptr(eh),
                                        status.ctypes.data_as(_ip), tol[0], tol[1], tol[2], _ptr(cl_stats))
        return uh, xh, eh, status, int(iters)

    def set_terminal_set(self, C=None):
        """Terminal set C (x_N[1:] - x_ref) <= 1 for every following solve of the linear controllers (LMPC2.py:212-218,
        commented out in the reference); None or an empty array removes it."""
        C = np.zeros((0, 5)) if C is None else np.ascontiguousarray(C, dtype=np.float64)
        assert C.ndim == 2 and C.shape[1] == 5
        if self.lib.orc_set_terminal_set(int(C.shape[0]), _ptr(C)) != 0:
            raise ValueError("too many terminal rows")

    def set_unconstrained_first(self, on: bool):
        """Timed CPU baseline only: try the unconstrained minimiser (one Riccati pass) before the interior point."""
        self.lib.orc_set_unconstrained_first(int(bool(on)))


# columns of the closed-loop statistics (cl_update in mpc_oracle.c)
CL_KEYS = ("stage_cost", "lateral_error_sq", "heading_pair_violation", "input_sq", "max_bound_violation",
           "max_heading_pair_violation", "max_predicted_heading_pair_violation", "steps")


# EKF.py:28-32: P0, Q and R = diag(sigma) of CarEKF
EKF_P0_DIAG = np.array([1, 1, 1, 1, 1, 10.0, 1.0, 1.0, 0.0, 1000000.0])
EKF_Q_DIAG = np.array([0.05, 0.05, 0.01, 0.01, 0.001, 0.1, 0.01, 0.01, 0.0, 0.0])
EKF_R_DIAG = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])


# ---------------------------------------------------------------------------
# controller configuration (host-side restatement of the reference constructors)
# ---------------------------------------------------------------------------
_DEFAULT_CFG_DIR = os.path.join(os.path.dirname(_HERE), "mpc_amd", "config")


def planner_zones(target_vel, max_accel=10.0, slowdown_speed_factor=0.5, start_progress=None):
    """The planner's zone edges (utils/path_planning.py:63-91): accel 1 start/end, braking 1 start/end, braking 2
    start/end, and the slow-down factor; start_progress = the argument of set_starting_position, if it was called."""
    r, center, end_straight = 9.125, 16.75, 15
    lap_length = 2 * r * np.pi
    accel_time = target_vel / max_accel
    a1s = 0 if start_progress is None else start_progress
    a1e = max_accel / 2 * accel_time * accel_time + (0 if start_progress is None else start_progress)
    decel_time = accel_time * (1 - slowdown_speed_factor)
    b1s = center + 1 + 4 * lap_length
    b1e = b1s + decel_time * target_vel + max_accel / 2 * decel_time * decel_time
    decel_time_2 = accel_time * slowdown_speed_factor
    b2s = center + end_straight + 4 * lap_length
    b2e = b2s + decel_time_2 * max_accel
    return np.array([a1s, a1e, b1s, b1e, b2s, b2e, slowdown_speed_factor], dtype=np.float64)


def load_params(kind: str, path: str | None = None) -> dict:
    name = {"L": "parameters_L.yaml", "OFL": "parameters_L.yaml", "LPV": "parameters_LPV.yaml",
            "NL": "parameters_NL.yaml"}[kind]
    with open(path or os.path.join(_DEFAULT_CFG_DIR, name)) as fh:
        return yaml.safe_load(fh)


def model_params(params: dict, orc: Oracle | None = None) -> np.ndarray:
    m = params["model"]
    lf = m["x_cg"] * m["wbase"]; lr = (1 - m["x_cg"]) * m["wbase"]
    Cf, Cr = (orc or Oracle()).tyre_stiffness(m["m"], m["x_cg"])
    return np.array([m["m"], m["I_z"], lf, lr, Cf, Cr], dtype=np.float64)


def l_model_continuous(mp, vx, eps=1e-3):
    """A_c(vx) of the linear model, LMPC2.py:127-149 (a12 divides the whole bracket)."""
    m, Iz, lf, lr, Cf, Cr = mp
    A = np.zeros((6, 6))
    A[1, 2] = vx; A[1, 3] = 1.0; A[2, 4] = 1.0
    A[3, 3] = -(Cf + Cr) / (m * vx + eps)
    A[3, 4] = (-vx + (Cr * lr - Cf * lf)) / (m * vx + eps)
    A[3, 5] = -Cf / m
    A[4, 3] = (lr * Cr - lf * Cf) / (Iz * vx + eps)
    A[4, 4] = -(lf * lf * Cf + lr * lr * Cr) / (Iz * vx + eps)
    A[4, 5] = -lf * Cf / Iz
    return A


def dare_terminal(mp, dt, W):
    """LMPC2.py:349-440: float32-cast forward-Euler (A,B) at vx=15 without px, Q'=diag(W11,W22,0,0,W33),
    R'=W[4,4]; returns (P, K, A5, B5)."""
    A = (np.eye(6) + dt * l_model_continuous(mp, 15.0)).astype(np.float32)[1:, 1:]
    B = np.zeros((6, 1)); B[5, 0] = dt
    B = B.astype(np.float32)[1:, :]
    Q = np.diag([W[1, 1], W[2, 2], 0.0, 0.0, W[3, 3]])
    R = np.array([[W[4, 4]]])
    Ad, Bd = A.astype(np.float64), B.astype(np.float64)
    P = scipy.linalg.solve_discrete_are(Ad, Bd, Q, R)
    K = np.linalg.solve(Bd.T @ P @ Bd + R, Bd.T @ P @ Ad)
    return P, K, A, B


def make_cfg(kind: str, N: int, Tf: float, params: dict | None = None, orc: Oracle | None = None,
             lin_mode: str = "reference", sqp_tol: float = 1e-11, sqp_max_iter: int = 50) -> np.ndarray:
    params = params or load_params(kind)
    mp = model_params(params, orc)
    dt = Tf / N
    if kind == "NL":
        # NLMPC.py:159-186: weights are hard-coded (the yaml's controller block is not read), W_e = 100 Q
        Q = np.array([0, 1e5, 1e-10, 1e0, 0, 1e0, 1e0])
        cfg = np.zeros(CFG_LEN)
        cfg[0] = KIND[kind]; cfg[1] = N; cfg[2] = dt; cfg[3:9] = mp
        cfg[9] = params["model"]["max_steering_angle"]; cfg[10] = params["model"]["max_steering_rate"]
        cfg[56:63] = Q; cfg[63] = 10.0; cfg[64:71] = 100 * Q; cfg[71] = 0.1
        cfg[72] = sqp_tol; cfg[73] = sqp_max_iter
        return cfg
    c = params["controller"]
    W = np.zeros((7, 7))
    W[:6, :6] = np.array(c["Q"], dtype=np.float64) * c["q"]
    W[6:, 6:] = np.array(c["R"], dtype=np.float64) * c["r"]
    We = np.zeros((6, 6))
    if kind == "LPV":
        We[:, :] = W[:6, :6]                       # LPVMPC.py:217: W_e = W
    else:
        P, _, _, _ = dare_terminal(mp, dt, W)
        beta = c["beta"] if kind == "L" else 1.0   # LMPC2.py:260 vs OFLMPC2.py:279
        We[1:, 1:] = beta * 0.5 * P
    cfg = np.zeros(CFG_LEN)
    cfg[0] = KIND[kind]; cfg[1] = N; cfg[2] = dt; cfg[3:9] = mp
    cfg[9] = params["model"]["max_steering_angle"]; cfg[10] = params["model"]["max_steering_rate"]
    cfg[11:18] = np.diag(W); cfg[18:54] = We.reshape(-1); cfg[54] = 1e-3
    cfg[55] = 1.0 if lin_mode == "prev_iter" else 0.0
    return cfg


# ---------------------------------------------------------------------------
# dense KKT certificate of the condensed QP (numpy; independent of the C solver)
# ---------------------------------------------------------------------------
def condensed_qp(cfg, x0_7, wp, p, d_hat=0.0, terminal_C=None):
    """Dense (H, g, A, lo, up) of the L/OFL problem built by explicit prediction matrices; with `terminal_C` (m x 5) the
    rows C (x_N[1:] - x_ref) <= 1 are appended (lower bound -inf)."""
    kind, N, dt = int(cfg[0]), int(cfg[1]), cfg[2]
    assert kind in (0, 1)
    mp = cfg[3:9]
    wq = dt * cfg[11:17]; wr = dt * cfg[17]; QN = cfg[18:54].reshape(6, 6)
    xs = [np.array([0, 0, 0, x0_7[4], x0_7[5], x0_7[6]], dtype=np.float64)]
    G = np.zeros((N + 1, 6, N))
    b = np.zeros(6); b[5] = dt
    for k in range(N):
        A = np.eye(6) + dt * l_model_continuous(mp, p[k])
        c = np.zeros(6); c[0] = dt * p[k]
        if kind == 1:
            c[3] = dt * d_hat
        xs.append(A @ xs[-1] + c)
        G[k + 1] = A @ G[k]
        G[k + 1][:, k] = b
    xr = np.zeros((N + 1, 6)); xr[:, 0] = wp[:, 0]; xr[:, 1] = wp[:, 1]; xr[:, 2] = wp[:, 3]
    H = wr * np.eye(N); g = np.zeros(N)
    for k in range(1, N):
        H += G[k].T @ (wq[:, None] * G[k]); g += G[k].T @ (wq * (xs[k] - xr[k]))
    H += G[N].T @ QN @ G[N]; g += G[N].T @ QN @ (xs[N] - xr[N])
    T = np.tril(np.ones((N, N)))[: N - 1] * dt          # delta_{k+1} - delta_0, k = 0..N-2
    A = np.vstack([np.eye(N), T])
    lo = np.concatenate([-cfg[10] * np.ones(N), (-cfg[9] - x0_7[6]) * np.ones(N - 1)])
    up = np.concatenate([cfg[10] * np.ones(N), (cfg[9] - x0_7[6]) * np.ones(N - 1)])
    if terminal_C is not None and len(terminal_C):
        C = np.asarray(terminal_C, dtype=np.float64)
        A = np.vstack([A, C @ G[N][1:, :]])
        up = np.concatenate([up, 1.0 + C @ xr[N, 1:] - C @ xs[N][1:]])
        lo = np.concatenate([lo, np.full(C.shape[0], -np.inf)])
    return H, g, A, lo, up


def kkt_residuals(H, g, A, lo, up, U, lam):
    """lam: signed multipliers (positive at the upper bound).  Returns (stationarity, primal, complementarity, sign)."""
    v = A @ U
    stat = np.abs(H @ U + g + A.T @ lam).max()
    prim = max((v - up).max(), (lo - v).max(), 0.0)
    comp = np.abs(np.where(lam > 0, lam * (up - v), lam * (v - lo))).max()
    return stat, prim, comp


# ---------------------------------------------------------------------------
# stability.py restated (numpy).  Parity unpinned: the reference holds no fixture for these functions and cvxpy / ECOS,
# which its LP loop needs, are absent; the vertex test and the bisection below are deterministic linear algebra.
# ---------------------------------------------------------------------------
def invariance_check(P, f, c) -> bool:
    """stability.py:155-189: box around x'Px <= c in the eigenbasis of P, every vertex mapped through f and back."""
    eigvals, Q = np.linalg.eigh(P)
    bounds = np.sqrt(c / eigvals)
    vertices = np.array(np.meshgrid(*[[-b, b] for b in bounds])).T.reshape(-1, len(bounds))
    for vertex in vertices:
        vertex_next = Q.T @ f(Q @ vertex)
        if np.any(np.abs(vertex_next) > bounds):
            return False
    return True


def invariance_binary_search(P, f, c_u, epsilon=1e-1):
    """stability.py:192-211."""
    c_l, c_best = 0, 0
    if invariance_check(P, f, c_u):
        return c_u
    while c_u - c_l > epsilon:
        c = (c_l + c_u) / 2
        if invariance_check(P, f, c):
            c_best = c; c_l = c
        else:
            c_u = c
    return c_best


def reconstruct_invariant_set(n, K, A, B, ubx, ubu):
    """stability.py:93-109 without the plot and the file."""
    phi = A - B @ K
    F = np.concatenate((np.diag(1 / ubx), np.diag(-1 / ubx), np.zeros((2 * len(ubu), len(ubx)))), axis=0)
    G = np.concatenate((np.zeros((2 * len(ubx), len(ubu))), np.diag(1 / ubu), np.diag(-1 / ubu)), axis=0)
    C_initial = F + G @ K
    C = C_initial
    for _ in range(n):
        C_initial = C_initial @ phi
        C = np.concatenate((C, C_initial), axis=0)
    return C


def estimate_invariant_set(K, A, B, ubx, ubu, n_limit=1000):
    """stability.py:7-90 with scipy's HiGHS in place of cvxpy / ECOS (neither is importable here): the loop's quirks kept
    (first power constrained is phi^1; stop when every optimum is below 5e-2; an unbounded programme counts as +inf; a
    programme the solver cannot finish skips the rest of its group).  Returns (C, n, optima of the last round)."""
    from scipy.optimize import linprog
    K = np.atleast_2d(np.asarray(K, dtype=np.float64)); A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    ubx = np.asarray(ubx, dtype=np.float64).reshape(-1); ubu = np.asarray(ubu, dtype=np.float64).reshape(-1)
    phi = A - B @ K
    phi_np1 = phi.copy()
    rows, rhs = [], []
    n = 1

    def maximise(c):
        r = linprog(-c, A_ub=np.concatenate(rows), b_ub=np.concatenate(rhs), bounds=(None, None), method="highs")
        if r.status == 0:
            return float(-r.fun)
        if r.status == 3:
            return float("inf")
        raise RuntimeError(r.message)

    while True:
        M = phi_np1
        rows += [M, -M, K @ M, -(K @ M)]
        rhs += [ubx, ubx, ubu, ubu]
        n += 1
        phi_np1 = phi_np1 @ phi
        opt_values = []
        groups = [[phi_np1[j] / ubx[j] for j in range(len(ubx))], [(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))],
                  [-phi_np1[j] / ubx[j] for j in range(len(ubx))], [-(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))]]
        for group in groups:
            try:
                for c in group:
                    opt_values.append(maximise(c))
            except RuntimeError:
                pass
        if np.all(np.array(opt_values) < 5e-2) or n > n_limit:
            return reconstruct_invariant_set(n, K, A, B, ubx, ubu), n, opt_values


# ---------------------------------------------------------------------------
# acados emulation (study aid; pins nothing: acados itself cannot run here).  What the reference's solver set-up
# (LMPC2.py:262-288: SQP, EXACT Hessian, regularize_method MIRROR, MERIT_BACKTRACKING, tol 1e-4, iterate kept between
# calls) returns for the linear-quadratic controllers differs from the unique minimiser of the posed QP because (i) the
# QP handed to HPIPM carries reg_epsilon on the unweighted states, anchored at the previous iterate, (ii) the SQP loop
# stops as soon as the KKT residual of the original problem is below 1e-4, (iii) HPIPM stops at about 1e-8.
# ---------------------------------------------------------------------------
class AcadosEmulation:
    """Carries the iterate from call to call like AcadosOcpSolver does; optimize() runs SQP iterations until the
    stationarity residual the regularisation leaves behind, reg_eps * max|x - x_prev| over the regularised components,
    is below `tol` (the other KKT residuals of a full step on a linear-quadratic problem are at the QP solver's 1e-8)."""

    def __init__(self, orc: Oracle, cfg, reg_eps: float = 1e-4, tol: float = 1e-4, max_iter: int = 200, qp_tol=(1e-8, 1e-8, 50)):
        """qp_tol: accuracy of the QP solves (complementarity, residual, iteration cap); the default stands for HPIPM's,
        orc.tol drives every QP to the accuracy of the arithmetic (what the library's emulate_acados mode is compared with)."""
        self.orc, self.cfg, self.reg_eps, self.tol, self.max_iter, self.qp_tol = orc, cfg, reg_eps, tol, max_iter, qp_tol
        N = int(cfg[1])
        self.x_prev = np.zeros((N + 1, 6))          # acados starts from a zero iterate
        self.sqp_iterations = []

    def _regularised(self):
        cfg, dt = self.cfg, self.cfg[2]
        w_stage = np.abs(dt * cfg[11:17]) < self.reg_eps
        QN = cfg[18:54].reshape(6, 6)
        w_term = np.array([not (QN[i].any() or QN[:, i].any()) for i in range(6)])
        return w_stage, w_term

    def optimize(self, x0_7, wp, p, d_hat=0.0, lin_x=None, lin_u=None):
        w_stage, w_term = self._regularised()
        for it in range(1, self.max_iter + 1):
            st, traj, U, x, info = self.orc.optimize_sqp_step(self.cfg, x0_7, wp, p, self.x_prev, self.reg_eps, d_hat,
                                                             None if lin_x is None else lin_x.copy(),
                                                             None if lin_u is None else lin_u.copy(), tol=self.qp_tol)
            d = np.abs(x - self.x_prev)
            res = self.reg_eps * max(d[:-1][:, w_stage].max(initial=0.0), d[-1][w_term].max(initial=0.0))
            self.x_prev = x
            if res <= self.tol:
                break
        self.sqp_iterations.append(it)
        return st, traj, U
