"""Host side of the MPC controllers: the reference's controller classes (LOcp, OFLOcp, LPVOcp) with the
acados solver object replaced by a handle of libmpc_amd.so.

Set-up (yaml, tyre stiffness, cost matrices, DARE terminal weight) is host arithmetic done once, as in the
reference's constructors (LMPC2.py:14-96,349-440; OFLMPC2.py:16-100; LPVMPC.py:10-92).  Every optimize()
runs on the GPU through the C ABI; there is no CPU solve path.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import time

import numpy as np
import scipy.linalg
import torch
import yaml

from . import _dev, _lib

log = logging.getLogger("mpc_amd")
_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")

# cornering stiffness over vertical load (LMPC2.py:98-112)
_C_DATA_Y = np.array([1.537405752168591e04, 2.417765976460659e04, 3.121158998819641e04, 3.636055041362088e04])
_C_DATA_X = [300, 500, 700, 900]


def tyre_stiffness(m: float, x_cg: float) -> tuple[float, float]:
    Cf = np.interp((9.81 * m / 2) * (1 - x_cg), _C_DATA_X, _C_DATA_Y)
    Cr = np.interp((9.81 * m / 2) * x_cg, _C_DATA_X, _C_DATA_Y)
    return float(Cf), float(Cr)


def read_parameters(filename: str, params=None) -> dict:
    """The reference opens parameters_*.yaml relative to the working directory (LMPC2.py:31); the same file is
    looked up there first, then the copy shipped in mpc_amd/config.  `params` may be a dict or a path."""
    if isinstance(params, dict):
        return params
    path = params if isinstance(params, str) else (filename if os.path.exists(filename) else os.path.join(_CONFIG_DIR, filename))
    with open(path, "r") as fh:
        return yaml.safe_load(fh)


class SolverHandle:
    """Owns one mpc_handle; stands where the reference keeps its AcadosOcpSolver (`ocp.solver`)."""

    def __init__(self, cfg: _lib.MpcConfig):
        self._lib = _lib.load()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self._lib.mpc_create(C.byref(cfg), C.byref(h))
        _lib.check(rc, None, "mpc_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._lib.mpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _CondensedMpc:
    """Shared machinery of the three quadratic controllers."""

    kind = "L"
    param_file = "parameters_L.yaml"
    n_states = 6
    n_inputs = 1
    n_outputs = 5

    def __init__(self, N: int, Tf: float, discrete: bool = True, debug: bool = False, *, params=None, device=None,
                 batch: int = 1, tol_mu: float = 1e-15, tol_res: float = 1e-12, max_iter: int = 80,
                 ipm_warm_start: bool = True, step_ramp_length: float = 0.0, solver_layout: str = "auto",
                 terminal_set=None, qp_method: str = "auto", emulate_acados: bool = False, reg_epsilon: float = 1e-4,
                 sqp_kkt_tol: float = 1e-4, sqp_max_iter: int = 200, _host_only: bool = False):
        if not discrete and self.kind != "LPV":
            # The reference's constructors do not get past their own stability analysis with discrete=False: that
            # branch keeps the 6x6 continuous-time A next to the 5x5 Q (LMPC2.py:370-382,406,411; OFLMPC2.py:389-402,
            # 424,429) and control.ctrb raises ValueError on the shapes, before any solver exists. The LPV controller
            # stores the flag and never reads it (LPVMPC.py:33), so there it changes nothing.
            raise ValueError("discrete=False: the reference's constructor fails here as well (6x6 continuous A against "
                             "the 5x5 Q in its stability analysis); only the forward-Euler model exists")
        log.setLevel(logging.DEBUG if debug else logging.INFO)
        self.params = read_parameters(self.param_file, params)
        self.discrete = discrete
        self.N = int(N)
        self.Tf = float(Tf)
        self.dt = self.Tf / self.N
        m = self.params["model"]
        self.m, self.I_z, self.wbase, self.x_cg = m["m"], m["I_z"], m["wbase"], m["x_cg"]
        self.lf = self.x_cg * self.wbase
        self.lr = (1 - self.x_cg) * self.wbase
        self.Cf, self.Cr = self.get_tyre_stiffness()
        self.max_steering = m["max_steering_angle"]
        self.max_steering_rate = m["max_steering_rate"]
        self.x_lin_point = np.zeros(6)
        self.u_lin_point = np.zeros(1)
        self.p_lin_point = np.array([15.0])
        self.W, self.W_e = self._cost_matrices()
        self.batch = int(batch)
        self._tol = (tol_mu, tol_res, max_iter)
        self._ipm_cold = 0 if ipm_warm_start else 1
        # the step-reference closed loops generate their waypoints on the GPU: StepPlanner's ramp_length travels with
        # the controller definition (utils/step_planning.py:51)
        self._step_ramp = float(step_ramp_length)
        # "wave": one wavefront per scenario on the condensed problem; "lane": one lane per scenario, stage-wise Riccati
        # interior point; "auto": lane where it is the faster one (L / LPV at horizons without a register-resident wave
        # kernel, from 2048 scenarios up), wave otherwise
        self._layout = _lib.LAYOUT[solver_layout]
        # "auto": constrained solves of the fixed-model path go through the active-set method first (interior point where
        # it does not end at the minimiser); "interior_point": interior point only; "active_set": the same as "auto"
        self._qp_method = _lib.QP_METHOD[qp_method]
        # emulate_acados: not the minimiser of the posed problem but what the reference's solver options make of it (MIRROR
        # regularisation about the iterate kept between calls, SQP steps until the KKT tolerance; LMPC2.py:262-288)
        self._emulate = (bool(emulate_acados), float(reg_epsilon), float(sqp_kkt_tol), int(sqp_max_iter))
        if terminal_set is not None and solver_layout == "auto":
            self._layout = _lib.LAYOUT["lane"]          # general terminal rows are handled by the stage-wise solver
        if _host_only:                      # set-up arithmetic only (used by the CPU tests); no solve is possible
            self.device = None
            self.solver = None
        else:
            self.device = _dev.require_gpu(device)
            self.solver = SolverHandle(self._make_config())
            if terminal_set is not None:
                self.set_terminal_set(terminal_set)
        self.metrics = {"runtime": []}
        self._bufs = {}
        self._auto_cache = self.batch == 1 and not emulate_acados          # optimize() of a single controller (batched users call condense_fixed themselves)
        self._cache_candidate = self._cached_vx = None

    # ---- set-up -----------------------------------------------------------------------------
    def get_tyre_stiffness(self) -> tuple[float, float]:
        return tyre_stiffness(self.m, self.x_cg)

    def model_params(self) -> np.ndarray:
        return np.array([self.m, self.I_z, self.lf, self.lr, self.Cf, self.Cr], dtype=np.float64)

    def continuous_A(self, vx: float) -> np.ndarray:
        """Jacobian of the linear model at speed vx (LMPC2.py:127-157)."""
        A = np.zeros((6, 6))
        A[1, 2] = vx; A[1, 3] = 1.0; A[2, 4] = 1.0
        A[3, 3] = -(self.Cf + self.Cr) / (self.m * vx + 0.001)
        A[3, 4] = (-vx + (self.Cr * self.lr - self.Cf * self.lf)) / (self.m * vx + 0.001)
        A[3, 5] = -self.Cf / self.m
        A[4, 3] = (self.lr * self.Cr - self.lf * self.Cf) / (self.I_z * vx + 0.001)
        A[4, 4] = -(self.lf * self.lf * self.Cf + self.lr * self.lr * self.Cr) / (self.I_z * vx + 0.001)
        A[4, 5] = -self.lf * self.Cf / self.I_z
        return A

    def _stage_weights(self) -> np.ndarray:
        c = self.params["controller"]
        Q = np.array(c["Q"], dtype=np.float64) * c["q"]
        R = np.array(c["R"], dtype=np.float64) * c["r"]
        if np.any(Q - np.diag(np.diag(Q))):
            raise NotImplementedError("off-diagonal state weights are not supported by the condensing kernel")
        W = np.zeros((7, 7))
        W[:6, :6] = Q
        W[6:, 6:] = R
        return W

    def _cost_matrices(self):
        W = self._stage_weights()
        return W, W.copy()

    def _make_config(self) -> _lib.MpcConfig:
        cfg = _lib.MpcConfig()
        cfg.kind = _lib.KIND[self.kind]
        cfg.N = self.N
        cfg.B = self.batch
        cfg.device = self.device.index if self.device is not None else 0
        cfg.dt = self.dt
        cfg.mp[:] = list(self.model_params())
        cfg.max_steering = self.max_steering
        cfg.max_rate = self.max_steering_rate
        cfg.w[:] = list(np.diag(self.W))
        cfg.We[:] = list(np.asarray(self.W_e)[:6, :6].reshape(-1))
        cfg.eps = 0.001
        cfg.tol_mu, cfg.tol_res, cfg.max_iter = self._tol
        cfg.lpv_lin_prev = 0
        cfg.ipm_cold_start = self._ipm_cold
        cfg.step_ramp_length = self._step_ramp
        cfg.solver_layout = self._layout
        cfg.qp_method = self._qp_method
        cfg.emulate_acados, cfg.reg_epsilon, cfg.sqp_kkt_tol, sqp_max = self._emulate
        if cfg.emulate_acados:
            cfg.sqp_max_iter = sqp_max                  # LMPC2.py:281 nlp_solver_max_iter = 200
        return cfg

    def stability(self):
        """DARE terminal weight exactly as LMPC2.py:349-440 poses it: forward-Euler (A,B) at vx=15 cast to
        float32, px dropped, Q' = diag(W11, W22, 0, 0, W33), R' = W[4,4].  The control-admissible set the
        reference also estimates there (LMPC2.py:444-449, stability.py:7-90) never enters its OCP (the terminal
        constraint is commented out, LMPC2.py:212-218); here it is computed on first use of `terminal_set_estimate()`
        and enters the problem only through `set_terminal_set` / the constructor's `terminal_set`."""
        A = (np.eye(6) + self.dt * self.continuous_A(float(self.p_lin_point[0]))).astype(np.float32)
        B = np.zeros((6, 1)); B[5, 0] = self.dt
        B = B.astype(np.float32)
        A = A[1:, 1:]; B = B[1:, :]
        self.A_stability, self.B_stability = A, B
        W = self.W
        Q = np.diag([W[1, 1], W[2, 2], 0.0, 0.0, W[3, 3]])
        R = np.array([[W[4, 4]]])
        Ad, Bd = A.astype(np.float64), B.astype(np.float64)
        self.P = scipy.linalg.solve_discrete_are(Ad, Bd, Q, R)
        self.K = np.linalg.solve(Bd.T @ self.P @ Bd + R, Bd.T @ self.P @ Ad)
        self.C = None

    def terminal_set_estimate(self):
        """The rows `self.C` of LMPC2.py:444-449: estimate_control_amissible_invariant_set(K, A, B, ubx, ubu) with
        ubx = (100, 100, 100, 100, max_steering), ubu = (max_steering_rate); computed once, on the host."""
        if self.C is None:
            from .stability import estimate_control_amissible_invariant_set
            ubx = np.array([100, 100, 100, 100, self.max_steering], dtype=np.float64)
            ubu = np.array([self.max_steering_rate], dtype=np.float64)
            self.C = estimate_control_amissible_invariant_set(self.K, self.A_stability.astype(np.float64),
                                                              self.B_stability.astype(np.float64), ubx, ubu)
        return self.C

    def set_terminal_set(self, C):
        """Adds the terminal constraint C (x_N[1:] - x_ref) <= 1 the reference leaves commented out (LMPC2.py:212-218) to
        every following solve; C (m, 5) over (py, psi, vy, r, delta), None removes it.  A controller created with
        solver_layout "auto" moves to the lane-per-scenario solver with its first set; "wave" refuses."""
        C = np.zeros((0, 5)) if C is None else np.ascontiguousarray(C, dtype=np.float64)
        if C.ndim != 2 or C.shape[1] != 5:
            raise ValueError("terminal set: rows of 5 coefficients over (py, psi, vy, r, delta)")
        rc = self.solver._lib.mpc_set_terminal_set(self.solver.h, int(C.shape[0]), _lib.hptr(C))
        _lib.check(rc, self.solver.h, "mpc_set_terminal_set")
        if C.shape[0]:
            self._cached_vx = self._cache_candidate = None       # the lane layout has no fixed-model cache
            self._auto_cache = False

    # ---- per step -----------------------------------------------------------------------------
    def waypoints_to_references(self, waypoints: np.ndarray) -> np.ndarray:
        references = np.zeros([self.N + 1, self.n_states + self.n_inputs])
        references[:, 0] = waypoints[:, 0]; references[:, 1] = waypoints[:, 1]; references[:, 2] = waypoints[:, 3]
        return references

    def _buf(self, name, shape, dtype=torch.float64):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    def solve_batched(self, x0, waypoints, p, d_hat=None, lin_x=None, lin_u=None, want_trajectory=True):
        """Batched optimize(): device tensors in, device tensors out, no host synchronisation.
        x0 (B,7), waypoints (B,N+1,4), p (B,N+1) [, d_hat (B,)]."""
        dev = self.device
        x0 = _dev.f64(x0, dev); waypoints = _dev.f64(waypoints, dev); p = _dev.f64(p, dev)
        B = x0.shape[0]
        if B > self.batch:
            raise ValueError(f"batch {B} exceeds the workspace of this controller ({self.batch})")
        assert x0.shape == (B, 7) and waypoints.shape == (B, self.N + 1, 4) and p.shape == (B, self.N + 1)
        U = torch.empty((B, self.N), dtype=torch.float64, device=dev)
        traj = torch.empty((B, self.N + 1, 7), dtype=torch.float64, device=dev) if want_trajectory else None
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        dh = _dev.f64(d_hat, dev) if d_hat is not None else None
        rc = self.solver._lib.mpc_solve_only(self.solver.h, _dev.stream_ptr(dev), B, _dev.ptr(x0), _dev.ptr(waypoints),
                                             _dev.ptr(p), _dev.ptr(dh), _dev.ptr(lin_x), _dev.ptr(lin_u), _dev.ptr(U),
                                             _dev.ptr(traj), _dev.ptr(status), _dev.ptr(iters))
        _lib.check(rc, self.solver.h, "mpc_solve_only")
        return U, traj, status, iters

    def _optimize_one(self, x0, waypoints, p, d_hat=None, lin_x=None, lin_u=None):
        t0 = time.perf_counter()
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        wp = np.asarray(waypoints, dtype=np.float64)
        pv = np.asarray(p, dtype=np.float64).reshape(-1)
        if x0.shape[0] < 7 or wp.shape != (self.N + 1, 4) or pv.shape[0] < self.N:
            raise ValueError("optimize: x0 needs 7 entries, waypoints (N+1,4), p at least N speeds")
        pfull = np.empty(self.N + 1); pfull[: min(len(pv), self.N + 1)] = pv[: self.N + 1]
        if len(pv) < self.N + 1:
            pfull[self.N] = pv[self.N - 1]          # the stage-N parameter is never used (LMPC2.py:301-303)
        dh = None if d_hat is None else np.array([float(d_hat)])
        if self.kind in ("L", "OFL") and self._auto_cache:
            # The planners hand the controller one constant speed (path_planning.py:238-240): the second call in a row
            # with the same constant profile condenses the horizon once for that speed (mpc_condense_fixed, non-exclusive:
            # any other profile still condenses its own problem), and the following calls are one cached affine map.
            v = pfull[0]
            if (pfull[: self.N] == v).all():
                if v == self._cache_candidate and v != self._cached_vx:
                    self.condense_fixed(v, exclusive=False)
                    self._cached_vx = v
                self._cache_candidate = v
        # host arrays in, host arrays out: one launch and one synchronisation (mpc_optimize_host); lin_x / lin_u are
        # C-contiguous float64 host arrays, updated in place
        x0c = np.ascontiguousarray(x0[:7]); wp = np.ascontiguousarray(wp)
        U = np.empty(self.N); traj = np.empty((self.N + 1, 7)); st_it = np.zeros(2, dtype=np.int32)
        hp = lambda a: None if a is None else a.ctypes.data
        rc = self.solver._lib.mpc_optimize_host(self.solver.h, _dev.stream_ptr(self.device), 1, hp(x0c), hp(wp), hp(pfull), hp(dh),
                                                hp(lin_x), hp(lin_u), hp(U), hp(traj), st_it.ctypes.data, st_it.ctypes.data + 4)
        _lib.check(rc, self.solver.h, "mpc_optimize_host")
        st = int(st_it[0])
        self.last_iterations = int(st_it[1])
        self.metrics["runtime"].append(time.perf_counter() - t0)
        if st == 2:
            log.warning("interior-point iteration cap reached (status 2)")
        elif st == 1 and self.kind == "NL":
            log.warning("SQP iteration cap reached before the step tolerance (status 1)")
        elif st != 0:
            raise RuntimeError(f"MPC solve failed with status {st}")
        return 0, traj, U.reshape(self.N, 1)

    def set_model_params(self, mp_per_scenario):
        """Per-scenario model parameters (batch, 6) = [m, I_z, lf, lr, Cf, Cr] for tyre / mass sweeps; None restores the
        parameters of the yaml file for every scenario.  The fixed-model cache is bypassed while they are set."""
        if mp_per_scenario is None:
            rc = self.solver._lib.mpc_set_model_params(self.solver.h, None)
        else:
            mp = _dev.f64(np.ascontiguousarray(mp_per_scenario, dtype=np.float64), self.device)
            if tuple(mp.shape) != (self.batch, 6):
                raise ValueError(f"model parameters must be ({self.batch}, 6)")
            torch.cuda.synchronize(self.device)
            rc = self.solver._lib.mpc_set_model_params(self.solver.h, _dev.ptr(mp))
        _lib.check(rc, self.solver.h, "mpc_set_model_params")

    def set_cost_weights(self, w_per_scenario, We_per_scenario):
        """Per-scenario cost for tuning sweeps: w (batch, 7) = diag(W), We (batch, 6, 6) terminal weight on the 6 states;
        (None, None) restores the weights of the yaml file.  Scenarios are then condensed individually."""
        if w_per_scenario is None:
            rc = self.solver._lib.mpc_set_cost_weights(self.solver.h, None)
        else:
            w = np.asarray(w_per_scenario, dtype=np.float64).reshape(self.batch, 7)
            We = np.asarray(We_per_scenario, dtype=np.float64).reshape(self.batch, 36)
            if np.any(w[:, 6] <= 0):
                raise ValueError("the input weight must be positive in every scenario")
            cost = _dev.f64(np.ascontiguousarray(np.hstack([w, We])), self.device)
            torch.cuda.synchronize(self.device)
            rc = self.solver._lib.mpc_set_cost_weights(self.solver.h, _dev.ptr(cost))
        _lib.check(rc, self.solver.h, "mpc_set_cost_weights")

    def condense_fixed(self, vx, exclusive: bool = False):
        """Cache the condensed problem for a constant speed profile (None switches the cache off).  With
        `exclusive` the caller promises that every scenario runs at vx (status 3 flags a broken promise)."""
        v = float("nan") if vx is None else float(vx)
        self._cached_vx = None if vx is None else float(vx)
        rc = self.solver._lib.mpc_condense_fixed(self.solver.h, _dev.stream_ptr(self.device), v, int(bool(exclusive)))
        _lib.check(rc, self.solver.h, "mpc_condense_fixed")

    def reset_warm_start(self):
        """Forget what earlier solves left in the handle (interior-point warm-start records, launch order): call after
        handing the controller unrelated scenarios."""
        rc = self.solver._lib.mpc_reset_warm_start(self.solver.h, _dev.stream_ptr(self.device))
        _lib.check(rc, self.solver.h, "mpc_reset_warm_start")

    def stats(self) -> dict:
        """Sums and maxima over the scenarios of the last call (include/mpc_amd.h, mpc_get_stats).  The keys from
        "sum_stage_cost" on are the closed-loop statistics of the last closed-loop run (zero after optimize())."""
        out = np.zeros(_lib.STATS_LEN)
        rc = self.solver._lib.mpc_get_stats(self.solver.h, _dev.stream_ptr(self.device), _lib.hptr(out))
        _lib.check(rc, self.solver.h, "mpc_get_stats")
        return {k: float(out[i]) for i, k in _lib.STATS_SLOTS.items()}


class LOcp(_CondensedMpc):
    """Linear MPC, speed-parameterised 6-state model (LMPC2.py:12)."""

    kind = "L"
    param_file = "parameters_L.yaml"

    def _cost_matrices(self):
        W = self._stage_weights()
        self.W = W
        self.stability()
        W_e = np.zeros((7, 7))
        W_e[1:-1, 1:-1] = self.params["controller"]["beta"] * 1 / 2 * self.P       # LMPC2.py:255-260
        return W, W_e

    def optimize(self, x0, waypoints, p):
        return self._optimize_one(x0, waypoints, p)


class OFLOcp(_CondensedMpc):
    """Offset-free linear MPC: the L model plus a side-force disturbance parameter (OFLMPC2.py:14)."""

    kind = "OFL"
    param_file = "parameters_L.yaml"
    n_parameters = 1
    n_disturbances = 1

    def _cost_matrices(self):
        W = self._stage_weights()
        self.W = W
        self.stability()
        W_e = np.zeros((7, 7))
        W_e[1:-1, 1:-1] = 1 / 2 * self.P                                            # OFLMPC2.py:274-279
        self.B_d = np.zeros((6, 1)); self.B_d[3, 0] = 1 / self.m
        self.C_d = np.eye(self.n_outputs, self.n_disturbances)
        return W, W_e

    def optimize(self, x0, waypoints, p, d_hat):
        return self._optimize_one(x0, waypoints, p, d_hat=d_hat)


class LPVOcp(_CondensedMpc):
    """LPV MPC: the nonlinear kinematic model re-linearised per stage and per step (LPVMPC.py:9)."""

    kind = "LPV"
    param_file = "parameters_LPV.yaml"
    n_params = 1

    def __init__(self, N: int, Tf: float, discrete: bool = True, debug: bool = False, **kw):
        super().__init__(N, Tf, discrete, debug, **kw)
        self.prev_x = np.zeros((6, self.N + 1))
        self.prev_u = np.zeros((1, self.N + 1))
        self.metrics = {"runtime": [], "cost": [], "status": []}
        self._lin_mode = "reference"

    def optimize(self, x0, waypoints, p, lin_mode: str = "reference"):
        if lin_mode not in ("reference", "prev_iter"):
            raise ValueError(f"unknown lin_mode {lin_mode!r}")
        if lin_mode != self._lin_mode:
            self._lin_mode = lin_mode
            _lib.check(self.solver._lib.mpc_set_lpv_lin_mode(self.solver.h, 1 if lin_mode == "prev_iter" else 0), self.solver.h,
                       "mpc_set_lpv_lin_mode")
        lin_x = np.ascontiguousarray(self.prev_x, dtype=np.float64).copy()
        lin_u = np.ascontiguousarray(np.asarray(self.prev_u, dtype=np.float64)[0, : self.N]).copy()
        out = self._optimize_one(x0, waypoints, p, lin_x=lin_x, lin_u=lin_u)
        self.prev_x = lin_x
        self.prev_u = lin_u[None, :]
        return out


class NLOcp(_CondensedMpc):
    """Nonlinear MPC on the 7-state model [x, y, cos, sin, vy, r, delta] (NLMPC.py:9): Gauss-Newton SQP with the
    explicit RK4 map of the reference's ERK integrator, solved to the step tolerance `sqp_tol` (at most
    `sqp_max_iter` iterations per call).  The input iterate is kept between calls like the reference's solver object
    keeps its own (NLMPC.py:232); `reset_iterate()` returns to the cold start."""

    kind = "NL"
    param_file = "parameters_NL.yaml"
    n_states = 7
    n_inputs = 1
    n_outputs = 8

    def __init__(self, N: int, Tf: float, debug: bool = False, *, sqp_tol: float = 1e-9, sqp_max_iter: int = 30, **kw):
        self._sqp = (float(sqp_tol), int(sqp_max_iter))
        super().__init__(N, Tf, True, debug, **kw)
        self._iterate = None if self.device is None else torch.zeros((self.batch, self.N), dtype=torch.float64,
                                                                     device=self.device)
        self._iterate_host = np.zeros(self.N)        # the iterate optimize() carries (solve_batched keeps its own on the device)

    def _cost_matrices(self):
        Q = np.array([0, 1e5, 1e-10, 1e0, 0, 1e0, 1e0])                              # NLMPC.py:174-179
        R = np.array([1e0]) * 10
        return np.diag(np.append(Q, R)), np.diag(np.append(Q * 100, 0))

    def _make_config(self) -> _lib.MpcConfig:
        cfg = _lib.MpcConfig()
        cfg.kind = _lib.KIND[self.kind]
        cfg.N = self.N
        cfg.B = self.batch
        cfg.device = self.device.index if self.device is not None else 0
        cfg.dt = self.dt
        cfg.mp[:] = list(self.model_params())
        cfg.max_steering = self.max_steering
        cfg.max_rate = self.max_steering_rate
        cfg.w_nl[:] = list(np.diag(self.W))
        cfg.we_nl[:] = list(np.diag(self.W_e)[:7])
        cfg.eps_nl = 0.1                                                             # NLMPC.py:116-126
        cfg.tol_mu, cfg.tol_res, cfg.max_iter = self._tol
        cfg.sqp_tol, cfg.sqp_max_iter = self._sqp
        cfg.ipm_cold_start = self._ipm_cold
        cfg.step_ramp_length = self._step_ramp
        cfg.qp_method = self._qp_method              # "interior_point": no active-set attempt at the QPs of the SQP iterations
        return cfg

    def stability(self):
        raise NotImplementedError("the reference's NLOcp has no terminal-set computation")

    def condense_fixed(self, vx, exclusive: bool = False):
        raise NotImplementedError("the NL model is re-linearised at every SQP iteration: nothing to cache")

    def waypoints_to_references(self, waypoints: np.ndarray) -> np.ndarray:
        references = np.zeros([self.N + 1, self.n_outputs])
        references[:, :4] = waypoints[:, :]
        return references

    def reset_iterate(self):
        self._iterate.zero_()
        self._iterate_host[:] = 0.0

    def solve_batched(self, x0, waypoints, p, d_hat=None, lin_x=None, lin_u=None, want_trajectory=True):
        """Batched optimize(); `lin_u` (B,N) is the SQP iterate, read and updated in place (default: the iterate this
        object keeps).  Returned trajectories hold the 7 NL states."""
        if lin_u is None:
            lin_u = self._iterate[: np.shape(x0)[0]]
        return super().solve_batched(x0, waypoints, p, None, None, lin_u, want_trajectory)

    def optimize(self, x0, waypoints, p):
        return self._optimize_one(x0, waypoints, p, lin_u=self._iterate_host)
