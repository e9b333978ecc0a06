"""Batched closed-loop simulators: B independent scenarios advanced by one fused kernel launch per step
(plan -> linearise/condense -> QP -> RK4 plant -> lap count), state resident in HBM.

Per scenario the semantics are those of SkidpadSimulator.step (skidpad_simulator.py:109-135) and of the
loop body of StepSimulator.simulate (simulator.py:178-202)."""
from __future__ import annotations

import numpy as np
import torch

from . import _dev, _lib
from .continuous_dynamics import Dynamics
from .controllers import LOcp, LPVOcp, NLOcp, OFLOcp, read_parameters
from .EKF import P0_DIAG, Q_DIAG


class _ClosedLoopBatch:
    planner = _lib.PLANNER_SKIDPAD

    def __init__(self, N, Tf, starting_states, starting_laps=None, model="L", device=None, target_vel=None,
                 fixed_model_cache=True, **ctl_kw):
        states = np.ascontiguousarray(starting_states, dtype=np.float64)
        if states.ndim != 2 or states.shape[1] != 8:
            raise ValueError("starting_states must be (B, 8): [x, y, cos, sin, vx, vy, r, delta]")
        self.B = B = states.shape[0]
        self.N, self.Tf, self.dt = int(N), float(Tf), float(Tf) / int(N)
        if model == "L":
            self.ocp = LOcp(N, Tf, batch=B, device=device, **ctl_kw)
        elif model == "LPV":
            self.ocp = LPVOcp(N, Tf, batch=B, device=device, **ctl_kw)
        elif model == "NL":
            self.ocp = NLOcp(N, Tf, batch=B, device=device, **ctl_kw)
        else:
            raise ValueError("model must be 'L', 'LPV' or 'NL' for the fused closed loop")
        self.MPC_controller = self.ocp
        dev = self.device = self.ocp.device
        self.state = _dev.f64(states, dev)
        self.lap = _dev.i32(np.zeros(B) if starting_laps is None else starting_laps, dev)
        self.prev_progress = torch.zeros(B, dtype=torch.float64, device=dev)
        tv = states[:, 4] if target_vel is None else np.broadcast_to(np.asarray(target_vel, dtype=np.float64), (B,))
        self.target_vel = _dev.f64(tv, dev)               # SkidpadPlanner(target_vel=starting_velocity[0])
        if model == "L" and fixed_model_cache and np.all(tv == tv[0]):
            self.ocp.condense_fixed(float(tv[0]), exclusive=True)   # fixed A,B: condense once, not per scenario and step
        self.u0 = torch.zeros(B, dtype=torch.float64, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        if model == "LPV":
            self.lin_x = torch.zeros((B, 6, self.N + 1), dtype=torch.float64, device=dev)
            self.lin_u = torch.zeros((B, self.N), dtype=torch.float64, device=dev)
        elif model == "NL":
            self.lin_x, self.lin_u = None, self.ocp._iterate          # the SQP iterate travels from step to step
        else:
            self.lin_x = self.lin_u = None
        self.planned_references = None
        self.planned_trajectory = None
        self.planned_inputs = None

    def step(self, record_plan: bool = False) -> torch.Tensor:
        """One closed-loop step for every scenario; returns the applied inputs (B,) (device tensor, no sync)."""
        dev = self.device
        if record_plan and self.planned_references is None:
            self.planned_references = torch.zeros((self.B, self.N + 1, 4), dtype=torch.float64, device=dev)
            self.planned_trajectory = torch.zeros((self.B, self.N + 1, 7), dtype=torch.float64, device=dev)
            self.planned_inputs = torch.zeros((self.B, self.N), dtype=torch.float64, device=dev)
        s = self.ocp.solver
        rc = s._lib.mpc_closed_loop_step(
            s.h, _dev.stream_ptr(dev), self.B, self.planner, _dev.ptr(self.state), _dev.ptr(self.lap),
            _dev.ptr(self.prev_progress), _dev.ptr(self.target_vel), _dev.ptr(self.lin_x), _dev.ptr(self.lin_u),
            _dev.ptr(self.u0), _dev.ptr(self.planned_references if record_plan else None),
            _dev.ptr(self.planned_trajectory if record_plan else None),
            _dev.ptr(self.planned_inputs if record_plan else None), _dev.ptr(self.status), _dev.ptr(self.iters))
        _lib.check(rc, s.h, "mpc_closed_loop_step")
        return self.u0

    def run(self, n_steps: int, record: bool = False):
        """n closed-loop steps in ONE kernel launch (each wavefront carries its scenario through all steps).
        Returns (states (n,B,8), inputs (n,B)) device tensors when record is set, else None.  self.iters then
        holds the interior-point iterations summed over the steps."""
        dev = self.device
        xs = torch.empty((n_steps, self.B, 8), dtype=torch.float64, device=dev) if record else None
        us = torch.empty((n_steps, self.B), dtype=torch.float64, device=dev) if record else None
        s = self.ocp.solver
        rc = s._lib.mpc_closed_loop_run(
            s.h, _dev.stream_ptr(dev), self.B, self.planner, int(n_steps), _dev.ptr(self.state), _dev.ptr(self.lap),
            _dev.ptr(self.prev_progress), _dev.ptr(self.target_vel), _dev.ptr(self.lin_x), _dev.ptr(self.lin_u),
            _dev.ptr(self.u0), None, None, None, _dev.ptr(us), _dev.ptr(xs), _dev.ptr(self.status), _dev.ptr(self.iters))
        _lib.check(rc, s.h, "mpc_closed_loop_run")
        return (xs, us) if record else None

    def simulate(self, n_steps: int):
        """n closed-loop steps; returns (states (n,B,8), inputs (n,B)) as device tensors."""
        return self.run(n_steps, record=True)


class SkidpadBatch(_ClosedLoopBatch):
    planner = _lib.PLANNER_SKIDPAD


class StepBatch(_ClosedLoopBatch):
    planner = _lib.PLANNER_STEP


class StepOFLBatch:
    """Batched output-feedback loop (StepSimulator(model="OFL").simulate_of, simulator.py:206-281): B scenarios,
    each with its own true plant state (10 states incl. disturbances), EKF estimate and covariance, all resident
    in HBM; one launch advances every scenario by n steps."""

    def __init__(self, N, Tf, true_states, estimates=None, device=None, target_vel=None, fixed_model_cache=True, **ctl_kw):
        x = np.ascontiguousarray(true_states, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 10:
            raise ValueError("true_states must be (B, 10)")
        self.B = B = x.shape[0]
        self.N, self.Tf, self.dt = int(N), float(Tf), float(Tf) / int(N)
        self.ocp = self.MPC_controller = OFLOcp(N, Tf, batch=B, device=device, **ctl_kw)
        dev = self.device = self.ocp.device
        self.state = _dev.f64(x, dev)
        if estimates is None:                       # simulator.py:214-224
            estimates = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
        self.estimate = _dev.f64(np.ascontiguousarray(estimates, dtype=np.float64), dev).clone()
        self.P = _dev.f64(np.tile(np.diag(P0_DIAG), (B, 1, 1)), dev).clone()
        tv = x[:, 4] if target_vel is None else np.broadcast_to(np.asarray(target_vel, dtype=np.float64), (B,))
        self.target_vel = _dev.f64(np.ascontiguousarray(tv), dev)
        if fixed_model_cache and np.all(tv == tv[0]):
            self.ocp.condense_fixed(float(tv[0]), exclusive=True)
        self._qd = np.ascontiguousarray(Q_DIAG, dtype=np.float64)
        self._rd = R_DIAG.copy()
        self.u0 = torch.zeros(B, dtype=torch.float64, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)

    def run(self, n_steps: int, record: bool = False, meas_noise=None):
        """meas_noise: optional (n_steps, B, 7) additive measurement noise (e.g. sigma * N(0,1) per channel)."""
        dev = self.device
        xs = torch.empty((n_steps, self.B, 10), dtype=torch.float64, device=dev) if record else None
        es = torch.empty((n_steps, self.B, 10), dtype=torch.float64, device=dev) if record else None
        us = torch.empty((n_steps, self.B), dtype=torch.float64, device=dev) if record else None
        noise = _dev.f64(meas_noise, dev) if meas_noise is not None else None
        if noise is not None and tuple(noise.shape) != (n_steps, self.B, 7):
            raise ValueError("meas_noise must be (n_steps, B, 7)")
        s = self.ocp.solver
        rc = s._lib.mpc_ofl_closed_loop_run(
            s.h, _dev.stream_ptr(dev), self.B, int(n_steps), _dev.ptr(self.state), _dev.ptr(self.estimate),
            _dev.ptr(self.P), _dev.ptr(self.target_vel), _lib.hptr(self._qd), _lib.hptr(self._rd), _dev.ptr(noise),
            _dev.ptr(self.u0), _dev.ptr(us), _dev.ptr(xs), _dev.ptr(es), _dev.ptr(self.status), _dev.ptr(self.iters))
        _lib.check(rc, s.h, "mpc_ofl_closed_loop_run")
        return (xs, us, es) if record else None


R_DIAG = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])             # R = diag(sigma) (EKF.py:32)


class StepLQRBatch:
    """Batched saturated-LQR baseline (StepSimulator.dlqr_sim, simulator.py:378-428): B scenarios, each with its own
    plant state and, optionally, its own gain (K of shape (B, 5)); one launch advances every scenario by n steps.  The
    limits are those of parameters_NL.yaml, as in the reference."""

    def __init__(self, dt, starting_states, K, device=None, params: dict | None = None):
        x = np.ascontiguousarray(starting_states, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] not in (8, 10):
            raise ValueError("starting_states must be (B, 8) or (B, 10)")
        self.B, self.nx = x.shape
        self.dt = float(dt)
        self.dynamics = Dynamics(self.dt, disturbance=self.nx == 10, device=device)
        dev = self.device = self.dynamics.device
        self.state = _dev.f64(x, dev).clone()
        K = np.asarray(K, dtype=np.float64)
        if K.size == 5:
            self._K, self._K_scen = np.ascontiguousarray(K.reshape(5)), None
        elif K.shape == (self.B, 5):
            self._K, self._K_scen = None, _dev.f64(np.ascontiguousarray(K), dev)
        else:
            raise ValueError("K must hold 5 gains, or (B, 5) for one gain per scenario")
        limits = read_parameters("parameters_NL.yaml", params)["model"]
        self.rate_limit = float(limits["max_steering_rate"])
        self.angle_limit = float(limits["max_steering_angle"])

    def run(self, n_steps: int, record: bool = True):
        """Returns (states (n, B, nx), inputs (n, B)) as device tensors when record is set."""
        dev = self.device
        xs = torch.empty((n_steps, self.B, self.nx), dtype=torch.float64, device=dev) if record else None
        us = torch.empty((n_steps, self.B), dtype=torch.float64, device=dev) if record else None
        lib = self.dynamics._lib
        rc = lib.mpc_dlqr_run(_dev.stream_ptr(dev), self.B, self.nx, int(n_steps), self.dt, _lib.hptr(self.dynamics._mp),
                              _lib.hptr(self._K) if self._K is not None else None, _dev.ptr(self._K_scen),
                              self.rate_limit, self.angle_limit, _dev.ptr(self.state), _dev.ptr(us), _dev.ptr(xs))
        _lib.check(rc, None, "mpc_dlqr_run")
        return (xs, us) if record else None


class LsimBatch:
    """Batched open-loop response with the EKF in the loop (StepSimulator.lsim, simulator.py:334-376): B scenarios, each
    with its own true 10-state plant, estimate and covariance; the input sequence is shared ((n,)) or per scenario
    ((n, B))."""

    def __init__(self, dt, true_states, estimates=None, device=None):
        x = np.ascontiguousarray(true_states, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 10:
            raise ValueError("true_states must be (B, 10)")
        self.B = B = x.shape[0]
        self.dt = float(dt)
        self.dynamics = Dynamics(self.dt, disturbance=True, device=device)
        dev = self.device = self.dynamics.device
        self.state = _dev.f64(x, dev).clone()
        if estimates is None:                       # simulator.py:342-353
            estimates = np.tile(np.array([0.0, 0.0, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
        self.estimate = _dev.f64(np.ascontiguousarray(estimates, dtype=np.float64), dev).clone()
        self.P = _dev.f64(np.tile(np.diag(P0_DIAG), (B, 1, 1)), dev).clone()
        self._qd = np.ascontiguousarray(Q_DIAG, dtype=np.float64)
        self._rd = R_DIAG.copy()

    def run(self, u, n_steps: int | None = None, meas_noise=None):
        """u: a float (held for n_steps), (n,) or (n, B).  Returns (states, estimates), both (n, B, 10) device tensors."""
        dev = self.device
        if np.isscalar(u):
            u = np.full(int(n_steps), float(u))
        u = _dev.f64(np.ascontiguousarray(u, dtype=np.float64), dev) if not torch.is_tensor(u) else u.contiguous()
        n = int(u.shape[0]) if n_steps is None else int(n_steps)
        if u.ndim not in (1, 2) or u.shape[0] < n or (u.ndim == 2 and u.shape[1] != self.B):
            raise ValueError("u must be a float, (n_steps,) or (n_steps, B)")
        noise = _dev.f64(meas_noise, dev) if meas_noise is not None else None
        if noise is not None and tuple(noise.shape) != (n, self.B, 7):
            raise ValueError("meas_noise must be (n_steps, B, 7)")
        xs = torch.empty((n, self.B, 10), dtype=torch.float64, device=dev)
        es = torch.empty((n, self.B, 10), dtype=torch.float64, device=dev)
        rc = self.dynamics._lib.mpc_lsim_run(
            _dev.stream_ptr(dev), self.B, n, self.dt, _lib.hptr(self.dynamics._mp), _lib.hptr(self._qd), _lib.hptr(self._rd),
            _dev.ptr(u), int(u.ndim == 2), _dev.ptr(noise), _dev.ptr(self.state), _dev.ptr(self.estimate), _dev.ptr(self.P),
            _dev.ptr(xs), _dev.ptr(es))
        _lib.check(rc, None, "mpc_lsim_run")
        return xs, es


def control_buffer_interp(traj: torch.Tensor, Tf: float, t, col: int = 6, scale: float = np.pi / 2 / 0.4) -> torch.Tensor:
    """Steering command between two solves (archive/mpc_node.py:421-427): np.interp of the predicted steering angle
    (column `col` of traj (B, N+1, ncol), e.g. the trajectory solve_batched returns) at t (B,) seconds since the solve
    over linspace(0, Tf, N+1), times `scale`."""
    if traj.ndim != 3 or not traj.is_cuda or traj.dtype != torch.float64:
        raise ValueError("traj must be a (B, N+1, ncol) float64 device tensor")
    traj = traj.contiguous()
    B, n1, ncol = traj.shape
    t = _dev.f64(np.broadcast_to(np.asarray(t, dtype=np.float64), (B,)).copy(), traj.device) if not torch.is_tensor(t) else t.contiguous()
    out = torch.empty(B, dtype=torch.float64, device=traj.device)
    rc = _lib.load().mpc_control_buffer_interp(_dev.stream_ptr(traj.device), B, n1 - 1, float(Tf), _dev.ptr(traj), ncol, int(col),
                                              _dev.ptr(t), float(scale), _dev.ptr(out))
    _lib.check(rc, None, "mpc_control_buffer_interp")
    return out
