"""Plant model and measurement model with the class surface of the reference's continuous_dynamics.py; the
arithmetic (single-track model, RK4, EKF Jacobian) runs on the GPU through libmpc_amd."""
from __future__ import annotations

import numpy as np
import torch

from . import _dev, _lib
from .controllers import tyre_stiffness

indices = {"pos_x": 0, "pos_y": 1, "heading_cos": 2, "heading_sin": 3, "vx": 4, "vy": 5, "r": 6, "steering": 7,
           "steering_dist": 8, "d_f": 9}

_MEASURED = [0, 1, 2, 3, 4, 6, 7]          # everything but vy (and the disturbances)


class Dynamics:
    """continuous_dynamics.py:17.  x = [px, py, cos, sin, vx, vy, r, delta(, delta_dist, d_f)]."""

    def __init__(self, dt=0.002, disturbance: bool = False, device=None, _host_only: bool = False):
        self.m, self.I_z, self.wbase, self.x_cg = 180, 294, 1.53, 0.57      # hard-coded in the reference (:21-24)
        self.lf = self.x_cg * self.wbase
        self.lr = (1 - self.x_cg) * self.wbase
        self.Cf, self.Cr = self.get_tyre_stiffness()
        self.dt = dt
        self.nx = 10 if disturbance else 8
        self.disturbed = bool(disturbance)
        self.measurement_matrix = np.zeros((7, self.nx))
        self.measurement_matrix[np.arange(7), _MEASURED] = 1
        self.measurement_noises = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])
        self.rng = np.random.default_rng(1)
        # _host_only: the measurement model alone (selection matrix, noise draws), for use without a GPU; the plant
        # model, RK4 and the EKF Jacobian need the device
        self.device = None if _host_only else _dev.require_gpu(device)
        self._lib = None if _host_only else _lib.load()
        self._mp = np.array([self.m, self.I_z, self.lf, self.lr, self.Cf, self.Cr], dtype=np.float64)

    def get_tyre_stiffness(self) -> tuple[float, float]:
        return tyre_stiffness(self.m, self.x_cg)

    # -- batched entry points (device tensors, no host sync) ---------------------------------
    def rk4_batched(self, x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
        x = _dev.f64(x, self.device).clone(); u = _dev.f64(u, self.device)
        rc = self._lib.mpc_plant_rk4(_dev.stream_ptr(self.device), x.shape[0], x.shape[1], float(self.dt),
                                     _lib.hptr(self._mp), _dev.ptr(x), _dev.ptr(u))
        _lib.check(rc, None, "mpc_plant_rk4")
        return x

    def model_batched(self, x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
        x = _dev.f64(x, self.device); u = _dev.f64(u, self.device)
        out = torch.empty_like(x)
        rc = self._lib.mpc_plant_f(_dev.stream_ptr(self.device), x.shape[0], x.shape[1], _lib.hptr(self._mp),
                                   _dev.ptr(x), _dev.ptr(u), _dev.ptr(out))
        _lib.check(rc, None, "mpc_plant_f")
        return out

    # -- the reference's single-state surface ---------------------------------------------------
    def _one(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if x.shape[0] != self.nx:
            raise ValueError(f"state has {x.shape[0]} entries, this Dynamics expects {self.nx}")
        return x

    def single_track_model(self, x, u) -> np.ndarray:
        x = self._one(x)
        return self.model_batched(x[None], np.array([float(np.asarray(u).reshape(-1)[0])])).cpu().numpy()[0]

    def rk4_integraton(self, xk, u) -> np.ndarray:
        xk = self._one(xk)
        return self.rk4_batched(xk[None], np.array([float(np.asarray(u).reshape(-1)[0])])).cpu().numpy()[0]

    def steering_dist_jacobian(self):
        return np.array([0, 0, 0, 0, 0, -self.Cf / self.m, -(self.Cf * self.lf) / self.I_z, 0, 0, 0])

    def side_force_dist_jacobian(self):
        return np.array([0, 0, 0, 0, 0, 1 / self.m, 0, 0, 0, 0])

    def jacobian_forward_euler(self, x):
        """(A, B, F = I + dt*A) at x; 10-state (disturbed) model only.  A is recovered from the kernel's F."""
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if x.shape[0] != 10:
            raise NotImplementedError("the EKF Jacobian is implemented for the 10-state model the estimator uses")
        xd = _dev.f64(x[None], self.device)
        F = torch.empty((1, 10, 10), dtype=torch.float64, device=self.device)
        rc = self._lib.mpc_ekf_jacobian(_dev.stream_ptr(self.device), 1, float(self.dt), _lib.hptr(self._mp),
                                        _dev.ptr(xd), _dev.ptr(F))
        _lib.check(rc, None, "mpc_ekf_jacobian")
        F = F.cpu().numpy()[0]
        B = np.zeros(10); B[7] = 1
        return (F - np.eye(10)) / self.dt, B, F

    def measurement_jacobian(self):
        return self.measurement_matrix

    def measure_state_noise(self, x):
        """Selection of the measured channels plus the reference's noise term: ONE scalar drawn from
        N(len(x), 1) scaled per channel (continuous_dynamics.py:216-220)."""
        return np.asarray(x)[_MEASURED] + self.measurement_noises * self.rng.normal(len(x))

    def measurement_noise(self, n_steps: int, batch: int, literal: bool = False) -> np.ndarray:
        """Additive measurement noise (n_steps, batch, 7) for the batched output-feedback loop.  literal=False: an
        independent N(0, 1) draw per channel, scaled by the channel's sigma (what a noisy measurement means).
        literal=True: what measure_state_noise adds, one N(len(x), 1) scalar per measurement scaled per channel."""
        if literal:
            return self.rng.normal(self.nx, 1.0, (n_steps, batch, 1)) * self.measurement_noises
        return self.rng.standard_normal((n_steps, batch, 7)) * self.measurement_noises

    def measure_state_noiseless(self, x):
        return np.asarray(x, dtype=np.float64)[_MEASURED]
