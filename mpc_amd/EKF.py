"""Extended Kalman filter with the class surface of the reference's EKF.py; predict/update run on the GPU."""
from __future__ import annotations

import numpy as np
import torch

from . import _dev, _lib
from .continuous_dynamics import Dynamics, indices  # noqa: F401

P0_DIAG = [1, 1, 1, 1, 1, 10.0, 1.0, 1.0, 0.0, 1000000]                       # EKF.py:28
Q_DIAG = [0.05, 0.05, 0.01, 0.01, 0.001, 0.1, 0.01, 0.01, 0.0, 0.0]           # EKF.py:29-31


class CarEKF:
    """EKF.py:6.  The keyword is spelled `inital_state` as in the reference."""

    def __init__(self, dt, disturbance: bool, inital_state=None, device=None):
        if not disturbance:
            raise NotImplementedError("the reference's filter only works with the 10-state (disturbed) model: "
                                      "its covariance is always 10x10 (EKF.py:28)")
        self.dynamics = Dynamics(dt, disturbance, device=device)
        self.device = self.dynamics.device
        self.disturbed = True
        self.nx = 10
        self.dt = dt
        x0 = np.zeros(10) if inital_state is None else np.asarray(inital_state, dtype=np.float64)
        self._x = _dev.f64(x0[None], self.device).clone()
        self._P = _dev.f64(np.diag(P0_DIAG)[None], self.device).clone()
        self.Q = np.diag(Q_DIAG)
        self.R = np.diag(self.dynamics.measurement_noises)                    # diag(sigma), not sigma^2 (EKF.py:32)
        self._qd = np.ascontiguousarray(np.diag(self.Q)); self._rd = np.ascontiguousarray(np.diag(self.R))
        self.x_real = np.zeros(10)
        self._lib = _lib.load()

    # state is kept in HBM; the attributes the reference exposes are views copied to the host on access
    @property
    def x_est(self):
        return self._x.cpu().numpy()[0]

    @x_est.setter
    def x_est(self, v):
        self._x = _dev.f64(np.asarray(v, dtype=np.float64)[None], self.device).clone()

    @property
    def P(self):
        return self._P.cpu().numpy()[0]

    @P.setter
    def P(self, v):
        self._P = _dev.f64(np.asarray(v, dtype=np.float64)[None], self.device).clone()

    def time_update(self, u):
        ud = _dev.f64(np.array([float(np.asarray(u).reshape(-1)[0])]), self.device)
        rc = self._lib.mpc_ekf_time_update(_dev.stream_ptr(self.device), 1, float(self.dt), _lib.hptr(self.dynamics._mp),
                                           _lib.hptr(self._qd), _dev.ptr(self._x), _dev.ptr(self._P), _dev.ptr(ud))
        _lib.check(rc, None, "mpc_ekf_time_update")

    def measurement_update(self, x_meas):
        assert len(x_meas) == 7
        y = _dev.f64(np.asarray(x_meas, dtype=np.float64)[None], self.device)
        rc = self._lib.mpc_ekf_measurement_update(_dev.stream_ptr(self.device), 1, _lib.hptr(self._rd), _dev.ptr(self._x),
                                                  _dev.ptr(self._P), _dev.ptr(y), None)
        _lib.check(rc, None, "mpc_ekf_measurement_update")

    def estimate_full_state(self):
        return self.x_est

    def measure_x_est(self):
        return self.dynamics.measurement_matrix @ self.x_est

    def estimated_red_state(self):
        return self.x_est[[0, 1, 2, 3, 5, 6, 7, 8, 9]]                        # EKF.py:57-63

    def kalman_gain(self):
        K = torch.empty((1, 10, 7), dtype=torch.float64, device=self.device)
        rc = self._lib.mpc_ekf_kalman_gain(_dev.stream_ptr(self.device), 1, _lib.hptr(self._rd), _dev.ptr(self._P), _dev.ptr(K))
        _lib.check(rc, None, "mpc_ekf_kalman_gain")
        return K.cpu().numpy()[0]
