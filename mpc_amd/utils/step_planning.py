"""Lateral-step reference planner with the class surface of the reference's utils/step_planning.py (GPU)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _dev, _lib


class StepPlanner:
    def __init__(self, target_vel: float, Nt: float = 10, dt: float = 0.2, ramp_length: float = 0.0, fields=None,
                 device=None):
        if not ramp_length >= 0.0:
            raise ValueError("ramp_length must be >= 0")
        self.target_vel = target_vel
        self.dt = dt
        self.Nt = Nt
        self.prev_progress = 0
        self.step_amplitude = 1.0
        self.ramp_length = ramp_length
        self.device = _dev.require_gpu(device)
        self._lib = _lib.load()

    @staticmethod
    def pos2progress(x, y):
        return x

    def request_waypoints_batched(self, pose, target_vel, Nt=None, dt=None, want_absolute=True):
        dev = self.device
        Nt = self.Nt if Nt is None else Nt
        dt = self.dt if dt is None else dt
        pose = _dev.f64(pose, dev); B = pose.shape[0]
        wp = torch.empty((B, Nt + 1, 4), dtype=torch.float64, device=dev)
        ab = torch.empty((B, Nt + 1, 4), dtype=torch.float64, device=dev) if want_absolute else None
        prog = torch.empty((B,), dtype=torch.float64, device=dev)
        rc = self._lib.mpc_plan_step(_dev.stream_ptr(dev), B, int(Nt), float(dt), float(self.ramp_length),
                                     _dev.ptr(pose), _dev.ptr(target_vel), _dev.ptr(wp), _dev.ptr(prog), _dev.ptr(ab))
        _lib.check(rc, None, "mpc_plan_step")
        return wp, prog, ab

    def request_waypoints(self, x, y, heading, Nt=-1, dt=-1):
        if Nt == -1:
            Nt = self.Nt
        if dt == -1:
            dt = self.dt
        wp, prog, ab = self.request_waypoints_batched(np.array([[x, y, heading]], dtype=np.float64),
                                                      _dev.f64(np.array([float(self.target_vel)]), self.device), Nt, dt)
        self.prev_progress = float(prog.item())
        speeds = np.full(Nt + 1, float(self.target_vel))
        heading_derotation = np.array([[np.cos(heading), -np.sin(heading)], [np.sin(heading), np.cos(heading)]])
        return wp.cpu().numpy()[0], speeds, float(prog.item()), heading_derotation, ab.cpu().numpy()[0]
