"""Skidpad reference planner with the class surface of the reference's utils/path_planning.py; the waypoints are
computed on the GPU (one wavefront per request, one lane per waypoint)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _dev, _lib

# skidpad geometry (utils/path_planning.py:7-11)
r = 9.125
center = 16.75
half_width = 1.5
lap_length = 2 * r * np.pi
end_straight_length = 15


class SkidpadPlanner:
    def __init__(self, target_vel: float, max_accel: float = 10, Nt: float = 10, dt: float = 0.2,
                 slowdown_speed_factor=0.5, device=None):
        self.target_vel = target_vel
        self.max_accel = max_accel
        self.dt = dt
        self.Nt = Nt
        accel_time = self.target_vel / self.max_accel
        self.accel_zone_1_start = 0
        self.accel_zone_1_end = self.max_accel / 2 * accel_time * accel_time
        self.slowdown_speed_factor = slowdown_speed_factor
        decel_time = accel_time * (1 - slowdown_speed_factor)
        self.braking_zone_1_start = center + 1 + 4 * lap_length
        self.braking_zone_1_end = (self.braking_zone_1_start + decel_time * self.target_vel
                                   + self.max_accel / 2 * decel_time * decel_time)
        decel_time_2 = accel_time * slowdown_speed_factor
        self.braking_zone_2_start = center + end_straight_length + 4 * lap_length
        self.braking_zone_2_end = self.braking_zone_2_start + decel_time_2 * self.max_accel
        self.prev_progress = 0
        self.device = _dev.require_gpu(device)
        self._lib = _lib.load()

    def set_starting_position(self, progress):
        accel_time = self.target_vel / self.max_accel
        self.accel_zone_1_start = progress
        self.accel_zone_1_end = self.max_accel / 2 * accel_time * accel_time + progress

    @staticmethod
    def pos2progress(x, y, lap):
        """Scalar helper (utils/path_planning.py:196-209); the kernels carry their own copy."""
        if lap == 0:
            return x
        if 0 < lap < 3:
            return (-np.arctan2(x - center, -r - y) + np.pi) * r + (lap - 1) * lap_length + center
        if 2 < lap < 5:
            return (np.arctan2(center - x, y - r) + np.pi) * r + (lap - 1) * lap_length + center
        return x + 4 * lap_length

    def request_progress(self, x, y, lap):
        current_progress = self.pos2progress(x, y, lap)
        lap_correct = True
        if self.prev_progress > (current_progress + 2 * r):
            lap += 1
            current_progress = self.pos2progress(x, y, lap)
            lap_correct = False
        return current_progress, lap_correct

    def zones(self) -> np.ndarray:
        """The seven numbers the speed profile depends on, in the order the kernel reads them."""
        return np.array([self.accel_zone_1_start, self.accel_zone_1_end, self.braking_zone_1_start,
                         self.braking_zone_1_end, self.braking_zone_2_start, self.braking_zone_2_end,
                         self.slowdown_speed_factor], dtype=np.float64)

    def progress2speed(self, progress):
        """Scalar helper (utils/path_planning.py:93-126); the kernel carries its own copy."""
        v, f = self.target_vel, self.slowdown_speed_factor
        if progress > self.braking_zone_2_end:
            return 0
        if self.braking_zone_2_end > progress > self.braking_zone_2_start:
            return v * f - (progress - self.braking_zone_2_start) * v * f
        if self.braking_zone_2_start > progress > self.braking_zone_1_end:
            return v * f
        if self.braking_zone_1_end > progress > self.braking_zone_1_start:
            return v - (progress - self.braking_zone_1_start) * v * (1 - f)
        if self.braking_zone_1_start > progress > self.accel_zone_1_end:
            return v
        if self.accel_zone_1_end > progress > self.accel_zone_1_start:
            return (progress - self.accel_zone_1_start) * (v - 0.1) + 0.1
        return 0.1

    def request_waypoints_batched(self, pose, lap, prev_progress, target_vel, Nt=None, dt=None, zones=None,
                                  return_speeds=False):
        """pose (B,3) = x, y, heading; lap (B,) int32; prev_progress (B,) is updated in place (device tensors).
        zones (B,7) (see zones()) switches the speed profile on for laps outside 0..7; without it every lap is planned
        at the constant target speed."""
        dev = self.device
        Nt = self.Nt if Nt is None else Nt
        dt = self.dt if dt is None else dt
        pose = _dev.f64(pose, dev); B = pose.shape[0]
        wp = torch.empty((B, Nt + 1, 4), dtype=torch.float64, device=dev)
        prog = torch.empty((B,), dtype=torch.float64, device=dev)
        if zones is None and not return_speeds:
            rc = self._lib.mpc_plan_skidpad(_dev.stream_ptr(dev), B, int(Nt), float(dt), _dev.ptr(pose), _dev.ptr(lap),
                                            _dev.ptr(prev_progress), _dev.ptr(target_vel), _dev.ptr(wp), _dev.ptr(prog))
            _lib.check(rc, None, "mpc_plan_skidpad")
            return wp, prog
        if zones is None:
            zones = np.tile(self.zones(), (B, 1))
        zones = _dev.f64(zones, dev)
        if tuple(zones.shape) != (B, 7):
            raise ValueError("zones must be (B, 7)")
        speeds = torch.empty((B, Nt + 1), dtype=torch.float64, device=dev)
        rc = self._lib.mpc_plan_skidpad_profile(_dev.stream_ptr(dev), B, int(Nt), float(dt), _dev.ptr(pose), _dev.ptr(lap),
                                                _dev.ptr(prev_progress), _dev.ptr(target_vel), _dev.ptr(zones), _dev.ptr(wp),
                                                _dev.ptr(prog), _dev.ptr(speeds))
        _lib.check(rc, None, "mpc_plan_skidpad_profile")
        return (wp, prog, speeds) if return_speeds else (wp, prog)

    def request_waypoints(self, x, y, heading, lap, Nt=-1, dt=-1):
        if Nt == -1:
            Nt = self.Nt
        if dt == -1:
            dt = self.dt
        dev = self.device
        pp = _dev.f64(np.array([float(self.prev_progress)]), dev).clone()
        wp, prog, speeds = self.request_waypoints_batched(
            np.array([[x, y, heading]], dtype=np.float64), _dev.i32(np.array([lap]), dev), pp,
            _dev.f64(np.array([float(self.target_vel)]), dev), Nt, dt, zones=self.zones()[None], return_speeds=True)
        self.prev_progress = float(pp.item())
        heading_derotation = np.array([[np.cos(heading), -np.sin(heading)], [np.sin(heading), np.cos(heading)]])
        return wp.cpu().numpy()[0], speeds.cpu().numpy()[0], float(prog.item()), heading_derotation

    def request_desired_speed(self, progress) -> float:
        if progress < self.braking_zone_1_start:
            return self.target_vel
        elif progress < self.braking_zone_2_start:
            return self.target_vel * self.slowdown_speed_factor
        return 0
