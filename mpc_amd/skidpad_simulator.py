"""Closed-loop skidpad simulator with the class surface of the reference's skidpad_simulator.py.
Every numerical piece (planner, controller, plant) runs on the GPU through libmpc_amd; matplotlib output of the
reference's step() is not reproduced.  For many scenarios at once use mpc_amd.batched.SkidpadBatch."""
from __future__ import annotations

import logging

import numpy as np

from .continuous_dynamics import Dynamics
from .controllers import LOcp, LPVOcp
from .utils import path_planning
from .utils.path_planning import SkidpadPlanner

log = logging.getLogger("mpc_amd")


def _make_controller(model, N, Tf, device=None):
    if model == "L":
        return LOcp(N, Tf, device=device)
    if model == "LPV":
        return LPVOcp(N, Tf, device=device)
    if model == "NL":
        from .NLMPC import NLOcp
        return NLOcp(N, Tf, device=device)
    return None


class SkidpadSimulator:
    def __init__(self, N, Tf, acados_print_level=0, starting_state=None, starting_lap=None, figures=False,
                 model="LPV", device=None):
        self.N = N
        self.Tf = Tf
        self.dt = self.Tf / self.N
        ctl = _make_controller(model, N, Tf, device)
        if ctl is not None:                          # an unknown model string leaves the simulator without controller
            self.ocp = ctl
            self.MPC_controller = ctl
        if starting_state is None:
            starting_pose = [15.0, 0.1, 1.0, 0]
            starting_velocity = [8.0, 0.0, 0.0]
            starting_steering_angle = [0.0]
        else:
            starting_pose = starting_state[:4]
            starting_velocity = starting_state[4:7]
            starting_steering_angle = starting_state[7]
        if starting_lap is None:
            starting_lap = 0
        self.pose = np.array(starting_pose, dtype=np.float64)
        self.vel = np.array(starting_velocity, dtype=np.float64)
        self.steering = np.array(starting_steering_angle, dtype=np.float64)
        self.lap = starting_lap
        self.waypoint_generator = SkidpadPlanner(target_vel=starting_velocity[0], Nt=self.N, dt=self.dt, device=device)
        self.planned_references = np.zeros([self.N, 4])
        self.dynamics = Dynamics(self.dt, device=device)
        self.figures = figures

    @property
    def full_state(self):
        return np.hstack((self.pose, self.vel, self.steering))

    @full_state.setter
    def full_state(self, new_state):
        self.pose = new_state[:4]
        self.vel = new_state[4:7]
        self.steering = new_state[7]

    @property
    def red_state(self):
        return np.hstack((self.pose, self.vel[1:], self.steering))

    def lapcounter(self, new_pose_x):
        if new_pose_x > path_planning.center and self.pose[0] <= path_planning.center:
            self.lap += 1

    def get_waypoints(self):
        heading = np.arctan2(self.pose[3], self.pose[2])
        return self.waypoint_generator.request_waypoints(self.pose[0], self.pose[1], heading, self.lap)

    def step(self):
        waypoints, speeds, progress, heading_derotation = self.get_waypoints()
        status, trajectory, inputs = self.MPC_controller.optimize(self.red_state, waypoints, speeds)
        steer = inputs[0]
        log.debug("steer: %s", steer)
        new_state = self.dynamics.rk4_integraton(self.full_state, steer)
        self.lapcounter(new_state[0])
        self.full_state = new_state
        self.planned_references = waypoints
        self.planned_trajectory = trajectory
        self.trig_violation = np.linalg.norm(self.planned_trajectory[:, 2:4], axis=1) - 1

    def dynamics_step(self, input):
        new_state = self.dynamics.rk4_integraton(self.full_state, input)
        self.lapcounter(new_state[0])
        self.full_state = new_state
