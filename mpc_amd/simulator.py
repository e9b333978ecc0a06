"""Step-reference closed-loop simulator with the class surface of the reference's simulator.py (StepSimulator).
Planner, controllers, plant and EKF all run on the GPU through libmpc_amd; the interactive __main__ and the
matplotlib output of the reference are not reproduced."""
from __future__ import annotations

import logging

import numpy as np

from .continuous_dynamics import Dynamics, indices
from .controllers import LOcp, LPVOcp, OFLOcp, read_parameters
from .EKF import CarEKF
from .skidpad_simulator import SkidpadSimulator  # noqa: F401  (re-exported like the reference module)
from .utils.step_planning import StepPlanner

log = logging.getLogger("mpc_amd")


class StepSimulator:
    def __init__(self, N, Tf, acados_print_level=0, starting_state=None, figures=False, model="LPV", device=None):
        self.N = N
        self.Tf = Tf
        self.dt = self.Tf / self.N
        self.model = model
        if model == "NL":
            from .NLMPC import NLOcp
            self.ocp = NLOcp(self.N, self.Tf, device=device); self.disturbed = False
        elif model == "L":
            self.ocp = LOcp(self.N, self.Tf, device=device); self.disturbed = False
        elif model == "LPV":
            self.ocp = LPVOcp(self.N, self.Tf, device=device); self.disturbed = False
        elif model == "OFL":
            self.ocp = OFLOcp(self.N, self.Tf, device=device); self.disturbed = True
        elif model == "none":
            self.disturbed = True
        else:
            log.error("Model not recognized. Please choose 'NL', 'L', or 'LPV'.")
            return
        if model != "none":
            self.MPC_controller = self.ocp
        if starting_state is None:
            starting_pose = [15.0, 0.1, 1.0, 0]
            starting_velocity = [8.0, 0.0, 0.0]
            starting_steering_angle = [0.0]
            starting_disturbances = [0.0, 0.0]
        else:
            starting_pose = starting_state[:4]
            starting_velocity = starting_state[4:7]
            starting_steering_angle = starting_state[7]
            if self.disturbed:
                starting_disturbances = [starting_state[indices["steering_dist"]], starting_state[indices["d_f"]]]
        self.pose = np.array(starting_pose, dtype=np.float64)
        self.vel = np.array(starting_velocity, dtype=np.float64)
        self.steering = np.array(starting_steering_angle, dtype=np.float64)
        if self.disturbed:
            self.disturbances = np.array(starting_disturbances, dtype=np.float64)
        self.waypoint_generator = StepPlanner(target_vel=starting_velocity[0], Nt=self.N, dt=self.dt, device=device)
        self.planned_references = np.zeros([self.N, 4])
        self.dynamics = Dynamics(self.dt, disturbance=self.disturbed, device=device)
        self.figures = figures
        self._device = device

    @property
    def full_state(self):
        if self.disturbed:
            return np.hstack((self.pose, self.vel, self.steering, self.disturbances))
        return np.hstack((self.pose, self.vel, self.steering))

    @full_state.setter
    def full_state(self, new_state):
        self.pose = new_state[:4]
        self.vel = new_state[4:7]
        self.steering = new_state[7]
        if self.disturbed:
            self.disturbances = new_state[[indices["steering_dist"], indices["d_f"]]]

    @property
    def red_state(self):
        return np.hstack((self.pose, self.vel[1:], self.steering))

    def get_waypoints(self, x=None, y=None, heading=None):
        x = self.pose[0] if x is None else x
        y = self.pose[1] if y is None else y
        heading = np.arctan2(self.pose[3], self.pose[2]) if heading is None else heading
        return self.waypoint_generator.request_waypoints(x, y, heading)

    def _advance(self, steer, waypoints, trajectory):
        new_state = self.dynamics.rk4_integraton(self.full_state, steer)
        self.full_state = new_state
        self.planned_references = waypoints
        if trajectory is not None:
            self.planned_trajectory = trajectory
        return new_state

    def simulate(self, n_steps):
        states = np.zeros((n_steps, self.dynamics.nx)); inputs_h = np.zeros((n_steps, 1)); reference = np.zeros((n_steps, 4))
        for i in range(n_steps):
            waypoints, speeds, progress, heading_derotation, absolute_waypoints = self.get_waypoints()
            status, trajectory, inputs = self.MPC_controller.optimize(self.red_state, waypoints, speeds)
            steer = inputs[0]
            new_state = self._advance(steer, waypoints, trajectory)
            reference[i, :] = absolute_waypoints[0, :]
            states[i, :] = new_state
            inputs_h[i, :] = steer
        return states, inputs_h, reference

    def simulate_of(self, n_steps, initial_state_estimate=None):
        """Output-feedback loop: the controller sees the EKF estimate (simulator.py:206-281)."""
        states = np.zeros((n_steps, 10)); inputs_h = np.zeros((n_steps, 1)); estimates = np.zeros((n_steps, 10))
        planned_path = np.zeros((n_steps, 2))
        if initial_state_estimate is None:
            initial_state_estimate = [-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        SE = CarEKF(self.dt, True, inital_state=initial_state_estimate, device=self._device)
        for i in range(n_steps):
            x_est = SE.x_est
            heading_est = np.arctan2(x_est[indices["heading_sin"]], x_est[indices["heading_cos"]])
            waypoints, speeds, progress, heading_derotation, absolute_waypoints = self.get_waypoints(
                x_est[indices["pos_x"]], x_est[indices["pos_y"]], heading_est)
            planned_path[i, :] = absolute_waypoints[0, :2]
            estimated_IC = SE.estimated_red_state()
            d_est = estimated_IC[-1]
            status, trajectory, inputs = self.MPC_controller.optimize(estimated_IC, waypoints, speeds, d_est)
            steer = inputs[0]
            new_state = self._advance(steer, waypoints, trajectory)
            SE.time_update(steer)
            SE.measurement_update(self.dynamics.measure_state_noiseless(self.full_state))
            estimates[i, :] = SE.estimate_full_state()
            states[i, :] = new_state
            inputs_h[i, :] = steer
        return states, inputs_h, estimates, planned_path

    def step(self):
        waypoints, speeds, progress, heading_derotation, _ = self.get_waypoints()
        status, trajectory, inputs = self.MPC_controller.optimize(self.red_state, waypoints, speeds)
        self._advance(inputs[0], waypoints, trajectory)

    def dynamics_step(self, input):
        self.full_state = self.dynamics.rk4_integraton(self.full_state, input)

    def lsim(self, u, n_steps, initial_state_estimate=None):
        """Open-loop input sequence through plant and EKF (simulator.py:334-376)."""
        if isinstance(u, float):
            u = np.ones(n_steps) * u
        states = np.zeros((n_steps, self.dynamics.nx)); inputs_h = np.zeros((n_steps, 1))
        estimates = np.zeros((n_steps, self.dynamics.nx))
        if initial_state_estimate is None:
            initial_state_estimate = [0.0, 0.0, 1.0, 0.0, 8.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        SE = CarEKF(self.dt, True, inital_state=initial_state_estimate, device=self._device)
        for i in range(n_steps):
            self.dynamics_step(u[i])
            states[i, :] = self.full_state
            inputs_h[i, :] = u[i]
            SE.time_update(u[i])
            SE.measurement_update(self.dynamics.measure_state_noiseless(self.full_state))
            estimates[i, :] = SE.estimate_full_state()
        return states, inputs_h, estimates

    def dlqr_sim(self, n_steps, K=None):
        """Saturated LQR baseline with the DARE gain of the L controller (simulator.py:378-428)."""
        if K is None:
            K = self.MPC_controller.K
        states = np.zeros((n_steps, self.dynamics.nx)); inputs_h = np.zeros((n_steps, 1)); reference = np.zeros((n_steps, 4))
        params = read_parameters("parameters_NL.yaml")
        rate_limit = params["model"]["max_steering_rate"]
        angle_limit = params["model"]["max_steering_angle"]
        for i in range(n_steps):
            waypoints, speeds, progress, heading_derotation, absolute_waypoints = self.get_waypoints()
            dt = self.dynamics.dt
            ref_state = np.zeros(5); ref_state[0] = 1 / 500
            effect_state = self.full_state[[1, 3, 5, 6, 7]]
            steer = K @ (ref_state - effect_state)
            steer = np.clip(steer, -rate_limit, rate_limit)
            current_steer = self.full_state[-1]
            steer = np.clip(steer, (-angle_limit - current_steer) / dt, (angle_limit - current_steer) / dt)
            new_state = self._advance(steer, waypoints, None)
            reference[i, :] = absolute_waypoints[0, :]
            states[i, :] = new_state
            inputs_h[i, :] = steer
        return states, inputs_h, reference
