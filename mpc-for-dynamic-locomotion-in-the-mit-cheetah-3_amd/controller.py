"""Controller glue around the MPC path, mirroring the reference's `Lite3Controller` (src/main.py:12-350) without DART.

Kept: the per-tick contract `customPreStep -> ground_controller(t) -> mpc.solve(t, logger)` (src/main.py:130-219), the
solve timing / `mpc_freq` running mean (:194-202), the force log (:216-218), the torque map tau = J^T (-f) (:212-214),
stance/swing leg selection and the foot log (:152-167), the end-of-run log dump (:184-186).
Replaced: the DART world by a kinematic single-rigid-body stand-in (`KinematicLite3`) whose state is advanced with the
model's own discrete dynamics; the operational-space swing controller (out of scope) by a callable.
"""
from __future__ import annotations

import time

import numpy as np

from . import lite3_model
from .foot_trajectory_generator import FootTrajectoryGenerator
from .footstep_planner import LEGS, FootstepPlanner
from .logger import Logger

# src/main.py:31-46 with the benchmark horizon / step (BASELINE config 1: N = 10, dt = 0.03 s).  The reference states its
# step durations in TICKS of its 0.01 s world step (ss 10 = 0.10 s, ds 5 = 0.05 s, src/main.py:35-36,41); config 1 keeps
# those durations IN SECONDS, i.e. 4 and 2 ticks of 0.03 s (0.12 s / 0.06 s).  Keeping the tick counts instead would
# mean 0.3 s of two-leg support per step with a 0.3 s horizon -- a different (and, in this formulation, diverging)
# gait: DESIGN.md section 10.  total_steps is raised so that the 300-tick run walks throughout (50 x 6 ticks).
DEFAULT_PARAMS = {
    "g": -9.81, "h": 0.285, "step_height": 0.08, "ss_duration": 4, "ds_duration": 2, "world_time_step": 0.03,
    "total_steps": 50, "first_swing": np.array([1, 0, 0, 1]), "µ": 1, "N": 10, "dof": 18,
    "v_com_ref": np.array([0.18, 0.0, 0.0]), "theta_dot": 0.0, "log_samples": 300,
}


class KinematicLite3:
    """`retrieve_state()` provider (src/main.py:286-350): SRB state + feet; stance feet stay where the plan puts them,
    swing feet follow the swing trajectory; joint angles by inverse kinematics of the analytic leg model."""

    def __init__(self, planner, trajectory_generator, x0):
        self.planner, self.tg = planner, trajectory_generator
        self.x = np.asarray(x0, float).copy()      # [rpy, com, omega, v]
        self.t = 0

    def _feet(self):
        st = self.planner.get_step_index_at_time(self.t)
        gait = self.planner.plan[st]["feet_id"]
        out = []
        for k, l in enumerate(LEGS):
            out.append(self.planner.pos[st, k] if gait[k] == 1
                       else self.tg.generate_feet_trajectories_at_time(self.t, l)["pos"][3:])
        return np.array(out)

    def retrieve_state(self):
        feet = self._feet()
        s = {l: {"pos": np.concatenate([np.zeros(3), feet[k]]), "vel": np.zeros(6), "acc": np.zeros(6)} for k, l in enumerate(LEGS)}
        s["TORSO"] = {"pos": self.x[0:3].copy(), "vel": self.x[6:9].copy(), "acc": np.zeros(3)}
        s["com"] = {"pos": self.x[3:6].copy(), "vel": self.x[9:12].copy(), "acc": np.zeros(3)}
        return s

    def body_rotation(self):
        from scipy.spatial.transform import Rotation
        return Rotation.from_rotvec(self.x[0:3]).as_matrix()

    def joint_angles(self):
        R = self.body_rotation()
        feet = self._feet()
        return np.array([lite3_model.leg_ik(k, R.T @ (feet[k] - self.x[3:6])) for k in range(4)])


class Lite3Controller:
    def __init__(self, mpc_factory, params=None, x0=None, swing_leg_controller=None):
        self.params = dict(DEFAULT_PARAMS if params is None else params)
        self.time = 0
        feet = np.array([[0.11648, 0.16078, 0.01713], [0.11648, -0.16022, 0.01713],
                         [-0.23252, 0.16078, 0.01713], [-0.23252, -0.16022, 0.01713]])   # tick-0 feet of the reference log
        x0 = np.array([0, 0, 0, 0, 0, self.params["h"], 0, 0, 0, 0, 0, 0], float) if x0 is None else np.asarray(x0, float)
        self.initial = {l: feet[k].copy() for k, l in enumerate(LEGS)}
        self.initial.update(roll=x0[0], pitch=x0[1], yaw=x0[2], com_position=x0[3:6].copy())
        self.footstep_planner = FootstepPlanner(self.initial, self.params, show=False)
        self.trajectory_generator = FootTrajectoryGenerator(self.footstep_planner, self.params)
        self.lite3 = KinematicLite3(self.footstep_planner, self.trajectory_generator, x0)
        self.mpc = mpc_factory(lite3=self, initial=self.initial, footstep_planner=self.footstep_planner, params=self.params)
        self.plot_keys = {"params": self.params, "total_sim_steps": self.params["log_samples"]}
        self.logger = Logger(self.plot_keys)
        self._swing = swing_leg_controller or (lambda leg: (np.zeros(3), self.trajectory_generator
                                                .generate_feet_trajectories_at_time(self.time, leg)["pos"][3:]))

    def retrieve_state(self):
        self.lite3.t = self.time
        return self.lite3.retrieve_state()

    def ground_controller(self, t):
        """src/main.py:193-219."""
        start = time.time()
        forces = self.mpc.solve(t, self.logger)
        mpc_freq = 1.0 / max(time.time() - start, 1e-9)
        self.logger.log["mpc_freq"] = (self.logger.log["mpc_freq"] * self.time + mpc_freq) / (self.time + 1)
        J = lite3_model.world_jacobians(self.lite3.body_rotation(), self.lite3.joint_angles())
        tau = {}
        for leg in LEGS:
            tau[leg] = J[leg].T @ -forces[leg]
            for a, ax in enumerate("xyz"):
                self.logger.log["FORCES"][leg][ax].append(forces[leg][a])
        return tau

    def customPreStep(self):
        """src/main.py:130-188, with the world step replaced by the model's own predicted next state."""
        step_index = self.footstep_planner.get_step_index_at_time(self.time)
        gait = self.footstep_planner.plan[step_index]["feet_id"]
        tau_ground = self.ground_controller(self.time)
        tau = {}
        for j, leg in enumerate(LEGS):
            if gait[j] == 1:
                tau[leg] = tau_ground[leg]
                p_des = self.footstep_planner.plan[step_index]["pos"][leg]
            else:
                tau[leg], p_des = self._swing(leg)
            state = self.retrieve_state()
            self.logger.log_feet_data(state[leg]["pos"][3:6], p_des, leg)
        for leg in LEGS:
            for n, joint in enumerate(("HipX", "HipY", "Knee")):
                self.logger.log["CONTROL EFFORT"][leg][f"{leg[:2]}_{joint}"].append(tau[leg][n])
        self.logger.log["time array"].append(self.time)
        self.lite3.x = np.asarray(self.mpc.x_log[:, 1], float).copy()      # kinematic "world step"
        self.time += 1
        return tau
