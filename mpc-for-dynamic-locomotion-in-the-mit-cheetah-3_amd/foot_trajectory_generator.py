"""Swing-foot trajectories (host glue; feeds the swing-leg lever arms of the QP, which multiply a zero force).

Call surface of the reference's ``FootTrajectoryGenerator`` (src/foot_trajectory_generator.py:4-157):
``generate_feet_trajectories_at_time(time, foot) -> {'pos','vel','acc'}`` (6-vectors: angle, position).
Cubic blend in x/y/angle and a quartic bump in z over the first 80 % of the single-support phase.
"""
from __future__ import annotations

import numpy as np

from .footstep_planner import LEGS, _LEG_INDEX


class FootTrajectoryGenerator:
    def __init__(self, footstep_planner, params):
        self.dt = params["world_time_step"]
        self.step_height = params["step_height"]
        self.footstep_planner = footstep_planner
        self.plan = footstep_planner.plan

    def _endpoints(self, step, k):
        p = self.footstep_planner
        nxt = min(step + 1, len(p.pos) - 1)            # past the plan: target = start (ftg.py:35-40)
        a0 = np.array((0.0, 0.0, p.ang[step])); a1 = np.array((0.0, 0.0, p.ang[nxt]))
        return p.pos[step, k].copy(), p.pos[nxt, k].copy(), a0, a1

    def generate_feet_trajectories_at_time(self, time, foot):
        p = self.footstep_planner
        k = _LEG_INDEX[foot]
        step = p.get_step_index_at_time(time)
        t = time - p.get_start_time(step)
        T = int(p.ss[step])
        start, target, a0, a1 = self._endpoints(step, k)
        zero = np.zeros(6)
        if step == 0:                                   # standing phase (ftg.py:42-47)
            return {"pos": np.hstack((a0, start)), "vel": zero, "acc": zero.copy()}
        t_swing = 0.80 * T                              # land before the end of single support (ftg.py:51)
        if t >= T:
            # reference side effect (ftg.py:53-54): once queried in double support the step is marked all-stance
            p.feet_id[step] = 1
            return {"pos": np.hstack((a1, target)), "vel": zero, "acc": zero.copy()}
        if t >= t_swing:
            return {"pos": np.hstack((a1, target)), "vel": zero, "acc": zero.copy()}
        # cubic s(t) = 3 (t/Ts)^2 - 2 (t/Ts)^3 in the plane and for the angle (ftg.py:69-77)
        c3, c2 = -2.0 / t_swing ** 3, 3.0 / t_swing ** 2
        s0 = c3 * t ** 3 + c2 * t ** 2
        s1 = (3 * c3 * t ** 2 + 2 * c2 * t) / self.dt
        s2 = (6 * c3 * t + 2 * c2) / self.dt ** 2
        dp, da = target - start, a1 - a0
        pos, vel, acc = start + dp * s0, dp * s1, dp * s2
        # quartic bump of height step_height, zero at both ends (ftg.py:79-87)
        h = self.step_height
        q4, q3, q2 = 16 * h / t_swing ** 4, -32 * h / t_swing ** 3, 16 * h / t_swing ** 2
        pos[2] = q4 * t ** 4 + q3 * t ** 3 + q2 * t ** 2 + start[2]
        vel[2] = (4 * q4 * t ** 3 + 3 * q3 * t ** 2 + 2 * q2 * t) / self.dt
        acc[2] = (12 * q4 * t ** 2 + 6 * q3 * t + 2 * q2) / self.dt ** 2
        return {"pos": np.hstack((a0 + da * s0, pos)), "vel": np.hstack((da * s1, vel)), "acc": np.hstack((da * s2, acc))}

    def show_trajectory(self, foot_to_sample, t_start=0, t_end=1000, string_axs="z"):  # pragma: no cover
        import matplotlib.pyplot as plt
        ax = {"x": 0, "y": 1, "z": 2}[string_axs.lower()]
        ts = np.arange(t_start, t_end)
        d = [self.generate_feet_trajectories_at_time(t, foot_to_sample) for t in ts]
        fig, axs = plt.subplots(3, 1, figsize=(10, 8))
        for a, key in zip(axs, ("pos", "vel", "acc")):
            a.plot(ts, [x[key][3 + ax] for x in d]); a.set_ylabel(key); a.grid()
        plt.suptitle(f"{foot_to_sample} along {string_axs}")
        plt.show()
