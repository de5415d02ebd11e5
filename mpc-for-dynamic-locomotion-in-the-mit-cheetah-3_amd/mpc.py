"""Single-robot controller surface of the reference, served by the batched HIP engine.

``MPC(lite3, initial, footstep_planner, params)`` / ``MPC.solve(t, logger)`` / ``MPC.update_r_num(...)`` keep the
reference's names, argument meaning, returned dict, public attributes, logger hooks and per-tick reference
roll-forward (src/mpc.py:25-318).  What changes is the inside of ``solve``: instead of seven ``opt.set_value`` calls
into a CasADi ``Opti('conic')`` model and an OSQP solve (src/mpc.py:242-258), the same parameter values are packed
into the compact operator tuple and handed to the engine as a batch of one (``MPCFleet`` does the same for many
robots per tick).  There is no CPU solver here: constructing an ``MPC`` without the HIP library and a GPU raises.
"""
from __future__ import annotations

import numpy as np

from .engine import MPCBatch
from .foot_trajectory_generator import FootTrajectoryGenerator
from .footstep_planner import LEGS


class MPCProblemBuilder:
    """Host-side parameter construction of ``MPC.solve`` (src/mpc.py:176-254): x0, x_des, lever arms, contact mask."""

    def __init__(self, initial, footstep_planner, params):
        self.params = params
        self.N = params["N"]
        self.delta = params["world_time_step"]
        self.h = params["h"]
        self.mu = params["µ"]
        self.initial = initial
        self.footstep_planner = footstep_planner
        # the reference aliases the caller's array and overwrites z with h in place (src/mpc.py:36-37); kept
        self.com_pos_start = initial["com_position"]
        self.com_pos_start[2] = self.h
        self.yaw_start = initial["yaw"]
        self.trajectory_generator = FootTrajectoryGenerator(footstep_planner=footstep_planner, params=params)

    def reference_velocity(self, t):
        """src/mpc.py:178-183: references are zeroed on the last plan step."""
        v, w = self.params["v_com_ref"], self.params["theta_dot"]
        if self.footstep_planner.get_step_index_at_time(t) == self.params["total_steps"] - 1:
            return v * 0, w * 0
        return v, w

    def update_r_num(self, time, leg_name, next_com):
        """Lever arm of one leg at a future tick (src/mpc.py:306-318): swing foot on its trajectory, stance foot on the plan."""
        p = self.footstep_planner
        gait = p.get_phase_at_time(time)
        if p.is_swing(leg_name, gait) == 1:
            leg_pos = self.trajectory_generator.generate_feet_trajectories_at_time(time, leg_name)["pos"][3:]
        else:
            leg_pos = np.asarray(p.plan[p.get_step_index_at_time(time)]["pos"][leg_name], dtype=float)
        return leg_pos - next_com

    def build(self, t, current_state):
        N, d = self.N, self.delta
        v_ref, omega = self.reference_velocity(t)
        x0 = np.concatenate([np.asarray(current_state["TORSO"]["pos"], float), np.asarray(current_state["com"]["pos"], float),
                             np.asarray(current_state["TORSO"]["vel"], float), np.asarray(current_state["com"]["vel"], float),
                             [self.params["g"]]])                                             # src/mpc.py:190-198
        k = np.arange(N + 1)[:, None]
        xdes = np.zeros((N + 1, 13))                                                          # src/mpc.py:202-214
        xdes[:, 0], xdes[:, 1] = self.initial["roll"], self.initial["pitch"]
        xdes[:, 2] = self.yaw_start + k[:, 0] * omega * d
        xdes[:, 3:6] = np.asarray(self.com_pos_start, float)[None, :] + k * d * np.asarray(v_ref, float)[None, :]
        xdes[:, 8] = omega
        xdes[:, 9:12] = v_ref
        xdes[:, 12] = self.params["g"]
        r = np.zeros((N, 4, 3))                                                               # src/mpc.py:218-239
        com = np.asarray(current_state["com"]["pos"], float)
        r[0] = [np.asarray(current_state[l]["pos"], float)[3:] - com for l in LEGS]
        for i in range(1, N):
            r[i] = [self.update_r_num(t + i, l, xdes[i, 3:6]) for l in LEGS]
        contact = self.footstep_planner.contact_mask(t, N)                                    # src/mpc.py:248-252
        return x0, r, contact, xdes, v_ref, omega

    def advance_reference(self, v_ref, omega):
        """src/mpc.py:261-262 (in place: `initial['com_position']` moves with it, as in the reference)."""
        self.com_pos_start += v_ref * self.delta
        self.yaw_start += omega * self.delta


class MPC:
    def __init__(self, lite3, initial, footstep_planner, params, device=0, precision="mixed", **engine_overrides):
        self.params = params
        self.lite3 = lite3
        self.N = params["N"]
        self.delta = params["world_time_step"]
        self.h = params["h"]
        self.mu = params["µ"]
        self.initial = initial
        self.footstep_planner = footstep_planner
        self._builder = MPCProblemBuilder(initial, footstep_planner, params)
        self.trajectory_generator = self._builder.trajectory_generator
        self.m = 8.885                                    # src/mpc.py:71 (engine default; kept as an attribute)
        self._solver = self._make_solver(device, precision, engine_overrides)
        self.status = None

    def _make_solver(self, device, precision, engine_overrides):
        """The HIP engine.  (tests/ override this hook to drive the same surface with the CPU checker.)"""
        # The reference seeds every solve with the previous solution (`opt.set_initial(U, sol.value(U))`,
        # src/mpc.py:270-271).  Pass warm_start=True for that (MPCQP_FLAG_WARM_START | MPCQP_FLAG_WARM_SHIFT: the engine
        # keeps the last solution and its multipliers and moves both up by one stage, the horizon step being the control
        # tick, src/main.py:32 / src/mpc.py:33).  Same optimum either way; off by default ("Warm start").
        if engine_overrides.get("warm_start"):
            engine_overrides.setdefault("warm_shift", True)
        return MPCBatch(N=self.N, delta=self.delta, device=device, io_dtype="f64", precision=precision, **engine_overrides)

    def _solve_one(self, x0, r, contact, xdes):
        import torch
        dev = self._solver.upload({"x0": x0[None], "r": r[None], "contact": contact[None], "xdes": xdes[None],
                                   "mu": np.array([float(self.mu)])})
        out = self._solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
        torch.cuda.synchronize(self._solver.device)
        return out["u"][0].cpu().numpy(), out["X"][0].cpu().numpy(), int(out["status"][0].item())

    # the reference keeps these on the instance and mutates them every tick
    @property
    def com_pos_start(self):
        return self._builder.com_pos_start

    @property
    def yaw_start(self):
        return self._builder.yaw_start

    def update_r_num(self, time, leg_name, next_com):
        return self._builder.update_r_num(time, leg_name, next_com)

    def solve(self, t, logger):
        current_state = self.lite3.retrieve_state()
        x0, r, contact, xdes, v_ref, omega = self._builder.build(t, current_state)
        self.x = x0.reshape(13, 1)
        u, Xs, self.status = self._solve_one(x0, r, contact, xdes)
        U = u.T.copy()                                    # 12 x N, like sol.value(U)
        X = Xs.T.copy()                                   # 13 x (N+1)
        self._builder.advance_reference(v_ref, omega)
        self.x_log = X[:-1, :]                            # src/mpc.py:265-268
        self.x_plot = X[3:6, :]
        self.u = U[:, 0]
        self.u_plot = U
        forces = {"FL_FOOT": self.u[0:3], "FR_FOOT": self.u[3:6], "HL_FOOT": self.u[6:9], "HR_FOOT": self.u[9:12]}
        forces_plot = np.array([U[2, :], U[5, :], U[8, :], U[11, :]])
        x_curr = x0[:12].tolist()
        x_des_num = xdes.T
        logger.log_tracking_data(x_curr, x_des_num[:-1, 0])                                   # src/mpc.py:295
        if t == 0 or t == 80:                                                                 # src/mpc.py:297-301
            logger.log_mpc_predictions(self.x_log, x_des_num[:-1, :], forces_plot, t)
        return forces


class MPCFleet:
    """Many robots, one tick: every robot's ``MPCProblemBuilder`` output is stacked and solved in ONE engine call."""

    def __init__(self, builders, device=0, precision="mixed", **engine_overrides):
        self.builders = list(builders)
        b0 = self.builders[0]
        self._solver = MPCBatch(N=b0.N, delta=b0.delta, device=device, io_dtype="f64", precision=precision, **engine_overrides)

    def solve(self, t, states):
        import torch
        parts = [b.build(t, s) for b, s in zip(self.builders, states)]
        batch = {"x0": np.stack([p[0] for p in parts]), "r": np.stack([p[1] for p in parts]),
                 "contact": np.stack([p[2] for p in parts]), "xdes": np.stack([p[3] for p in parts]),
                 "mu": np.array([float(b.mu) for b in self.builders])}
        dev = self._solver.upload(batch)
        out = self._solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
        torch.cuda.synchronize(self._solver.device)
        for b, p in zip(self.builders, parts):
            b.advance_reference(p[4], p[5])
        return {"u": out["u"].cpu().numpy(), "X": out["X"].cpu().numpy(), "status": out["status"].cpu().numpy(), "inputs": batch}
