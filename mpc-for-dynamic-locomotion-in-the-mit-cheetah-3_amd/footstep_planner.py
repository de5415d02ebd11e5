"""Footstep plan + the planner queries the MPC path needs (host glue feeding the QP engine).

Call surface kept from the reference's ``FootstepPlanner`` (src/footstep_planner.py:5-256): constructor
``(initial_configuration, params, show=...)``, attribute ``plan`` (list of dicts with ``pos``, ``ang``,
``ss_duration``, ``ds_duration``, ``feet_id``), ``get_step_index_at_time``, ``get_start_time``,
``get_phase_at_time``, ``is_swing``.  Internals are array based (the reference walks the list linearly for every
query, src/footstep_planner.py:226-237): step boundaries are a cumulative table, so queries are O(log S) and the
horizon-wide contact mask / stance positions the engine needs are produced in one vectorised call.
"""
from __future__ import annotations

import numpy as np

LEGS = ("FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT")     # FL, FR, HL, HR (src/footstep_planner.py:47)
_LEG_INDEX = {name: i for i, name in enumerate(LEGS)}
# sign pattern of (torso_displacement/2, leg_displacement_y) per foot (src/footstep_planner.py:93-152)
_FOOT_SIGNS = np.array([[+1, -1], [+1, +1], [-1, -1], [-1, +1]], dtype=float)


def _rot2(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s], [s, c]])


class FootstepPlanner:
    def __init__(self, initial_configuration, params, show=False):
        ss, ds = int(params["ss_duration"]), int(params["ds_duration"])
        dt = params["world_time_step"]
        v_ref = np.asarray(params["v_com_ref"], dtype=float)
        omega_ref = float(params["theta_dot"])
        total_steps = int(params["total_steps"])
        stance_flags = np.asarray(params["first_swing"]).astype(int).copy()   # 1 = stays down during the step
        feet0 = np.array([np.asarray(initial_configuration[l], dtype=float) for l in LEGS])   # [4,3]
        theta = float(initial_configuration["yaw"])

        centre = feet0.mean(axis=0)             # the virtual unicycle starts under the feet centroid (:42-43)
        R = _rot2(theta)
        pos_rows, ang_rows, fid_rows, hip_rows = [], [], [], []
        if total_steps == 0:                    # standing: 100 identical all-stance steps (:53-71)
            for _ in range(100):
                pos_rows.append(feet0.copy()); ang_rows.append(theta); fid_rows.append([1, 1, 1, 1])
                hip_rows.append([np.nan, np.nan, params["h"]])
        for j in range(total_steps):
            if j >= 1:                          # the unicycle only moves from the second step on (:78-84)
                for _ in range(ss + ds):
                    theta += omega_ref * dt
                    R = _rot2(theta)
                    centre[:2] += R @ v_ref[:2] * dt
            torso = R @ (feet0[0, :2] - feet0[2, :2])          # FL - HL
            half_width = R @ (feet0[3, :2] - feet0[2, :2]) / 2.0   # (HR - HL)/2
            fresh = np.empty((4, 3))
            fresh[:, :2] = centre[:2] + _FOOT_SIGNS[:, :1] * torso / 2.0 + _FOOT_SIGNS[:, 1:] * half_width
            fresh[:, 2] = centre[2]
            if j >= 1:                          # feet that stay down keep their previous placement (:91-122)
                keep = stance_flags.astype(bool)
                row = np.where(keep[:, None], pos_rows[-1], fresh)
            else:
                row = fresh
            pos_rows.append(row)
            ang_rows.append(theta)
            hip_rows.append([centre[0] - torso[0] / 2.0, centre[1] - torso[1] / 2.0, params["h"]])
            fid_rows.append([1, 1, 1, 1] if j == 0 else stance_flags.tolist())
            if j > 0:
                stance_flags = 1 - stance_flags             # alternate with the complement (:176-177)

        self.pos = np.array(pos_rows, dtype=float)          # [S,4,3]
        self.ang = np.array(ang_rows, dtype=float)          # yaw per step
        self.feet_id = np.array(fid_rows, dtype=int)        # [S,4]; mutable (see FootTrajectoryGenerator)
        self.hip = np.array(hip_rows, dtype=float)
        S = len(self.pos)
        self.ss = np.full(S, ss, dtype=int)
        self.ds = np.full(S, ds, dtype=int)
        self.step_end = np.cumsum(self.ss + self.ds)        # tick at which step i ends
        self.plan = _PlanView(self)
        if show:
            self.show()

    # ------------------------------------------------------------------ reference queries (:226-256)
    def get_step_index_at_time(self, time):
        i = int(np.searchsorted(self.step_end, time, side="right"))
        return min(i, len(self.step_end) - 1)

    def get_start_time(self, step_index):
        return int(self.step_end[step_index - 1]) if step_index > 0 else 0

    def get_phase_at_time(self, time):
        i = self.get_step_index_at_time(time)
        if time - self.get_start_time(i) < self.ss[i]:
            return self.feet_id[i].tolist()
        return [1, 1, 1, 1]

    def is_swing(self, leg_name, gait):
        return 1 - gait[_LEG_INDEX[leg_name]]

    # ------------------------------------------------------------------ vectorised forms used by the batched path
    def step_indices(self, ticks):
        ticks = np.asarray(ticks)
        return np.minimum(np.searchsorted(self.step_end, ticks, side="right"), len(self.step_end) - 1)

    def contact_mask(self, t, N):
        """contact[N,4] for stages t..t+N-1 (the `1 - swing_inverted` of src/mpc.py:248-252)."""
        ticks = t + np.arange(N)
        idx = self.step_indices(ticks)
        start = np.where(idx > 0, self.step_end[np.maximum(idx - 1, 0)], 0)
        in_ss = (ticks - start) < self.ss[idx]
        return np.where(in_ss[:, None], self.feet_id[idx], 1).astype(np.uint8)

    def show(self):  # pragma: no cover - visual aid only
        import matplotlib.pyplot as plt
        plt.figure(figsize=(8, 6))
        for k, (name, col) in enumerate(zip(LEGS, "gmbr")):
            plt.plot(self.pos[:, k, 0], self.pos[:, k, 1], col + "o", label=name)
        plt.plot(self.hip[:, 0], self.hip[:, 1], "k.", label="hip")
        plt.xlabel("x (m)"); plt.ylabel("y (m)"); plt.legend(); plt.grid(True); plt.title("Footstep plan")
        plt.show()


class _StepView(dict):
    """Dict-shaped view of one step so reference-style code (`plan[i]['pos'][leg]`, `plan[i]['feet_id'] = ...`) works."""

    def __init__(self, owner, i):
        super().__init__()
        self._o, self._i = owner, i

    def __getitem__(self, key):
        o, i = self._o, self._i
        if key == "pos":
            d = {l: o.pos[i, k].tolist() for k, l in enumerate(LEGS)}
            d["hip"] = o.hip[i].tolist()
            return d
        if key == "ang":
            return np.array((0.0, 0.0, o.ang[i]))
        if key == "ss_duration":
            return int(o.ss[i])
        if key == "ds_duration":
            return int(o.ds[i])
        if key == "feet_id":
            return o.feet_id[i].tolist()
        raise KeyError(key)

    def __setitem__(self, key, value):
        if key != "feet_id":
            raise KeyError(f"plan entries are read-only except 'feet_id' (got {key!r})")
        self._o.feet_id[self._i] = np.asarray(value, dtype=int)

    def keys(self):
        return ("pos", "ang", "ss_duration", "ds_duration", "feet_id")


class _PlanView:
    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return len(self._o.pos)

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return _StepView(self._o, i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))
