"""Synthetic Lite3 QP batches for the benchmark configurations.

Produces the operator tuple that crosses into the solver -- the batched, compact form of the seven
``opt.set_value`` parameters of the reference (src/mpc.py:242-255):

    x0[B,13]  r[B,N,4,3]  contact[B,N,4] (u8, 1 = stance)  xdes[B,N+1,13]  mu[B]

Pure numpy (host side); the arrays are uploaded once and stay resident in HBM for the timed region.
"""
from __future__ import annotations

import numpy as np

# Lite3 constants (src/main.py:32-33, src/mpc.py:45-46,71-76) and nominal stance geometry taken from the
# reference's committed log at tick 0 (FEET POS.actual[0] - com[0]); SURVEY.md section 8(d).
G_ACC = -9.81
H_COM = 0.285
FOOT_Z = 0.01713
NOMINAL_FEET = np.array([
    [0.11648, 0.16078, -0.26780],    # FL
    [0.11648, -0.16022, -0.26780],   # FR
    [-0.23252, 0.16078, -0.26780],   # HL
    [-0.23252, -0.16022, -0.26780],  # HR
])
V_REF_BODY = np.array([0.18, 0.0, 0.0])          # src/main.py:43
SS_TICKS, DS_TICKS = 10, 5                       # src/main.py:35-36

# first_swing encodings as the reference's planner uses them (1 = stance during the step's ss phase,
# src/footstep_planner.py:11-13,159-177): the pattern alternates with its complement every step.
GAITS = {
    "trot": (1, 0, 0, 1),
    "pronk": (0, 0, 0, 0),
    "amble": (1, 0, 1, 0),
    "gallop": (0, 0, 1, 1),          # the committed default, "pseudo-galloping" (src/main.py:39)
}


def contact_schedule(gait_ids, t0, N, gaits=None):
    """contact[B,N,4]: phase semantics of src/footstep_planner.py:239-246 for an endless two-beat gait.

    A step lasts ss+ds ticks; during the first ss ticks the step's feet_id applies, then all four feet
    are in stance.  feet_id alternates with its complement each step (src/footstep_planner.py:176-177).
    gait_ids[B] index into `gaits` (list of 4-tuples); t0[B] is the tick offset of stage 0.
    """
    gaits = np.asarray(gaits if gaits is not None else list(GAITS.values()), dtype=np.uint8)
    B = len(t0)
    t = np.asarray(t0)[:, None] + np.arange(N)[None, :]               # [B,N]
    step = t // (SS_TICKS + DS_TICKS)
    in_step = t % (SS_TICKS + DS_TICKS)
    base = gaits[np.asarray(gait_ids)]                                # [B,4]
    fid = np.where((step % 2 == 0)[:, :, None], base[:, None, :], 1 - base[:, None, :])
    contact = np.where((in_step < SS_TICKS)[:, :, None], fid, 1).astype(np.uint8)
    assert contact.shape == (B, N, 4)
    return contact


def make_batch(B, N=10, delta=0.03, seed=20250808, gait_names=("trot",), mus=(1.0,), dtype=np.float64):
    """State distribution of SURVEY.md section 8(d) config 2 (and 3-5 via gait_names / mus / N / seed)."""
    rng = np.random.default_rng(seed)
    roll = rng.uniform(-0.1, 0.1, B)
    pitch = rng.uniform(-0.1, 0.1, B)
    yaw = rng.uniform(-np.pi, np.pi, B)
    com = np.array([0.0, 0.0, H_COM])[None, :] + rng.normal(0.0, 0.02, (B, 3)) * np.array([1.0, 1.0, 0.5])
    omega = rng.normal(0.0, 0.2, (B, 3))
    c, s = np.cos(yaw), np.sin(yaw)
    v_ref = np.stack([c * V_REF_BODY[0] - s * V_REF_BODY[1], s * V_REF_BODY[0] + c * V_REF_BODY[1],
                      np.full(B, V_REF_BODY[2])], axis=1)
    v = v_ref + rng.normal(0.0, 0.1, (B, 3))
    feet_xy = np.stack([c[:, None] * NOMINAL_FEET[None, :, 0] - s[:, None] * NOMINAL_FEET[None, :, 1],
                        s[:, None] * NOMINAL_FEET[None, :, 0] + c[:, None] * NOMINAL_FEET[None, :, 1]], axis=2)
    feet = np.concatenate([com[:, None, :2] + feet_xy + rng.normal(0.0, 0.01, (B, 4, 2)),
                           np.full((B, 4, 1), FOOT_Z)], axis=2)      # [B,4,3] world, fixed over the horizon
    t0 = rng.integers(0, 30, B)
    gait_ids = rng.integers(0, len(gait_names), B)
    mu = np.asarray(mus, float)[rng.integers(0, len(mus), B)]

    x0 = np.concatenate([roll[:, None], pitch[:, None], yaw[:, None], com, omega, v,
                         np.full((B, 1), G_ACC)], axis=1)            # src/mpc.py:189-198
    # x_des, src/mpc.py:202-214 with roll0 = pitch0 = 0, omega_ref = 0, yaw_start = yaw, com_start = (x,y,h)
    k = np.arange(N + 1)[None, :, None]
    com_start = np.concatenate([com[:, :2], np.full((B, 1), H_COM)], axis=1)
    xdes = np.zeros((B, N + 1, 13))
    xdes[:, :, 2] = yaw[:, None]
    xdes[:, :, 3:6] = com_start[:, None, :] + k * delta * v_ref[:, None, :]
    xdes[:, :, 9:12] = v_ref[:, None, :]
    xdes[:, :, 12] = G_ACC
    # lever arms, src/mpc.py:218-239: stage 0 from the measured com, stage k>=1 from the reference com
    r = feet[:, None, :, :] - xdes[:, :N, None, 3:6]
    r[:, 0] = feet - com[:, None, :]
    contact = contact_schedule(gait_ids, t0, N, gaits=[GAITS[g] for g in gait_names])
    return {
        "x0": np.ascontiguousarray(x0, dtype=dtype),
        "r": np.ascontiguousarray(r, dtype=dtype),
        "contact": np.ascontiguousarray(contact, dtype=np.uint8),
        "xdes": np.ascontiguousarray(xdes, dtype=dtype),
        "mu": np.ascontiguousarray(mu, dtype=dtype),
        "gait_ids": gait_ids, "t0": t0,
    }


def config2(B=1024, dtype=np.float64):
    """B=1024 randomised states, trot, N=10, delta=0.03, mu=1 (seed 20250808)."""
    return make_batch(B, 10, 0.03, 20250808, ("trot",), (1.0,), dtype)


def config3(B=4096, dtype=np.float64):
    """B=4096 mixed gaits + friction sweep, N=10 (seed 20250809) -- the configuration the metric is quoted on."""
    return make_batch(B, 10, 0.03, 20250809, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0), dtype)


def config4(B=65536, dtype=np.float64):
    """B=65536 of the config-3 distribution (seed 20250810), sharded contiguously across ranks."""
    return make_batch(B, 10, 0.03, 20250810, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0), dtype)


def config5(B=4096, dtype=np.float64):
    """B=4096, N=20 (seed 20250811)."""
    return make_batch(B, 20, 0.03, 20250811, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0), dtype)


# ----------------------------------------------------------------------------------------------------------------------
# Compact gait descriptors (include/mpcqp.h, mpcqp_solve_batch_gait) and their host expansion
# ----------------------------------------------------------------------------------------------------------------------
def make_gait_batch(B, N=10, delta=0.03, seed=20250812, gait_names=("trot", "pronk", "amble", "gallop"), mus=(0.3, 0.5, 0.7, 1.0),
                    stride=0.06, steps=2):
    """Same state distribution as `make_batch`, but described the way the controller knows it: measured feet, the
    planned footholds of the current and the following `steps - 1` steps (swing feet land `stride` ahead along the heading, the
    gait alternates feet_id with its complement step by step, src/footstep_planner.py:159-177), the gait clock."""
    rng = np.random.default_rng(seed)
    base = make_batch(B, N, delta, seed, gait_names, mus)
    x0 = base["x0"]
    yaw = x0[:, 2]
    c, s = np.cos(yaw), np.sin(yaw)
    v_ref = np.stack([c * V_REF_BODY[0] - s * V_REF_BODY[1], s * V_REF_BODY[0] + c * V_REF_BODY[1], np.zeros(B)], axis=1)
    com_start = np.concatenate([x0[:, 3:5], np.full((B, 1), H_COM)], axis=1)
    ref = np.concatenate([np.zeros((B, 2)), yaw[:, None], com_start, v_ref, np.zeros((B, 1))], axis=1)
    feet0 = base["r"][:, 0] + x0[:, None, 3:6]                       # measured feet (stage-0 lever arm + com)
    gaits = np.asarray([GAITS[g] for g in gait_names], dtype=np.uint8)
    fid0 = gaits[base["gait_ids"]]
    t0 = base["t0"]
    step_par = (t0 // (SS_TICKS + DS_TICKS)) % 2
    fid_cur = np.where(step_par[:, None] == 0, fid0, 1 - fid0).astype(np.uint8)
    heading = np.stack([c, s, np.zeros(B)], axis=1)
    fids = [fid_cur]
    fhs = [feet0 + rng.normal(0.0, 0.003, (B, 4, 3)) * np.array([1.0, 1.0, 0.0])]
    for _ in range(1, steps):
        fhs.append(np.where(fids[-1][:, :, None] == 0, fhs[-1] + stride * heading[:, None, :], fhs[-1]))   # feet swinging now land ahead
        fids.append((1 - fids[-1]).astype(np.uint8))
    gait = np.stack([t0 % (SS_TICKS + DS_TICKS), np.full(B, SS_TICKS), np.full(B, DS_TICKS), np.zeros(B, int)], axis=1).astype(np.int32)
    return {"x0": x0, "ref": ref, "feet0": feet0, "footholds": np.stack(fhs, axis=1), "gait": gait, "feet_id": np.stack(fids, axis=1).astype(np.uint8),
            "mu": base["mu"]}


def expand_gait_batch(g, N=10, delta=0.03):
    """numpy expansion of the descriptors into the operator tuple (the host loop of src/mpc.py:178-254, vectorised): stage k lies in
    step min((t_in_step + k) / (ss + ds), S - 1) of the S described steps (include/mpcqp.h)."""
    x0, ref = np.asarray(g["x0"], float), np.asarray(g["ref"], float)
    B, S = len(x0), np.asarray(g["footholds"]).shape[1]
    k = np.arange(N + 1)[None, :, None]
    xdes = np.zeros((B, N + 1, 13))
    xdes[:, :, 0], xdes[:, :, 1] = ref[:, None, 0], ref[:, None, 1]
    xdes[:, :, 2] = ref[:, None, 2] + k[:, :, 0] * delta * ref[:, None, 9]
    xdes[:, :, 3:6] = ref[:, None, 3:6] + k * delta * ref[:, None, 6:9]
    xdes[:, :, 8] = ref[:, None, 9]
    xdes[:, :, 9:12] = ref[:, None, 6:9]
    xdes[:, :, 12] = x0[:, None, 12]
    tis, ss, ds = (np.asarray(g["gait"])[:, i][:, None] for i in range(3))
    tau = tis + np.arange(N)[None, :]
    st = np.minimum(tau // (ss + ds), S - 1)
    tau = tau - st * (ss + ds)
    fid = np.asarray(g["feet_id"])[np.arange(B)[:, None], st]                     # [B,N,4]
    contact = np.where((tau < ss)[:, :, None], fid, 1).astype(np.uint8)
    fh = np.asarray(g["footholds"], float)[np.arange(B)[:, None], st]             # [B,N,4,3]
    r = fh - xdes[:, :N, None, 3:6]
    r[:, 0] = np.asarray(g["feet0"], float) - x0[:, None, 3:6]
    return {"x0": x0, "r": r, "contact": contact, "xdes": xdes, "mu": np.asarray(g["mu"], float)}



# ----------------------------------------------------------------------------------------------------------------------
# Closed-loop roll-out batches (include/mpcqp.h, mpcqp_rollout): per-robot plan tables from the footstep planner
# ----------------------------------------------------------------------------------------------------------------------
def make_rollout_batch(B, N=10, delta=0.03, seed=20250813, gait_names=("trot", "gallop", "amble"), mus=(0.5, 0.7, 1.0), ss=4, ds=2,
                       total_steps=50, v_ref=(0.18, 0.0, 0.0)):
    """B robots at the start of a walk: state, reference descriptor, plan table (host planner, one plan per robot), gait clock.
    BASELINE config-1 timing by default (the reference's step durations in seconds: 4 / 2 ticks of 0.03 s).  Heading 0 for every
    robot: the reference integrates v_com_ref in world axes (src/mpc.py:202-214) while its planner walks along the heading."""
    from .footstep_planner import LEGS, FootstepPlanner
    rng = np.random.default_rng(seed)
    x = np.zeros((B, 13)); ref = np.zeros((B, 10))
    pos = np.zeros((B, total_steps, 4, 3)); fid = np.ones((B, total_steps, 4), np.uint8); meta = np.zeros((B, 4), np.int32)
    gid = rng.integers(0, len(gait_names), B)
    mu = np.asarray(mus, float)[rng.integers(0, len(mus), B)]
    for b in range(B):
        com = np.array([0.0, 0.0, H_COM]) + rng.normal(0.0, 0.005, 3) * np.array([1.0, 1.0, 0.5])
        feet = NOMINAL_FEET + np.array([com[0], com[1], H_COM]) + np.concatenate([rng.normal(0.0, 0.003, (4, 2)), np.zeros((4, 1))], axis=1)
        params = {"g": G_ACC, "h": H_COM, "step_height": 0.08, "ss_duration": ss, "ds_duration": ds, "world_time_step": delta,
                  "total_steps": total_steps, "first_swing": np.array(GAITS[gait_names[gid[b]]]), "µ": float(mu[b]), "N": N, "dof": 18,
                  "v_com_ref": np.asarray(v_ref, float), "theta_dot": 0.0, "log_samples": 0}
        initial = {l: feet[k].copy() for k, l in enumerate(LEGS)}
        initial.update(roll=0.0, pitch=0.0, yaw=0.0, com_position=com.copy())
        pl = FootstepPlanner(initial, params, show=False)
        pos[b] = pl.pos[:total_steps]; fid[b] = pl.feet_id[:total_steps]; meta[b] = (total_steps, ss, ds, 0)
        x[b, 3:6] = com; x[b, 6:9] = rng.normal(0.0, 0.02, 3); x[b, 9:12] = rng.normal(0.0, 0.02, 3); x[b, 12] = G_ACC
        ref[b] = [0.0, 0.0, 0.0, com[0], com[1], H_COM, v_ref[0], v_ref[1], v_ref[2], 0.0]
    return {"x": x, "ref": ref, "plan_pos": pos, "plan_feet_id": fid, "plan_meta": meta, "tick": np.zeros(B, np.int32), "mu": mu,
            "gait_ids": gid}
