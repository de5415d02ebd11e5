"""Closed-loop roll-out (include/mpcqp.h, mpcqp_rollout).

CPU: the checker's literal C loop reproduces, tick for tick, the Python caller chain MPC.solve(t, logger) over the kinematic
stand-in (the parameter fill validated against the reference's own planner objects in tests/test_planner_glue.py).
GPU: the device roll-out (3 launches per tick, no host round trips) matches the checker on the same robots."""
import numpy as np
import pytest

import mpcqp
from conftest import ORACLE_SO
from mpcqp.footstep_planner import LEGS, FootstepPlanner
from mpcqp.logger import Logger
from test_controller_plumbing import OracleMPC


def _oracle_engine(**kw):
    lib = mpcqp.Library(ORACLE_SO)
    return mpcqp.Engine(lib, lib.default_config(N=10, delta=0.03, max_iter=4000, **kw))


class _StandIn:
    """retrieve_state() provider: stance and swing feet alike stand on the plan (swing feet carry no force)."""

    def __init__(self, planner, x):
        self.planner, self.x, self.t = planner, x.copy(), 0

    def retrieve_state(self):
        st = self.planner.get_step_index_at_time(self.t)
        s = {l: {"pos": np.concatenate([np.zeros(3), self.planner.pos[st, k]])} for k, l in enumerate(LEGS)}
        s["TORSO"] = {"pos": self.x[0:3].copy(), "vel": self.x[6:9].copy()}
        s["com"] = {"pos": self.x[3:6].copy(), "vel": self.x[9:12].copy()}
        return s


def test_oracle_rollout_equals_python_caller_chain(oracle_lib):
    T = 40
    rb = mpcqp.synth.make_rollout_batch(3, total_steps=5, seed=11)          # 5 steps x 6 ticks: the walk ends inside the roll-out
    out = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], rb["plan_meta"], rb["tick"], rb["mu"], T)
    assert np.all(out["solved"] == T) and np.all(out["tick"] == T)
    for b in range(3):
        params = {"g": -9.81, "h": 0.285, "step_height": 0.08, "ss_duration": 4, "ds_duration": 2, "world_time_step": 0.03, "total_steps": 5,
                  "first_swing": np.array(mpcqp.synth.GAITS[("trot", "gallop", "amble")[rb["gait_ids"][b]]]), "µ": float(rb["mu"][b]), "N": 10,
                  "dof": 18, "v_com_ref": np.array([0.18, 0.0, 0.0]), "theta_dot": 0.0, "log_samples": T}
        # the same initial configuration the batch generator gave the planner
        initial = {l: rb["plan_pos"][b, 0, k].copy() for k, l in enumerate(LEGS)}
        initial.update(roll=0.0, pitch=0.0, yaw=0.0, com_position=rb["x"][b, 3:6].copy())
        planner = FootstepPlanner(initial, params, show=False)
        assert np.abs(planner.pos[:5] - rb["plan_pos"][b]).max() <= 1e-12
        robot = _StandIn(planner, rb["x"][b, :12])
        mpc = OracleMPC(lite3=robot, initial=initial, footstep_planner=planner, params=params)
        logger = Logger({"params": params, "total_sim_steps": T})
        for t in range(T):
            robot.t = t
            f = mpc.solve(t, logger)
            assert np.abs(np.concatenate([f[l] for l in LEGS]) - out["forces"][b, t]).max() <= 1e-6 * max(1.0, np.abs(out["forces"][b, t]).max())
            assert np.abs(robot.x - out["actual"][b, t]).max() <= 1e-9
            robot.x = mpc.x_log[:, 1].copy()
        assert np.abs(np.array(logger.log["TRACKING PERFORMANCE"]["desired"]) - out["desired"][b]).max() <= 1e-12
        assert np.abs(robot.x - out["x"][b, :12]).max() <= 1e-8


def test_oracle_rollout_tracks_the_reference(oracle_lib):
    """Config-1 timing, three gaits: the closed loop tracks (the quantities the reference inspects by eye, src/plot.py)."""
    T = 120
    rb = mpcqp.synth.make_rollout_batch(6, seed=5)
    out = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], rb["plan_meta"], rb["tick"], rb["mu"], T)
    a, dsr = out["actual"], out["desired"]
    assert np.all(out["solved"] == T)
    assert np.abs(a[:, :, 5] - 0.285).max() < 0.03                                   # height (trot alone: 4 mm; gallop / amble at mu 0.5: 2.4 cm)
    assert np.abs(a[:, 20:, 9].mean(axis=1) - 0.18).max() < 0.05                     # mean forward speed
    assert np.abs(a[:, :, 0:2]).max() < 0.1                                          # roll, pitch
    assert np.abs(a[:, :, 3] - dsr[:, :, 3]).max() < 0.05                            # com x follows the rolled-forward reference


def _malformed(rb):
    """Plan rows no planner would write: zero steps, zero-length steps, a step count beyond the table, negative durations and ticks."""
    meta, tick = rb["plan_meta"].copy(), rb["tick"].copy()
    meta[0] = (0, 4, 2, 0); meta[1] = (5, 0, 0, 0); meta[2] = (99, 4, 2, 0); meta[3] = (5, -3, -1, 0)
    tick[4] = -7
    return meta, tick


def test_oracle_rollout_clamps_malformed_plan_rows(oracle_lib):
    """include/mpcqp.h: the plan table is clamped (1 <= S_b <= S, ss >= 0, ss + ds >= 1, tick >= 0), never indexed with or divided by as
    it stands -- a malformed row gives a finite, bounded roll-out, not a fault (round-2 advisor)."""
    rb = mpcqp.synth.make_rollout_batch(6, total_steps=5, seed=3)
    meta, tick = _malformed(rb)
    out = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], meta, tick, rb["mu"], 8)
    assert np.all(np.isfinite(out["forces"])) and np.all(np.isfinite(out["x"])) and np.all(out["tick"] == tick + 8)
    assert np.all(out["solved"] == 8)
    # S_b = 0 is served as S_b = 1 (the first step, reference velocities gated off), S_b = 99 as the table's 5 steps
    good = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], rb["plan_meta"], rb["tick"], rb["mu"], 8)
    assert np.array_equal(out["forces"][2], good["forces"][2]) and np.array_equal(out["forces"][5], good["forces"][5])


@pytest.mark.gpu
def test_device_rollout_clamps_malformed_plan_rows():
    import torch
    rb = mpcqp.synth.make_rollout_batch(6, total_steps=5, seed=3)
    meta, tick = _malformed(rb)
    ref = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], meta, tick, rb["mu"], 8)
    sol = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f64", precision="mixed")
    f = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda").contiguous()
    x, rf, tk = f(rb["x"]), f(rb["ref"]), f(tick, torch.int32)
    out = sol.rollout(x, rf, f(rb["plan_pos"]), f(rb["plan_feet_id"], torch.uint8), f(meta, torch.int32), tk, f(rb["mu"]), 8)
    torch.cuda.synchronize()
    assert np.array_equal(tk.cpu().numpy(), tick + 8) and np.all(out["solved"].cpu().numpy() == 8)
    F = out["forces"].cpu().numpy()
    assert np.all(np.isfinite(F)) and np.abs(F - ref["forces"]).max() <= 1e-4 * np.abs(ref["forces"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("warm", [False, True])
def test_device_rollout_matches_oracle(warm):
    import torch
    T, B = 30, 24
    rb = mpcqp.synth.make_rollout_batch(B, seed=7)
    ref = _oracle_engine().rollout_host(rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], rb["plan_meta"], rb["tick"], rb["mu"], T)
    sol = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f64", precision="mixed", warm_start=warm, warm_shift=warm)
    f = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda().contiguous()
    x, rf = f(rb["x"], torch.float64), f(rb["ref"], torch.float64)
    pos, fid = f(rb["plan_pos"], torch.float64), f(rb["plan_feet_id"], torch.uint8)
    meta, tick, mu = f(rb["plan_meta"], torch.int32), f(rb["tick"], torch.int32), f(rb["mu"], torch.float64)
    out = sol.rollout(x, rf, pos, fid, meta, tick, mu, T)
    torch.cuda.synchronize()
    assert np.all(out["solved"].cpu().numpy() == T) and np.all(tick.cpu().numpy() == T)
    sc = max(1.0, np.abs(ref["forces"]).max())
    assert np.abs(out["forces"].cpu().numpy() - ref["forces"]).max() <= 1e-4 * sc      # same closed loop, tick for tick
    assert np.abs(out["actual"].cpu().numpy() - ref["actual"]).max() <= 1e-5
    assert np.abs(out["desired"].cpu().numpy() - ref["desired"]).max() <= 1e-12
    assert np.abs(x.cpu().numpy() - ref["x"]).max() <= 1e-5 and np.abs(rf.cpu().numpy() - ref["ref"]).max() <= 1e-12


@pytest.mark.gpu
def test_device_rollout_large_batch_tracks():
    """B = 2048 robots x 60 ticks, f32 buffers, warm-started: every tick solved, tracking within the config-1 bounds."""
    import torch
    T, B0 = 60, 32
    rb = mpcqp.synth.make_rollout_batch(B0, seed=9)
    rep = 64                                                       # 2048 robots: 64 copies of 32 plans
    tile = lambda a: np.concatenate([a] * rep, axis=0)
    sol = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f32", precision="mixed", warm_start=True, warm_shift=True)
    f = lambda a, dt: torch.as_tensor(np.ascontiguousarray(tile(a)), dtype=dt).cuda().contiguous()
    x, rf = f(rb["x"], torch.float32), f(rb["ref"], torch.float32)
    pos, fid = f(rb["plan_pos"], torch.float32), f(rb["plan_feet_id"], torch.uint8)
    meta, tick, mu = f(rb["plan_meta"], torch.int32), f(rb["tick"], torch.int32), f(rb["mu"], torch.float32)
    out = sol.rollout(x, rf, pos, fid, meta, tick, mu, T)
    torch.cuda.synchronize()
    a = out["actual"].cpu().numpy()
    assert (out["solved"].cpu().numpy() == T).mean() >= 0.999
    assert np.abs(a[:, :, 5] - 0.285).max() < 0.03 and np.abs(a[:, :, 0:2]).max() < 0.1
    assert np.abs(a[:, 20:, 9].mean(axis=1) - 0.18).max() < 0.05
    assert np.array_equal(a[:B0], a[B0:2 * B0])                    # identical robots, identical trajectories (deterministic)
