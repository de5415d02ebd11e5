"""The reference's caller-facing surface (src/mpc.py:25,176,303,306; caller at src/main.py:193-219) on the HIP engine:
a single-robot receding-horizon run (no DART: a kinematic single-rigid-body stand-in advances the state with the
predicted X[:,1]) must exercise MPC.solve(t, logger), the logger hooks, update_r_num and the per-tick reference
roll-forward, and every tick's forces must match the CPU oracle on the same inputs."""
import numpy as np
import pytest

import mpcqp
from conftest import rel_err
from mpcqp.footstep_planner import LEGS, FootstepPlanner
from mpcqp.logger import Logger
from mpcqp.mpc import MPC, MPCFleet, MPCProblemBuilder

pytestmark = pytest.mark.gpu


class KinematicLite3:
    """Stand-in for the controller's `retrieve_state()` (src/main.py:286-350): stance feet stay on the plan."""

    def __init__(self, planner, x0):
        self.planner, self.x, self.t = planner, x0.copy(), 0

    def retrieve_state(self):
        st = self.planner.get_step_index_at_time(self.t)
        s = {l: {"pos": np.concatenate([np.zeros(3), self.planner.pos[st, k]])} for k, l in enumerate(LEGS)}
        s["TORSO"] = {"pos": self.x[0:3].copy(), "vel": self.x[6:9].copy()}
        s["com"] = {"pos": self.x[3:6].copy(), "vel": self.x[9:12].copy()}
        return s


def _setup(N=10, dt=0.03, first_swing=(1, 0, 0, 1)):
    # config 1: the reference's step durations in seconds (0.10 / 0.05 s, src/main.py:35-36,41) = 4 / 2 ticks of 0.03 s
    params = {"g": -9.81, "h": 0.285, "step_height": 0.08, "ss_duration": 4, "ds_duration": 2, "world_time_step": dt,
              "total_steps": 50, "first_swing": np.array(first_swing), "µ": 1, "N": N, "dof": 18,
              "v_com_ref": np.array([0.18, 0.0, 0.0]), "theta_dot": 0.0, "log_samples": 1000}
    feet = mpcqp.synth.NOMINAL_FEET + np.array([0.0, 0.0, 0.285])
    initial = {l: feet[k].copy() for k, l in enumerate(LEGS)}
    initial.update(roll=0.0, pitch=0.0, yaw=0.0, com_position=np.array([0.0, 0.0, 0.285]))
    planner = FootstepPlanner(initial, params, show=False)
    x0 = np.array([0, 0, 0, 0, 0, 0.285, 0, 0, 0, 0, 0, 0], float)
    return params, initial, planner, x0


def test_single_robot_receding_horizon_surface(oracle_solve):
    params, initial, planner, x0 = _setup()
    robot = KinematicLite3(planner, x0)
    mpc = MPC(lite3=robot, initial=initial, footstep_planner=planner, params=params)
    logger = Logger({"params": params, "total_sim_steps": 100})
    inputs, xs = [], []
    for t in range(100):
        robot.t = t
        b = mpc._builder
        snap = b.build(t, robot.retrieve_state())[:4]          # what this tick's solve will be fed
        forces = mpc.solve(t, logger)
        inputs.append(snap)
        assert set(forces) == set(LEGS) and all(f.shape == (3,) for f in forces.values())
        assert mpc.x.shape == (13, 1) and mpc.x_log.shape == (12, 11) and mpc.x_plot.shape == (3, 11)
        assert mpc.u.shape == (12,) and mpc.u_plot.shape == (12, 10) and mpc.status in (1, 2)
        robot.x = mpc.x_log[:, 1].copy()                       # apply the first predicted step
        xs.append(robot.x.copy())
    xs = np.array(xs)
    assert len(logger.log["TRACKING PERFORMANCE"]["actual"]) == 100
    assert [p["time step"] for p in logger.log["MPC PREDICTIONS"]] == [0, 80]
    assert logger.log["MPC PREDICTIONS"][0]["predicted forces"].shape == (4, 10)
    # reference roll-forward (src/mpc.py:261-262): 100 ticks at v = 0.18 m/s, dt = 0.03 s
    assert abs(mpc.com_pos_start[0] - 100 * 0.03 * 0.18) < 1e-9 and initial["com_position"] is mpc.com_pos_start
    # tracking on every tick: height within 2 cm, mean forward speed within 0.05 m/s of v_ref, pitch / roll small
    assert np.abs(xs[:, 5] - 0.285).max() < 0.02 and abs(xs[:, 9].mean() - 0.18) < 0.05
    assert np.abs(xs[:, 1]).max() < 0.1 and np.abs(xs[:, 0]).max() < 0.1
    # parity of a sample of ticks with the oracle on identical inputs
    idx = list(range(0, 100, 9))
    batch = {"x0": np.stack([inputs[i][0] for i in idx]), "r": np.stack([inputs[i][1] for i in idx]),
             "contact": np.stack([inputs[i][2] for i in idx]), "xdes": np.stack([inputs[i][3] for i in idx]),
             "mu": np.ones(len(idx))}
    ref = oracle_solve(batch)
    sol = mpcqp.MPCBatch(io_dtype="f64")
    import torch
    dev = sol.upload(batch)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    assert rel_err(out["u"].cpu().numpy(), ref["u"]).max() <= 1e-4


def test_reference_parameters_closed_loop():
    """The reference's own `params` (src/main.py:31-46: N = 60, world_time_step = 0.01, ss / ds = 10 / 5 ticks, first_swing
    [0, 0, 1, 1], v_com_ref 0.18) through MPC.solve(t, logger) on the device -- the stage-wise engine -- for 120 ticks of the kinematic
    stand-in: same surface, logger hooks at t = 0 and 80 with the reference's 12 x 61 / 4 x 60 shapes, tracking on every tick."""
    params = {"g": -9.81, "h": 0.285, "step_height": 0.08, "ss_duration": 10, "ds_duration": 5, "world_time_step": 0.01,
              "total_steps": 20, "first_swing": np.array([0, 0, 1, 1]), "µ": 1, "N": 60, "dof": 18,
              "v_com_ref": np.array([0.18, 0.0, 0.0]), "theta_dot": 0.0, "log_samples": 1000}
    feet = mpcqp.synth.NOMINAL_FEET + np.array([0.0, 0.0, 0.285])
    initial = {l: feet[k].copy() for k, l in enumerate(LEGS)}
    initial.update(roll=0.0, pitch=0.0, yaw=0.0, com_position=np.array([0.0, 0.0, 0.285]))
    planner = FootstepPlanner(initial, params, show=False)
    robot = KinematicLite3(planner, np.array([0, 0, 0, 0, 0, 0.285, 0, 0, 0, 0, 0, 0], float))
    mpc = MPC(lite3=robot, initial=initial, footstep_planner=planner, params=params)
    logger = Logger({"params": params, "total_sim_steps": 120})
    xs = []
    for t in range(120):
        robot.t = t
        forces = mpc.solve(t, logger)
        assert mpc.status == 1 and mpc.x_log.shape == (12, 61) and mpc.u_plot.shape == (12, 60)
        phase = planner.get_phase_at_time(t)
        for k, l in enumerate(LEGS):                              # swing legs carry no force, stance legs at least f_min
            assert (forces[l][2] >= 3.0 - 1e-6) if phase[k] else np.all(forces[l] == 0.0)
        robot.x = mpc.x_log[:, 1].copy()
        xs.append(robot.x.copy())
    xs = np.array(xs)
    assert [p["time step"] for p in logger.log["MPC PREDICTIONS"]] == [0, 80]
    assert logger.log["MPC PREDICTIONS"][1]["predicted_state"].shape == (12, 61)
    assert logger.log["MPC PREDICTIONS"][1]["predicted forces"].shape == (4, 60)
    assert np.abs(xs[:, 5] - 0.285).max() < 0.02 and abs(xs[40:, 9].mean() - 0.18) < 0.05
    assert np.abs(xs[:, 1]).max() < 0.1 and np.abs(xs[:, 0]).max() < 0.1


def test_warm_started_controller_matches_cold_one():
    """MPC(..., warm_start=True) -- the reference's `opt.set_initial(U, sol.value(U))` (src/mpc.py:270-271), with the engine
    carrying the solution and its multipliers from tick to tick -- must command the same forces as the cold controller on
    the same receding-horizon run, with fewer ADMM iterations."""
    runs = {}
    for warm in (False, True):
        params, initial, planner, x0 = _setup()
        robot = KinematicLite3(planner, x0)
        mpc = MPC(lite3=robot, initial=initial, footstep_planner=planner, params=params, warm_start=warm)
        logger = Logger({"params": params, "total_sim_steps": 60})
        F, its = [], []
        for t in range(60):
            robot.t = t
            forces = mpc.solve(t, logger)
            assert mpc.status in (1, 2)
            F.append(np.concatenate([forces[l] for l in LEGS]))
            its.append(int(mpc._solver._out[(1, True)]["iters"][0].item()) % 1000)
            robot.x = mpc.x_log[:, 1].copy()
        runs[warm] = (np.array(F), np.array(its))
    Fc, ic = runs[False]
    Fw, iw = runs[True]
    assert np.abs(Fw - Fc).max() <= 1e-4 * max(1.0, np.abs(Fc).max())     # same commands (states stay in lockstep)
    assert iw[1:].mean() < 0.8 * ic[1:].mean(), (iw.mean(), ic.mean())


def test_fleet_one_call_per_tick(oracle_solve):
    fleet_b, states = [], []
    for k, fs in enumerate(((1, 0, 0, 1), (0, 0, 1, 1), (1, 0, 1, 0), (0, 0, 0, 0))):
        params, initial, planner, x0 = _setup(first_swing=fs)
        fleet_b.append(MPCProblemBuilder(initial, planner, params))
        r = KinematicLite3(planner, x0); r.t = 20
        states.append(r.retrieve_state())
    fleet = MPCFleet(fleet_b)
    out = fleet.solve(20, states)
    ref = oracle_solve(out["inputs"])
    assert np.all((out["status"] == 1) | (out["status"] == 2))
    assert rel_err(out["u"], ref["u"]).max() <= 1e-4
