"""Config 1 of BASELINE.json (plumbing, no GPU): a single Lite3 trotting, horizon 10, dt 0.03, 300 ticks of the whole
caller chain customPreStep -> ground_controller -> MPC.solve(t, logger) -> update_r_num / planner / logger hooks, with the
DART world replaced by the kinematic single-rigid-body stand-in.  The engine has no CPU path, so here the solve hook of
`MPC` is overridden IN THE TEST to call the CPU checker (tests may do that; the product never does)."""
import numpy as np

import mpcqp
from conftest import ORACLE_SO
from mpcqp import lite3_model
from mpcqp.controller import Lite3Controller
from mpcqp.mpc import MPC


class OracleMPC(MPC):
    def _make_solver(self, device, precision, engine_overrides):
        lib = mpcqp.Library(ORACLE_SO)
        return mpcqp.Engine(lib, lib.default_config(N=self.N, delta=self.delta, max_iter=4000, **engine_overrides))

    def _solve_one(self, x0, r, contact, xdes):
        out = self._solver.solve_batch_host(x0[None], r[None], contact[None], xdes[None], np.array([float(self.mu)]))
        return out["u"][0], out["X"][0], int(out["status"][0])


def test_leg_jacobian_matches_finite_differences():
    rng = np.random.default_rng(0)
    for leg in range(4):
        q = np.array([0.1, -1.0, 1.6]) + rng.normal(0, 0.2, 3)
        p, J = lite3_model.leg_fk_jac(leg, q)
        Jn = np.stack([(lite3_model.leg_fk_jac(leg, q + 1e-6 * np.eye(3)[i])[0] - p) / 1e-6 for i in range(3)], axis=1)
        assert np.abs(J - Jn).max() < 1e-5
        assert np.abs(lite3_model.leg_fk_jac(leg, lite3_model.leg_ik(leg, p, q0=q + 0.05))[0] - p).max() < 1e-9
    # the reference's spawn pose (HipY -60 deg, Knee 90 deg, src/main.py:65-68) puts the feet under the hips
    p, _ = lite3_model.leg_fk_jac(0, np.radians([0.0, -60.0, 90.0]))
    assert abs(p[1] - (0.062 + 0.0985)) < 1e-12 and -0.32 < p[2] < -0.2


def test_config1_closed_loop_plumbing(oracle_lib):
    ctl = Lite3Controller(OracleMPC)
    T = 300
    contact, xs = [], []
    for t in range(T):
        contact.append(ctl.footstep_planner.get_phase_at_time(t))     # before the tick: the swing query below mutates feet_id
        tau = ctl.customPreStep()
        assert set(tau) == set(mpcqp.footstep_planner.LEGS) and all(np.all(np.isfinite(v)) for v in tau.values())
        assert ctl.mpc.status in (1, 2)
        xs.append(ctl.lite3.x.copy())
    xs = np.array(xs)
    # tracking over ALL 300 ticks (the reference's acceptance is visual, src/plot.py; these are the same quantities):
    # height within 2 cm, mean forward speed within 0.05 m/s of v_ref = 0.18 while the plan walks, pitch / roll small
    assert np.abs(xs[:, 5] - 0.285).max() < 0.02
    assert abs(xs[:294, 9].mean() - 0.18) < 0.05
    assert np.abs(xs[:, 1]).max() < 0.1 and np.abs(xs[:, 0]).max() < 0.1 and np.abs(xs[:, 4]).max() < 0.05
    log = ctl.logger.log
    assert len(log["time array"]) == T and len(log["TRACKING PERFORMANCE"]["actual"]) == T
    assert all(len(log["FORCES"][l]["z"]) == T for l in log["FORCES"])
    assert all(len(log["FEET POS"][l]["des"]) == T for l in log["FEET POS"])
    assert [p["time step"] for p in log["MPC PREDICTIONS"]] == [0, 80] and log["mpc_freq"] > 0
    fz = np.array([log["FORCES"][l]["z"] for l in mpcqp.footstep_planner.LEGS])        # [4,T]
    # trot: diagonal pairs alternate; a swing leg carries exactly zero force, stance legs respect 3 <= fz <= 100
    contact = np.array(contact).T
    assert np.all(fz[contact == 0] == 0) and np.all(fz[contact == 1] >= 3 - 1e-6) and np.all(fz <= 100 + 1e-6)
    # the body is carried: mean total vertical force = weight
    assert abs(fz.sum(axis=0).mean() - 8.885 * 9.81) < 0.05 * 8.885 * 9.81
    # reference roll-forward of the targets (src/mpc.py:261-262)
    # ... which stops on the last plan step (src/mpc.py:181-183): 49 moving steps of 6 ticks
    assert abs(ctl.mpc.com_pos_start[0] - 49 * 6 * 0.03 * 0.18) < 1e-9
    # log dump / reload without pickle
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "log.npz")
        ctl.logger.save_log(f)
        back = ctl.logger.load_log(f)
        assert back["FORCES/FL_FOOT/z"].shape == (T,) and "MPC PREDICTIONS/1/predicted_state" in back


def test_log_file_in_the_reference_format(oracle_lib, tmp_path):
    """the run log is written as the reference writes it (src/logger.py:64-66: pickle of the nested
    dict) so that `Logger.load_log` / plot.py-style consumers (src/logger.py:69-72, src/plot.py:12-83) read it unchanged.  The
    schema is compared key by key with the reference's own committed log (tests/golden/ref_log.npz holds its arrays)."""
    import pickle
    ctl = Lite3Controller(OracleMPC)
    T = 81
    for _ in range(T):
        ctl.customPreStep()
    f = str(tmp_path / "simulation_log.pkl")
    ctl.logger.save_log(f)
    with open(f, "rb") as fh:
        raw = pickle.load(fh)                                        # what the reference's load_log does (our own file)
    assert set(raw) == {"mpc_freq", "sim_params", "total_sim_steps", "time array", "FEET POS", "MPC PREDICTIONS", "TRACKING PERFORMANCE",
                        "FORCES", "CONTROL EFFORT"}                 # src/logger.py:22-46
    legs = ["FL_FOOT", "FR_FOOT", "HL_FOOT", "HR_FOOT"]
    assert list(raw["FEET POS"]) == legs and list(raw["FORCES"]) == legs and list(raw["CONTROL EFFORT"]) == legs
    assert list(raw["FORCES"]["FL_FOOT"]) == ["x", "y", "z"] and list(raw["CONTROL EFFORT"]["HR_FOOT"]) == ["HR_HipX", "HR_HipY", "HR_Knee"]
    assert len(raw["time array"]) == T and len(raw["TRACKING PERFORMANCE"]["actual"]) == T and len(raw["FORCES"]["HL_FOOT"]["z"]) == T
    assert len(raw["TRACKING PERFORMANCE"]["actual"][0]) == 12 and np.asarray(raw["TRACKING PERFORMANCE"]["desired"][0]).shape == (12,)
    assert [p["time step"] for p in raw["MPC PREDICTIONS"]] == [0, 80]
    p0 = raw["MPC PREDICTIONS"][0]
    assert set(p0) == {"time step", "predicted_state", "desired_state", "predicted forces"}
    assert p0["predicted_state"].shape == (12, 11) and p0["desired_state"].shape == (12, 11) and p0["predicted forces"].shape == (4, 10)
    assert raw["sim_params"]["N"] == 10 and raw["sim_params"]["µ"] == 1
    # a plot.py-style consumer: load through the Logger and slice the way src/plot.py / src/utils.py:131-209 do
    from mpcqp.logger import Logger
    lg = Logger({"params": {}, "total_sim_steps": 0})
    log = lg.load_log(f)
    z = np.array(log["TRACKING PERFORMANCE"]["actual"])[:, 5]
    fz = np.array([log["FORCES"][l]["z"] for l in legs])
    assert z.shape == (T,) and fz.shape == (4, T) and abs(z.mean() - 0.285) < 0.01
    # the restricted loader refuses anything but arrays
    bad = str(tmp_path / "bad.pkl")
    with open(bad, "wb") as fh:
        pickle.dump({"x": np.linalg.norm}, fh)
    import pytest
    with pytest.raises(pickle.UnpicklingError):
        lg.load_log(bad)
    # one robot of a device roll-out in the same layout
    act, des, frc = np.zeros((5, 12)), np.ones((5, 12)), np.arange(60.0).reshape(5, 12)
    lr = Logger.from_rollout({"N": 10}, act, des, frc)
    assert lr.log["FORCES"]["FR_FOOT"]["z"] == [5.0, 17.0, 29.0, 41.0, 53.0] and len(lr.log["TRACKING PERFORMANCE"]["actual"]) == 5
