"""The reference's OWN configuration on the device: N = 60, delta = 0.01 (src/main.py:37,41; src/mpc.py:30-31), served by the
stage-wise engine (csrc/mpcqp_stage.h).

* The stage-wise engine against the oracle and against the dense wrench-space engine at N = 10 / 20 (MPCQP_FLAG_STAGE_KERNEL).
* The ten golden ticks of the logged run at N = 60 against the committed oracle optima (tests/golden/qp_optima.npz): forces and
  states within 1e-4 (alpha = 1e-2, 1e-4); the reference's cost itself (alpha = 0): objective 1e-6, states 1e-4.
* ALL 1000 logged ticks through the HIP engine, with the checks tests/test_oracle_pinning.py applies to the log and the oracle:
  the engine's stage-0 support pattern equals the log's on every tick, forces obey the box and the friction pyramid, and at the two
  ticks whose predictions the reference kept (t = 0, 80) the engine's objective is not worse than what OSQP returned.  This is
  the HIP path next to numbers the reference itself produced (src/simulation_log.pkl, exported to tests/golden/ref_log.npz).
Tolerances are stated at each assertion.
"""
import numpy as np
import pytest
import torch

import mpcqp
import qp_spec as S
from conftest import rel_err
from mpcqp.footstep_planner import LEGS, FootstepPlanner
from mpcqp.mpc import MPCProblemBuilder
from test_planner_glue import _initial, _params

pytestmark = pytest.mark.gpu


def gpu_solve(batch, N, delta, precision="mixed", flags=mpcqp.FLAG_POLISH, **kw):
    sol = mpcqp.MPCBatch(N=N, delta=delta, io_dtype="f64", precision=precision, flags=flags, **kw)
    dev = sol.upload(batch)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
    res["ms"] = sol.last_kernel_ms()
    return res


def logged_run_inputs(golden, N, ticks):
    """The operator tuple of MPC.solve (src/mpc.py:176-254) at the given ticks of the logged run, built by the package's own host
    glue from the log (measured state, rolled-forward references, measured feet) -- the construction tests/test_planner_glue.py pins
    against a replay with the reference's planner objects at the golden ticks."""
    L = golden["ref_log"]
    x0s, rs, cs, xds = [], [], [], []
    for t in ticks:
        params = _params(L, N=N)
        ini = _initial(L)
        pl = FootstepPlanner(ini, params, show=False)            # fresh per tick: the swing generator mutates the plan (ftg.py:53-54)
        b = MPCProblemBuilder(ini, pl, params)
        b.yaw_start = float(L["desired"][t, 2])
        b.com_pos_start = L["desired"][t, 3:6].copy()
        state = {l: {"pos": np.concatenate([np.zeros(3), L["feet_actual"][t, k]])} for k, l in enumerate(LEGS)}
        state["TORSO"] = {"pos": L["actual"][t, 0:3], "vel": L["actual"][t, 6:9]}
        state["com"] = {"pos": L["actual"][t, 3:6], "vel": L["actual"][t, 9:12]}
        x0, r, contact, xdes, _, _ = b.build(int(t), state)
        x0s.append(x0); rs.append(r); cs.append(contact); xds.append(xdes)
    return {"x0": np.array(x0s), "r": np.array(rs), "contact": np.array(cs, dtype=np.uint8), "xdes": np.array(xds),
            "mu": np.full(len(ticks), float(L["param_mu"]))}


@pytest.mark.parametrize("N,precision", [(10, "f64"), (10, "mixed"), (20, "mixed")])
def test_stage_engine_agrees_with_oracle_and_dense_engine(oracle_solve, N, precision):
    b = mpcqp.synth.config3(96) if N == 10 else mpcqp.synth.config5(48)
    ref = oracle_solve(b, N=N)
    st = gpu_solve(b, N, 0.03, precision, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL)
    de = gpu_solve(b, N, 0.03, precision)
    sok = st["status"] == 1
    assert sok.mean() >= (1.0 if precision == "f64" else 0.97), sok.mean()
    assert rel_err(st["u"], ref["u"])[sok].max() <= 1e-4 and np.abs(st["X"] - ref["X"])[sok].max() <= 1e-4      # (measured 2e-10 / 2e-9)
    ok = sok & (de["status"] == 1)
    assert rel_err(st["u"][ok], de["u"][ok]).max() <= 1e-4                                            # two engines, one optimum


@pytest.mark.parametrize("precision", ["mixed", "f64"])
def test_reference_horizon_golden_ticks(golden, precision):
    q, opt = golden["qp_inputs"], golden["qp_optima"]
    N = 60
    b = {"x0": q[f"N{N}_x0"], "r": q[f"N{N}_r"], "contact": q[f"N{N}_contact"], "xdes": q[f"N{N}_xdes"],
         "mu": np.full(len(q["ticks"]), float(q["mu"]))}
    for alpha, tag in ((1e-2, "a1e-2"), (1e-4, "a1e-4")):
        out = gpu_solve(b, N, float(q["delta"]), precision, alpha=alpha)
        assert np.all(out["status"] == 1), out["status"]
        assert rel_err(out["u"], opt[f"N{N}_{tag}_u"]).max() <= 1e-4                     # (measured 1e-10 / 1e-8)
        assert np.abs(out["X"] - opt[f"N{N}_{tag}_X"]).max() <= 1e-4
    # the reference's cost itself (alpha = 0.0, src/mpc.py:121): forces are not unique, objective and states are
    out = gpu_solve(b, N, float(q["delta"]), precision, alpha=0.0)
    cfg = S.QPConfig(N=N, delta=float(q["delta"]), alpha=0.0)
    ok = out["status"] == 1
    assert ok.sum() >= len(ok) - 1
    for i in np.where(ok)[0]:
        J = S.objective(out["X"][i], out["u"][i], b["xdes"][i], cfg)
        assert abs(J - opt["N60_a0_J"][i]) <= 1e-6 * max(1.0, abs(opt["N60_a0_J"][i])), (i, J, opt["N60_a0_J"][i])
        assert np.abs(out["X"][i] - opt["N60_a0_X"][i]).max() <= 1e-4


def test_all_logged_ticks_through_the_engine(golden):
    L, G = golden["ref_log"], golden["planner_golden"]
    N, d = 60, float(L["param_world_time_step"])
    ticks = np.arange(1000)
    b = logged_run_inputs(golden, N, ticks)
    # the host glue reproduces the golden inputs where the reference replay recorded them
    q = golden["qp_inputs"]
    for j, t in enumerate(q["ticks"]):
        assert np.array_equal(b["x0"][t], q["N60_x0"][j]) and np.array_equal(b["contact"][t], q["N60_contact"][j])
        assert np.abs(b["r"][t] - q["N60_r"][j]).max() <= 1e-12 and np.abs(b["xdes"][t] - q["N60_xdes"][j]).max() <= 1e-12
    out = gpu_solve(b, N, d, "mixed", alpha=1e-2)
    assert np.mean(out["status"] == 1) >= 0.999, np.bincount(out["status"] + 1)
    F = out["u"][:, 0].reshape(1000, 4, 3)                                  # the stage-0 forces MPC.solve returns (src/mpc.py:273-278)
    Flog = L["forces"].reshape(1000, 4, 3)
    stance = G["replay_phase"].astype(bool)
    mu = float(L["param_mu"])
    # (a) support pattern: zero force exactly on the legs the log shows unloaded (logged swing forces are < 0.11 N, OSQP slack)
    assert np.all(F[~stance] == 0.0)
    assert np.array_equal(np.abs(F).max(axis=2) > 1.0, np.abs(Flog).max(axis=2) > 1.0)
    # (b) box and friction pyramid of src/mpc.py:45-46,151-173 on every stance leg of every stage of every tick (1e-6 N)
    U = out["u"].reshape(1000, N, 4, 3)
    c = b["contact"].astype(bool)
    ok = out["status"] == 1
    fz = U[..., 2]
    assert fz[c & ok[:, None, None]].min() >= 3.0 - 1e-6 and fz[c & ok[:, None, None]].max() <= 100.0 + 1e-6
    assert (np.abs(U[..., :2]).max(axis=3) - mu * fz)[ok].max() <= 1e-6
    assert np.all(U[~c] == 0.0)
    # (c) the model: X_out obeys the literal Euler recursion of src/mpc.py:113-117 with the engine's forces
    cfg = S.QPConfig(N=N, delta=d, alpha=1e-2)
    for t in (0, 80, 500, 999):
        X = S.predict_states(b["x0"][t], out["u"][t].reshape(-1), b["r"][t], b["contact"][t], cfg)
        assert np.abs(X - out["X"][t]).max() <= 1e-9
    # (d) the reference's own cost at the two ticks whose predictions it logged: J_engine <= J_log (alpha = 0, continuation)
    two = {k: v[[0, 80]] for k, v in b.items()}
    o0 = gpu_solve(two, N, d, "mixed", alpha=0.0)
    cfg0 = S.QPConfig(N=N, delta=d, alpha=0.0)
    for i, t in enumerate((0, 80)):
        assert o0["status"][i] == 1
        Xlog = np.vstack([L[f"pred{i}_state"], np.full((1, N + 1), -9.81)]).T
        Jlog = S.objective(Xlog, np.zeros(1), two["xdes"][i], cfg0)
        Jeng = S.objective(o0["X"][i], o0["u"][i], two["xdes"][i], cfg0)
        assert Jeng <= Jlog * (1 + 1e-9) and Jeng > 0.2 * Jlog, (t, Jeng, Jlog)
        j = int(np.where(q["ticks"] == t)[0][0])
        assert abs(Jeng - golden["qp_optima"]["N60_a0_J"][j]) <= 1e-6 * golden["qp_optima"]["N60_a0_J"][j]
    # (e) the engine's converged stage-0 forces next to the log's unconverged ones: same order of magnitude of total vertical
    # force on the ticks where all four feet stand (the log's OSQP answers are within its 1e-3 tolerances of SOME feasible point)
    four = stance.all(axis=1)
    tot, totlog = F[four][:, :, 2].sum(axis=1), Flog[four][:, :, 2].sum(axis=1)
    assert np.median(np.abs(tot - totlog) / totlog) <= 0.5


def test_reference_horizon_nonfinite_and_ragged_batches(golden):
    q = golden["qp_inputs"]
    b = {"x0": q["N60_x0"].copy(), "r": q["N60_r"], "contact": q["N60_contact"], "xdes": q["N60_xdes"], "mu": np.full(10, 1.0)}
    b["x0"][3, 4] = np.nan
    out = gpu_solve(b, 60, 0.01, "mixed")
    assert out["status"][3] == -1 and np.all(out["u"][3] == 0)
    keep = np.arange(10) != 3
    assert np.all(out["status"][keep] == 1)
    assert rel_err(out["u"][keep], golden["qp_optima"]["N60_a1e-2_u"][keep]).max() <= 1e-4
    # determinism: same inputs, same bits; and a batch larger than the resident workgroups (persistent loop)
    out2 = gpu_solve(b, 60, 0.01, "mixed")
    assert np.array_equal(out["u"], out2["u"], equal_nan=True)
    # other horizons than the three the fixtures hold: N = 30 (the presentation's second setting, slide 17) against the oracle
    b30 = mpcqp.synth.make_batch(24, 30, 0.02, 5, ("trot", "gallop", "amble"), (0.5, 1.0))
    o30 = gpu_solve(b30, 30, 0.02, "mixed")
    import os
    from conftest import ORACLE_SO
    olib = mpcqp.Library(ORACLE_SO)
    eng = mpcqp.Engine(olib, olib.default_config(N=30, delta=0.02, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
    ref = eng.solve_batch_host(b30["x0"], b30["r"], b30["contact"], b30["xdes"], b30["mu"])
    ok = o30["status"] == 1
    assert ok.mean() >= 0.95 and rel_err(o30["u"][ok], ref["u"][ok]).max() <= 1e-4


@pytest.mark.parametrize("w7,precision", [(5e3, "mixed"), (4e4, "mixed"), (4e4, "f64")])
def test_stage_engine_zoh_and_other_constants(oracle_solve, w7, precision):
    """Nothing in the stage-wise engine is tied to Euler or to the Lite3 defaults: exact zero-order hold (theta = 1/2 in Gam) and other
    mass / inertia / weights / force bounds / alpha at N = 30, against the oracle; and the two discretisations differ (not vacuous).
    w7 != w[6]: the weight on the horizontal angular velocity is anisotropic, i.e. a yaw-dependent coupled 2 x 2 block in the body-rate
    coordinates the recursion works in (the dense engine does not take such weights; every horizon runs them here)."""
    b = mpcqp.synth.make_batch(48, 30, 0.02, 77, ("trot", "gallop", "amble"), (0.4, 0.8))
    kw = dict(m=12.5, Ibody_inv=[1 / 0.4, 1 / 0.9, 1 / 1.3], w=[2e4, 1e4, 3e4, 1e5, 2e5, 3e5, 5e3, w7, 1e4, 1e4, 2e4, 3e4, 0.0],
              alpha=3e-2, f_min=5.0, f_max=150.0)
    assert np.abs(np.sin(b["x0"][:, 2])).max() > 0.1    # yaw away from zero: the coupling term is exercised
    outs = {}
    for disc in (mpcqp.DISC_EULER, mpcqp.DISC_ZOH):
        ref = oracle_solve(b, N=30, delta=0.02, disc=disc, **kw)
        out = gpu_solve(b, 30, 0.02, precision, disc=disc, **kw)
        ok = out["status"] == 1
        assert ok.mean() >= 0.95, ok.mean()
        assert rel_err(out["u"], ref["u"])[ok].max() <= 1e-4 and np.abs(out["X"] - ref["X"])[ok].max() <= 1e-4
        outs[disc] = ref["u"]
    assert rel_err(outs[mpcqp.DISC_ZOH], outs[mpcqp.DISC_EULER]).max() > 1e-3


def test_device_rollout_at_the_reference_horizon():
    """mpcqp_rollout with the stage-wise engine underneath (N = 60, delta = 0.01, the reference's step durations 10 / 5 ticks): the device
    loop (expand -> solve -> advance per tick, no host round trips) against the checker's literal loop on the same robots."""
    from conftest import ORACLE_SO
    rb = mpcqp.synth.make_rollout_batch(6, N=60, delta=0.01, seed=5, ss=10, ds=5, total_steps=20)
    olib = mpcqp.Library(ORACLE_SO)
    T = 6
    ref = mpcqp.Engine(olib, olib.default_config(N=60, delta=0.01, max_iter=20000, eps_abs=1e-10, eps_rel=1e-10, polish_max=30)).rollout_host(
        rb["x"], rb["ref"], rb["plan_pos"], rb["plan_feet_id"], rb["plan_meta"], rb["tick"], rb["mu"], T)
    sol = mpcqp.MPCBatch(N=60, delta=0.01, io_dtype="f64", precision="mixed")
    f = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda").contiguous()
    x, rf, tk = f(rb["x"]), f(rb["ref"]), f(rb["tick"], torch.int32)
    out = sol.rollout(x, rf, f(rb["plan_pos"]), f(rb["plan_feet_id"], torch.uint8), f(rb["plan_meta"], torch.int32), tk, f(rb["mu"]), T)
    torch.cuda.synchronize()
    assert np.all(out["solved"].cpu().numpy() == T) and np.all(ref["solved"] == T)
    F = out["forces"].cpu().numpy()
    assert np.abs(F - ref["forces"]).max() <= 1e-4 * np.abs(ref["forces"]).max()      # forces of every tick
    assert np.abs(x.cpu().numpy() - ref["x"]).max() <= 1e-5                            # the states after T ticks


def test_reference_horizon_warm_start(golden):
    """The reference seeds every solve with its previous solution (src/mpc.py:270-271).  On the stage-wise engine: restarting the ten
    golden N = 60 ticks from their own optimum (and the engine's multiplier record) costs no ADMM block and returns the cold answer; and
    a tick seeded with the PREVIOUS tick's solution, shifted by the engine (MPCQP_FLAG_WARM_SHIFT), reaches the same optimum as a cold
    solve with fewer iterations."""
    q, opt = golden["qp_inputs"], golden["qp_optima"]
    b = {"x0": q["N60_x0"], "r": q["N60_r"], "contact": q["N60_contact"], "xdes": q["N60_xdes"], "mu": np.full(10, float(q["mu"]))}
    sol = mpcqp.MPCBatch(N=60, delta=0.01, io_dtype="f64", precision="mixed", warm_start=True)
    dev = sol.upload(b)
    o1 = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()      # zeros = no guess: cold
    u1, it1 = o1["u"].cpu().numpy().copy(), o1["iters"].cpu().numpy().copy()
    o2 = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()      # seeded with its own solution
    u2, it2, st2 = o2["u"].cpu().numpy(), o2["iters"].cpu().numpy(), o2["status"].cpu().numpy()
    assert np.all(st2 == 1)
    assert rel_err(u1, opt["N60_a1e-2_u"]).max() <= 1e-4 and rel_err(u2, opt["N60_a1e-2_u"]).max() <= 1e-4
    assert np.all(it2 % 1000 == 0) and np.all(it1 % 1000 > 0)               # a KKT point as the guess: one polish step, no ADMM block
    # consecutive ticks of the logged run, the guess one tick old
    ticks = np.arange(100, 140)
    run = logged_run_inputs(golden, 60, ticks)
    cold = gpu_solve(run, 60, 0.01, "mixed")
    ws = mpcqp.MPCBatch(N=60, delta=0.01, io_dtype="f64", precision="mixed", warm_start=True, warm_shift=True)
    its = []
    for i in range(len(ticks)):
        one = {k: v[i:i + 1] for k, v in run.items()}
        d = ws.upload(one)
        o = ws.solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"]); torch.cuda.synchronize()              # the buffer holds tick i - 1's solution
        assert int(o["status"][0]) == 1
        assert rel_err(o["u"].cpu().numpy(), cold["u"][i:i + 1]).max() <= 1e-4
        its.append(int(o["iters"][0]) % 1000)
    # (measured 42 against 70: the cold solves are Anderson-accelerated first blocks of 0.7 x 100 iterations, all solved in that block)
    assert np.mean(its[1:]) < 0.75 * np.mean(cold["iters"][1:] % 1000), (np.mean(its[1:]), np.mean(cold["iters"][1:] % 1000))


@pytest.mark.parametrize("N", [1, 2, 7, 33, 64])
def test_stage_engine_horizon_edges(oracle_solve, N):
    """Every horizon the header promises (1..64): the shortest ones, an odd one, and the largest (4 N = 256 leg-stages: every lane of the
    workgroup owns one), against the oracle; batch sizes around the resident-workgroup count are covered by the 1000-tick test."""
    b = mpcqp.synth.make_batch(40, N, 0.02, 100 + N, ("trot", "gallop", "amble", "pronk"), (0.5, 1.0))
    ref = oracle_solve(b, N=N, delta=0.02)
    out = gpu_solve(b, N, 0.02, "mixed")
    ok = out["status"] == 1
    assert ok.mean() >= 0.95, (N, ok.mean())
    assert rel_err(out["u"], ref["u"])[ok].max() <= 1e-4 and np.abs(out["X"] - ref["X"])[ok].max() <= 1e-4
    swing = np.repeat(b["contact"] == 0, 3, axis=2).reshape(40, N, 12)
    assert np.all(out["u"][swing] == 0)
