"""Warm start (MPCQP_FLAG_WARM_START): the reference seeds every solve with its previous solution
(`opt.set_initial(U, sol.value(U))`, src/mpc.py:270-271).  The engine must return the SAME optimum as a cold solve
(tolerances of tests/test_gpu_parity.py: |u - u_oracle|_inf / |u_oracle|_inf <= 1e-4 for MIXED), whatever the guess,
and use the guess when it is good: no ADMM iterations for the QPs whose active set the guess already has."""
import numpy as np
import pytest
import torch

import mpcqp
from conftest import rel_err

pytestmark = pytest.mark.gpu


def solved(st):
    return (st == 1) | (st == 2)


def next_tick(b, X, delta=0.03):
    """The QP one control tick later: the robot sits at the first predicted state, the references move on by one
    step (src/mpc.py:261-262) and the gait clock advances (mpcqp.synth.contact_schedule)."""
    nb = {k: np.array(v, copy=True) for k, v in b.items()}
    nb["x0"] = X[:, 1, :].copy()
    nb["x0"][:, 12] = b["x0"][:, 12]
    step = b["xdes"][:, 1, :] - b["xdes"][:, 0, :]
    nb["xdes"] = b["xdes"] + step[:, None, :]
    nb["xdes"][:, :, 12] = b["xdes"][:, :, 12]
    nb["t0"] = b["t0"] + 1
    gaits = [mpcqp.synth.GAITS[g] for g in ("trot", "pronk", "amble", "gallop")]
    nb["contact"] = mpcqp.synth.contact_schedule(b["gait_ids"], nb["t0"], b["contact"].shape[1], gaits=gaits)
    feet = b["r"][:, 1] + b["xdes"][:, 1, None, 3:6]                      # fixed world footholds of the synthetic batch
    nb["r"] = feet[:, None, :, :] - nb["xdes"][:, :-1, None, 3:6]
    nb["r"][:, 0] = feet - nb["x0"][:, None, 3:6]
    return nb


@pytest.fixture(scope="module")
def warm_setup():
    b = mpcqp.synth.config3(512)
    cold = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed")
    warm = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", warm_start=True)
    return b, cold, warm


def run(sol, b, u_init=None):
    dev = sol.upload(b)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True, u_init=u_init)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in out.items()}


def test_first_call_without_guess_is_a_cold_solve(warm_setup, oracle_solve):
    b, cold, warm = warm_setup
    c = run(cold, b)
    w = run(mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", warm_start=True), b)   # fresh engine: zeros in the buffer
    assert np.array_equal(c["status"], w["status"]) and np.array_equal(c["iters"], w["iters"])
    assert np.array_equal(c["u"], w["u"])


def test_restart_from_the_optimum_needs_no_admm(warm_setup, oracle_solve):
    b, cold, warm = warm_setup
    ref = oracle_solve(b)
    c = run(cold, b)
    w = run(warm, b, u_init=torch.as_tensor(c["u"]).cuda())
    ok = solved(w["status"])
    assert ok.all()
    assert rel_err(w["u"], ref["u"]).max() <= 1e-4
    assert np.mean(w["iters"] % 1000 == 0) >= 0.98          # no ADMM block
    assert np.mean(w["iters"] // 1000 == 1) >= 0.95          # one polish step on the guess's own active set
    # and the engine keeps its own solution as the next guess: a second call without u_init behaves the same
    w2 = run(warm, b)
    assert solved(w2["status"]).all() and np.mean(w2["iters"] % 1000 == 0) >= 0.98
    assert rel_err(w2["u"], ref["u"]).max() <= 1e-4


def test_next_tick_warm_equals_cold_and_is_cheaper(warm_setup, oracle_solve):
    """Closed-loop use: tick t+1 seeded from tick t.  Whatever the seed, the optimum is the cold one; a good seed is used."""
    b, cold, _ = warm_setup
    c0 = run(cold, b)
    nb = next_tick(b, c0["X"])
    ref = oracle_solve(nb)
    c1 = run(cold, nb)
    okc = solved(c1["status"])
    assert okc.mean() >= 0.97 and rel_err(c1["u"], ref["u"])[okc].max() <= 1e-4
    cold_admm = (c1["iters"] % 1000).mean()

    def check(w1, name):
        ok = solved(w1["status"])
        assert ok.mean() >= 0.97, name
        assert rel_err(w1["u"], ref["u"])[ok].max() <= 1e-4, name
        assert np.abs(w1["X"][ok] - ref["X"][ok]).max() <= 1e-4, name

    # primal-only guesses on fresh engines (no multiplier record yet): unshifted like the reference, and shifted by the caller
    shifted = np.concatenate([c0["u"][:, 1:], c0["u"][:, -1:]], axis=1)
    for name, guess in (("unshifted", c0["u"]), ("shifted", shifted)):
        eng = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", warm_start=True)
        w1 = run(eng, nb, u_init=torch.as_tensor(np.ascontiguousarray(guess)).cuda())
        check(w1, name)
        if name == "shifted":   # a good guess is used: a third of the QPs need no ADMM block at all
            assert np.mean(w1["iters"] % 1000 == 0) >= 0.3
    # the natural flow: one engine solves tick t, then tick t+1; its output buffer and multiplier record carry over and the
    # engine moves both up by one stage (MPCQP_FLAG_WARM_SHIFT)
    eng = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", warm_start=True, warm_shift=True)
    w0 = run(eng, b)
    assert np.array_equal(w0["u"], c0["u"])                                  # first call: nothing to start from
    w1 = run(eng, nb)
    check(w1, "carried over")
    assert (w1["iters"] % 1000).mean() < 0.7 * cold_admm, (w1["iters"] % 1000).mean()


def test_garbage_guess_still_reaches_the_optimum(warm_setup, oracle_solve):
    b, cold, warm = warm_setup
    ref = oracle_solve(b)
    rng = np.random.default_rng(5)
    g = rng.uniform(-50.0, 120.0, size=(512, 10, 12))
    g[::7] = np.nan                                            # non-finite entries are dropped, not propagated
    w = run(warm, b, u_init=torch.as_tensor(g).cuda())
    ok = solved(w["status"])
    assert ok.mean() >= 0.97
    assert rel_err(w["u"], ref["u"])[ok].max() <= 1e-4
    swing = np.repeat(b["contact"] == 0, 3, axis=2).reshape(512, 10, 12)
    assert np.all(w["u"][swing] == 0)


def test_gait_entry_and_general_kernel_accept_the_flag(warm_setup, oracle_solve):
    b, cold, warm = warm_setup
    g = mpcqp.synth.make_gait_batch(128)
    exp = mpcqp.synth.expand_gait_batch(g)
    ref = oracle_solve(exp)
    sol = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", warm_start=True)
    dev = sol.upload_gait(g)
    args = (dev["x0"], dev["ref"], dev["feet0"], dev["footholds"], dev["gait"], dev["feet_id"], dev["mu"])
    o1 = {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in sol.solve_batch_gait(*args).items()}
    o2 = {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in sol.solve_batch_gait(*args).items()}
    for o in (o1, o2):
        ok = solved(o["status"])
        assert ok.mean() >= 0.97 and rel_err(o["u"], ref["u"])[ok].max() <= 1e-4
    assert (o2["iters"] % 1000).mean() < 0.2 * (o1["iters"] % 1000).mean()
    # stage-wise engine at the same horizon: same flag, same optimum
    gen = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_WARM_START | mpcqp.FLAG_STAGE_KERNEL)
    sub = {k: b[k][:64] for k in ("x0", "r", "contact", "xdes", "mu")}
    o = run(gen, sub)
    o = run(gen, sub)
    refs = oracle_solve(sub)
    ok = solved(o["status"])
    assert ok.mean() >= 0.97 and rel_err(o["u"], refs["u"])[ok].max() <= 1e-4
