"""Developer diagnostic: the logged ticks (N = 60) the stage-wise engine leaves unsolved -- how far from the optimum, what the other
precision / a larger iteration cap / the oracle do with them; and the alpha = 0 continuation on the golden ticks."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
from test_gpu_reference_horizon import logged_run_inputs, gpu_solve
g = {k: np.load(os.path.join(REPO, "tests", "golden", k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}
b = logged_run_inputs(g, 60, np.arange(1000))
out = gpu_solve(b, 60, 0.01, "mixed", alpha=1e-2)
bad = np.where(out["status"] != 1)[0]
print("unsolved ticks", bad.tolist(), "iters", out["iters"][bad].tolist(), "res", out["res"][bad].tolist(), flush=True)
print("iters histogram (/100):", np.bincount((out["iters"] % 1000) // 100).tolist(), "polish steps:", np.bincount(out["iters"] // 1000).tolist(), f"kernel {out['ms']:.1f} ms")
if len(bad):
    sub = {k: v[bad] for k, v in b.items()}
    olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
    ref = mpcqp.Engine(olib, olib.default_config(N=60, delta=0.01, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30)).solve_batch_host(sub["x0"], sub["r"], sub["contact"], sub["xdes"], sub["mu"])
    e = np.abs(out["u"][bad] - ref["u"]).reshape(len(bad), -1).max(axis=1) / np.maximum(np.abs(ref["u"]).reshape(len(bad), -1).max(axis=1), 1)
    print("oracle status", ref["status"].tolist(), "iters", ref["iters"].tolist(), "engine (last ADMM iterate) rel err vs oracle", np.array2string(e, precision=2))
    for kw in (dict(precision="f64"), dict(precision="mixed", max_iter=6000), dict(precision="mixed", check_every=200, max_iter=2400), dict(precision="mixed", polish_patience=3),
               dict(precision="mixed", rho=3.0)):
        prec = kw.pop("precision")
        o = gpu_solve(sub, 60, 0.01, prec, alpha=1e-2, **kw)
        e = np.abs(o["u"] - ref["u"]).reshape(len(bad), -1).max(axis=1) / np.maximum(np.abs(ref["u"]).reshape(len(bad), -1).max(axis=1), 1)
        print(prec, kw, "status", o["status"].tolist(), "iters", o["iters"].tolist(), "err", np.array2string(e, precision=1), flush=True)
# alpha = 0 continuation on the golden ticks
q = g["qp_inputs"]
bb = {"x0": q["N60_x0"], "r": q["N60_r"], "contact": q["N60_contact"], "xdes": q["N60_xdes"], "mu": np.full(10, float(q["mu"]))}
for alpha in (1e-5, 3e-6, 0.0):
    for prec in ("mixed", "f64"):
        o = gpu_solve(bb, 60, 0.01, prec, alpha=alpha)
        print(f"alpha {alpha} {prec}: status {o['status'].tolist()} iters {o['iters'].tolist()} res {np.array2string(o['res'].max(axis=1), precision=1)}", flush=True)
for N in (10, 20):
    bb = {"x0": q[f"N{N}_x0"], "r": q[f"N{N}_r"], "contact": q[f"N{N}_contact"], "xdes": q[f"N{N}_xdes"], "mu": np.full(10, float(q["mu"]))}
    o = gpu_solve(bb, N, 0.01, "mixed", mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL, alpha=0.0)
    print(f"N={N} stage alpha 0: status {o['status'].tolist()} iters {o['iters'].tolist()}", flush=True)
