"""Developer diagnostic: where does the fp32-chain ADMM of the stage-wise engine leave the fp64 one on the logged ticks it loses?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
from test_gpu_reference_horizon import logged_run_inputs, gpu_solve
g = {k: np.load(os.path.join(REPO, "tests", "golden", k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}
ticks = [22, 85, 212, 23, 0]
b = logged_run_inputs(g, 60, ticks)
for K in (5, 25, 26, 50, 100, 200, 420):
    o = {}
    for prec in ("f64", "mixed"):
        o[prec] = gpu_solve(b, 60, 0.01, prec, 0, max_iter=K, check_every=K, eps_abs=0.0, eps_rel=0.0)
    d = np.abs(o["mixed"]["u"] - o["f64"]["u"]).reshape(len(ticks), -1).max(axis=1)
    print(f"K={K}: |u_mixed - u_f64|_inf per tick {np.array2string(d, precision=2)}; |u_f64|_inf {np.array2string(np.abs(o['f64']['u']).reshape(len(ticks), -1).max(axis=1), precision=1)} res mixed {o['mixed']['res'].tolist()}", flush=True)
# with the polish: per-round view
for prec in ("mixed", "f64"):
    o = gpu_solve(b, 60, 0.01, prec, alpha=1e-2, max_iter=420)
    print(prec, "one round: status", o["status"].tolist(), "iters", o["iters"].tolist(), "res", o["res"].tolist())
q = g["qp_inputs"]
bb = {"x0": q["N60_x0"], "r": q["N60_r"], "contact": q["N60_contact"], "xdes": q["N60_xdes"], "mu": np.full(10, float(q["mu"]))}
for alpha in (1e-5, 0.0):
    o = gpu_solve(bb, 60, 0.01, "mixed", alpha=alpha)
    print(f"alpha {alpha} mixed (12 refinement steps): status {o['status'].tolist()} iters {o['iters'].tolist()} res {np.array2string(o['res'].max(axis=1), precision=1)}", flush=True)
bb = {"x0": q["N10_x0"], "r": q["N10_r"], "contact": q["N10_contact"], "xdes": q["N10_xdes"], "mu": np.full(10, float(q["mu"]))}
o = gpu_solve(bb, 10, 0.01, "mixed", mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL, alpha=0.0)
print(f"N=10 stage alpha 0: status {o['status'].tolist()} iters {o['iters'].tolist()}", flush=True)
