"""Developer study: where can the alpha = 0 continuation of the stage-wise engine end (MpcQpConfig.alpha_floor)?  Golden log ticks, N = 60."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp, qp_spec as S
from test_gpu_reference_horizon import gpu_solve
q = np.load(os.path.join(REPO, "tests", "golden", "qp_inputs.npz")); opt = np.load(os.path.join(REPO, "tests", "golden", "qp_optima.npz"))
for N, flags in ((60, mpcqp.FLAG_POLISH), (10, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL), (20, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL)):
    bb = {"x0": q[f"N{N}_x0"], "r": q[f"N{N}_r"], "contact": q[f"N{N}_contact"], "xdes": q[f"N{N}_xdes"], "mu": np.full(10, float(q["mu"]))}
    cfg = S.QPConfig(N=N, delta=0.01, alpha=0.0)
    for floor in (1e-4, 5e-5, 3e-5, 2e-5, 1e-5):
        o = gpu_solve(bb, N, 0.01, "mixed", flags, alpha=0.0, alpha_floor=floor)
        ok = o["status"] == 1
        eJ = [abs(S.objective(o["X"][i], o["u"][i], bb["xdes"][i], cfg) / opt[f"N{N}_a0_J"][i] - 1) for i in range(10)]
        eX = np.abs(o["X"] - opt[f"N{N}_a0_X"]).reshape(10, -1).max(axis=1)
        print(f"N={N} floor {floor:.0e}: solved {int(ok.sum())}/10, objective rel err max (solved) {max([e for e, k in zip(eJ, ok) if k] or [float('nan')]):.1e}, "
              f"states max (solved) {eX[ok].max() if ok.any() else float('nan'):.1e}; polish steps {(o['iters'] // 1000).tolist()}", flush=True)
