import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp, qp_spec as S
from mpcqp import _capi
_capi._product = _capi.Library(os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc", os.environ.get("MPCQP_LIB", "libmpcqp.so")))
B = 256
batch = mpcqp.synth.config3(B)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
# alpha = 0 reference quantities (unique: objective, X): oracle ADMM-only long run
o0 = mpcqp.Engine(olib, olib.default_config(alpha=0.0, rho=0.3, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, polish_max=30))
r0 = o0.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"])
cfg0 = S.QPConfig(N=10, delta=0.03, alpha=0.0)
J0 = np.array([S.objective(r0["X"][i], r0["u"][i], batch["xdes"][i], cfg0) for i in range(B)])
print("oracle alpha=0 status", np.bincount(r0["status"] + 1, minlength=5).tolist(), flush=True)
for alpha, mi in ((1e-4, 400), (1e-4, 800), (1e-5, 800), (1e-6, 800), (1e-7, 800)):
    for prec in ("mixed", "f64"):
        sol = mpcqp.MPCBatch(N=10, precision=prec, io_dtype="f64", alpha=alpha, max_iter=mi)
        dev = sol.upload(batch)
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
        torch.cuda.synchronize()
        st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
        u = out["u"].cpu().numpy(); X = out["X"].cpu().numpy()
        ok = (st == 1) | (st == 2)
        J = np.array([S.objective(X[i], u[i], batch["xdes"][i], cfg0) for i in range(B)])     # the alpha = 0 objective of the answer
        good = ok & (r0["status"] != 3)
        dJ = np.abs(J - J0) / np.maximum(1, np.abs(J0)); dX = np.abs(X - r0["X"]).reshape(B, -1).max(axis=1)
        print(f"alpha {alpha:7.0e} mi {mi} {prec:5s}: solved {ok.mean():.3f}  vs alpha=0 optimum: dJ max {dJ[good].max():.1e} med {np.median(dJ[good]):.1e}  dX max {dX[good].max():.1e}  "
              f"admm {np.mean(it%1000):6.1f} polish {np.mean(it//1000):4.2f}  {sol.last_kernel_ms():.2f} ms", flush=True)
