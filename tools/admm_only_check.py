"""Developer check: ADMM-only convergence (polish off, fp64, eps 1e-9) of the wrench engine against the stage-wise engine on the
same QPs -- the two share the algorithm and differ in the linear solve (explicit swept inverse vs Riccati recursion)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
b = mpcqp.synth.config2(32)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
ref = mpcqp.Engine(olib, olib.default_config(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30)).solve_batch_host(b["x0"], b["r"], b["contact"], b["xdes"], b["mu"])
for name, flags in (("wrench", 0), ("stage", mpcqp.FLAG_STAGE_KERNEL)):
    sol = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="f64", flags=flags, max_iter=20000, check_every=100, eps_abs=1e-9, eps_rel=1e-9)
    dev = sol.upload(b)
    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy(); res = o["res"].cpu().numpy()
    u = o["u"].cpu().numpy().reshape(32, -1); ur = ref["u"].reshape(32, -1)
    e = np.abs(u - ur).max(axis=1) / np.maximum(np.abs(ur).max(axis=1), 1)
    print(f"{name}: converged {int((st == 2).sum())}/32; iters {sorted(it.tolist())}; worst err {e.max():.2e}; unconverged residuals {res[st != 2].tolist()}", flush=True)
