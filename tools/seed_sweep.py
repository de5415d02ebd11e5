"""Developer tool: robustness over workload seeds -- solved fraction, KKT residuals, parity with the CPU oracle on a sample."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
oeng = mpcqp.Engine(olib, olib.default_config(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
sol = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed")
worst = 0.0
for seed in range(101, 109):
    b = mpcqp.synth.make_batch(4096, 10, 0.03, seed, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
    dev = sol.upload(b)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy(); res = out["res"].cpu().numpy(); u = out["u"].cpu().numpy()
    ok = (st == 1) | (st == 2)
    sub = {k: b[k][:96] for k in ("x0", "r", "contact", "xdes", "mu")}
    ref = oeng.solve_batch_host(sub["x0"], sub["r"], sub["contact"], sub["xdes"], sub["mu"], want_X=False)["u"].reshape(96, 10, 12)
    e = np.abs(u[:96] - ref).max(axis=(1, 2)) / np.maximum(np.abs(ref).max(axis=(1, 2)), 1e-12)
    worst = max(worst, e[ok[:96]].max())
    print(f"seed {seed}: solved {ok.mean():.4f}  kernel {sol.last_kernel_ms():.3f} ms  max primal res {res[ok, 0].max():.1e}  max dual res {res[ok, 1].max():.1e}  "
          f"max rel err vs oracle (96 QPs) {e[ok[:96]].max():.1e}", flush=True)
print("worst rel err", worst)
