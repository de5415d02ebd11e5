import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
B = 512
batch = mpcqp.synth.config3(B)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
for alpha in (1.0, 1e-1, 1e-2, 1e-3, 1e-4):
    oeng = mpcqp.Engine(olib, olib.default_config(alpha=alpha, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, polish_max=30))
    ref = oeng.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"], want_X=False)
    ur = ref["u"].reshape(B, -1)
    for prec in ("mixed", "f32"):
        sol = mpcqp.MPCBatch(N=10, precision=prec, io_dtype="f64", alpha=alpha)
        dev = sol.upload(batch)
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
        u = out["u"].cpu().numpy().reshape(B, -1)
        ok = (st == 1) | (st == 2)
        e = np.abs(u - ur).max(axis=1) / np.maximum(np.abs(ur).max(axis=1), 1.0)
        print(f"alpha {alpha:7.0e} {prec:5s}: oracle polished {np.mean(ref['status']==1):.3f}  gpu solved {ok.mean():.3f}  err(solved) med {np.median(e[ok]) if ok.any() else float('nan'):.1e} "
              f"max {e[ok].max() if ok.any() else float('nan'):.1e}  admm {np.mean(it%1000):6.1f} polish {np.mean(it//1000):4.2f}  {sol.last_kernel_ms():.2f} ms", flush=True)
