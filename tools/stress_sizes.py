"""Developer tool: one engine, many batch sizes in a row (buffer growth, both launch forms, warm-start record)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
big = mpcqp.synth.config4(70000)
for warm in (False, True):
    sol = mpcqp.MPCBatch(N=10, precision="mixed", warm_start=warm, warm_shift=warm)
    for B in (1500, 5000, 300, 70000, 1024, 1023, 800, 600, 513, 512, 1, 4096, 4096):
        sub = {k: big[k][:B] for k in ("x0", "r", "contact", "xdes", "mu")}
        dev = sol.upload(sub)
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
        torch.cuda.synchronize()
        st = out["status"].cpu().numpy(); u = out["u"].cpu().numpy()
        assert np.isfinite(u).all() and np.mean((st == 1) | (st == 2)) >= 0.97, (warm, B)
        print(f"warm={warm} B={B:6d}: solved {np.mean((st == 1) | (st == 2)):.4f}  kernel {sol.last_kernel_ms():.3f} ms")
print("ok")
