import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import mpcqp
b = mpcqp.synth.config3(2048)
ref = mpcqp.MPCBatch(N=10, io_dtype="f64", precision="mixed")
dev64 = ref.upload(b)
r = ref.solve_batch(dev64["x0"], dev64["r"], dev64["contact"], dev64["xdes"], dev64["mu"]); torch.cuda.synchronize()
ur = r["u"].cpu().numpy().copy()
for kw in ({}, {"warm_start": True, "warm_shift": False}):
    sol = mpcqp.MPCBatch(N=10, io_dtype="f32", precision="f32", **kw)
    dev = sol.upload(b)
    for rep in range(2):
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
        st = o["status"].cpu().numpy(); u = o["u"].cpu().numpy().astype(np.float64)
        ok = (st == 1) | (st == 2)
        e = np.abs(u - ur).max(axis=(1, 2)) / np.maximum(np.abs(ur).max(axis=(1, 2)), 1.0)
        print(kw, "rep", rep, "solved", ok.mean(), "max err vs mixed", e[ok].max(), "kernel", sol.last_kernel_ms(), "admm mean", (o["iters"].cpu().numpy() % 1000).mean())
