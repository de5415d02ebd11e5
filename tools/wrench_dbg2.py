"""Developer check: QPs the wrench engine fails on, solved alone with growing iteration caps."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from gpu_check import oracle_solve, relerr
B = 4096
batch = mpcqp.synth.config3(B)
sol = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f64", precision="mixed")
dev = sol.upload(batch)
o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
st = o["status"].cpu().numpy().copy(); it = o["iters"].cpu().numpy().copy()
bad = np.where(st != 1)[0]
print("failing", bad.tolist())
sub = {k: (v[bad] if isinstance(v, np.ndarray) and len(v) == B else v) for k, v in batch.items()}
ref = oracle_solve(sub, 10, 0.03)
for mi in (100, 200, 300, 400):
    for prec in ("mixed", "f64"):
        s2 = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f64", precision=prec, max_iter=mi)
        d2 = s2.upload(sub)
        o2 = s2.solve_batch(d2["x0"], d2["r"], d2["contact"], d2["xdes"], d2["mu"]); torch.cuda.synchronize()
        u = o2["u"].cpu().numpy(); e = relerr(u, ref["u"])
        print(f"max_iter {mi} prec {prec}: status {o2['status'].cpu().numpy().tolist()} iters {o2['iters'].cpu().numpy().tolist()}")
        print("     res", np.round(o2["res"].cpu().numpy(), 4).tolist(), " nan-u", np.isnan(u).reshape(len(u), -1).any(axis=1).astype(int).tolist(), " err", [f"{x:.1e}" for x in e])
s3 = mpcqp.MPCBatch(N=10, delta=0.03, io_dtype="f64", precision="mixed", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_TILE_KERNEL)
d3 = s3.upload(sub)
o3 = s3.solve_batch(d3["x0"], d3["r"], d3["contact"], d3["xdes"], d3["mu"]); torch.cuda.synchronize()
print("tile kernel: status", o3["status"].cpu().numpy().tolist(), "iters", o3["iters"].cpu().numpy().tolist())
