"""Developer check: throughput at B = 4096 and a close look at QPs the wrench engine fails on."""
import os, sys, json
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from gpu_check import oracle_solve, relerr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = mpcqp.synth.config3(B)
ref = oracle_solve(batch, 10, 0.03)
for label, kw in (("wrench f32io", dict(io_dtype="f32", precision="mixed")), ("wrench f64io", dict(io_dtype="f64", precision="mixed")),
                  ("tile f32io", dict(io_dtype="f32", precision="mixed", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_TILE_KERNEL))):
    sol = mpcqp.MPCBatch(N=10, delta=0.03, **kw)
    dev = sol.upload(batch)
    ms = []
    for _ in range(6):
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
    u = o["u"].cpu().numpy().astype(np.float64); st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy(); res = o["res"].cpu().numpy()
    e = relerr(u, ref["u"])
    ok = (st == 1)
    print(f"[{label}] B={B} ms {['%.3f' % m for m in ms]} -> {B / min(ms) / 1e3:.3f} M QP/s; status {np.bincount(st + 1, minlength=5).tolist()} "
          f"err_solved_max {e[ok].max():.2e} iters {(it % 1000).mean():.1f} psteps {(it // 1000).mean():.2f}", flush=True)
    bad = np.where(~ok | ~np.isfinite(e))[0]
    for i in bad[:10]:
        print(f"   QP {i}: status {st[i]} iters {it[i]} res {res[i]} err {e[i]} gait {batch['gait_ids'][i]} mu {batch['mu'][i]} t0 {batch['t0'][i]} "
              f"u nan {np.isnan(u[i]).sum()} umax {np.nanmax(np.abs(u[i])):.3g}", flush=True)
    hist = np.bincount(it // 1000, minlength=12)
    print("   polish-step histogram", hist.tolist(), " admm-iter histogram", np.unique(it % 1000, return_counts=True)[0].tolist()[:12], flush=True)
