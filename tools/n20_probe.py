"""Diagnostic: horizon-20 (config 5) throughput / solved fraction over a few knob settings; what the unsolved QPs look like."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
B = 4096
batch = mpcqp.synth.config5(B)
for prec, kw in (("mixed", {}), ("mixed", dict(polish_max=6)), ("mixed", dict(check_every=300, max_iter=1200)), ("mixed", dict(max_iter=1600)), ("f64", {})):
    sol = mpcqp.MPCBatch(N=20, delta=0.03, precision=prec, **kw)
    dev = sol.upload(batch)
    ms = []
    for _ in range(4):
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy(); res = o["res"].cpu().numpy()
    bad = np.nonzero(st != 1)[0]
    print(prec, kw, f"{B / min(ms) / 1e3:.3f} M QP/s  ms {min(ms):.3f} solved {np.mean(st == 1):.5f} iters mean {np.mean(it % 1000):.1f} psteps mean {np.mean(it // 1000):.2f}", flush=True)
    print("   unsolved:", [(int(b), int(it[b]), [float(f"{x:.2e}") for x in res[b]], int(batch["gait_id"][b]) if "gait_id" in batch else -1, float(batch["mu"][b])) for b in bad[:12]], flush=True)
