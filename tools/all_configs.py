"""Developer tool: throughput on every BASELINE.json configuration (2-5) with the default engine settings.
usage: all_configs.py [config5 ...] [--precision f64]   (names select configurations by prefix)"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
PREC = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "mixed"
ONLY = [a for a in sys.argv[1:] if a.startswith("config")]
for name, mk, N in (("config2 B=1024 trot mu=1", lambda: mpcqp.synth.config2(1024), 10), ("config3 B=4096 mixed", lambda: mpcqp.synth.config3(4096), 10),
                    ("config4 B=65536 mixed", lambda: mpcqp.synth.config4(65536), 10), ("config5 B=4096 N=20", lambda: mpcqp.synth.config5(4096), 20)):
    if ONLY and not any(name.startswith(o) for o in ONLY):
        continue
    b = mk()
    sol = mpcqp.MPCBatch(N=N, precision=PREC)
    dev = sol.upload(b)
    for _ in range(3):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        ms.append(sol.last_kernel_ms())
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    print(f"{name:26s}: {np.median(ms):8.3f} ms  {len(st) / np.median(ms) * 1e3:12,.0f} QP/s  solved {np.mean((st == 1) | (st == 2)):.4f}  admm {np.mean(it % 1000):.0f}  polish {np.mean(it // 1000):.2f}")
