import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
LISTED_MAX = int(os.environ.get('LISTED_MAX', '0'))   # MpcQpConfig.listed_max (0: default 4 device-fills, -1: always queued)
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
for prec in ("mixed", "f64"):
    for B in (1024, 4096, 16384):
        for seed in (20250811, 3):
            b = mpcqp.synth.make_batch(B, 20, 0.03, seed, G, M)
            for flags in (1, 1 | 8):
                sol = mpcqp.MPCBatch(listed_max=LISTED_MAX, N=20, precision=prec, flags=flags)
                dev = sol.upload(b)
                ms = []
                for _ in range(4):
                    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
                st = o["status"].cpu().numpy()
                print(f"LISTED_MAX={LISTED_MAX} {prec} B={B} seed={seed} {'plain ' if flags & 8 else 'ordered'}: {np.median(ms):.3f} ms  {B / np.median(ms) / 1e3:.3f} M QP/s unsolved {int((st != 1).sum())}", flush=True)
