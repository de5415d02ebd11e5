"""Diagnostic: listed vs queued launch form by batch size (MpcQpConfig.listed_max via env LISTED_MAX of this script, in device-fills)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
LISTED_MAX = int(os.environ.get('LISTED_MAX', '0'))   # MpcQpConfig.listed_max (0: default 4 device-fills, -1: always queued)
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
for B in (2049, 4096, 8192, 16384, 32768, 65536):
    for seed in (20250809, 3):
        b = mpcqp.synth.make_batch(B, 10, 0.03, seed, G, M)
        sol = mpcqp.MPCBatch(listed_max=LISTED_MAX, N=10, precision="mixed")
        dev = sol.upload(b)
        ms = []
        for _ in range(6):
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
        st = o["status"].cpu().numpy()
        print(f"LISTED_MAX={LISTED_MAX} B={B} seed={seed}: {np.median(ms) * 1e3:.0f} us  {B / np.median(ms) / 1e3:.2f} M QP/s unsolved {int((st != 1).sum())}", flush=True)
