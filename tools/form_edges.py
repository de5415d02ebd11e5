"""Developer check: batch sizes around the launch-form thresholds (plain <= 2048 < listed <= 8192 < queued at N = 10; 512 / 2048 at
N = 20) give the bits of the plain form, both buffer types, X requested or not."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
bad = 0
for N, sizes in ((10, (1, 63, 2048, 2049, 2050, 8191, 8192, 8193, 9000)), (20, (511, 512, 513, 2048, 2049, 2300))):
    for B in sizes:
        b = mpcqp.synth.make_batch(B, N, 0.03, 77 + B, G, M)
        outs = []
        for flags in (1 | 8, 1, 1 | 64):
            sol = mpcqp.MPCBatch(N=N, precision="mixed", io_dtype="f32", flags=flags)
            dev = sol.upload(b)
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True); torch.cuda.synchronize()
            outs.append({k: v.cpu().numpy().copy() for k, v in o.items() if v is not None})
        same = all(np.array_equal(outs[0][k], outs[j][k]) for k in ("u", "X", "status", "iters") for j in (1, 2))
        bad += not same
        print(f"N={N} B={B}: {'same bits' if same else 'DIFFERENT'} as the plain form; solved {np.mean(outs[1]['status'] == 1):.4f}", flush=True)
print("FAILED" if bad else "all equal")
