"""Developer study: how many of the QPs that need three or four ADMM rounds are among the H dearest-PREDICTED ones of the dispatch order
(the friction-demand predictor of mpcqp_common.h, restated in numpy)?  Decides whether giving the first H workgroups a SIMD of their own pays."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
allg = ("trot", "pronk", "amble", "gallop")
for seed in (20250809, 1, 2, 3, 4):
    b = mpcqp.synth.make_batch(4096, 10, 0.03, seed, allg, (0.3, 0.5, 0.7, 1.0))
    r, c, mu = b["r"], b["contact"].astype(bool), b["mu"]
    B = 4096
    score = np.zeros(B); nst_tot = c.sum(axis=(1, 2)).astype(float)
    for k in range(10):
        n = c[:, k].sum(axis=1)
        for i in np.where(n == 2)[0]:
            a, bb = np.where(c[i, k])[0]
            ra, rb = r[i, k, a], r[i, k, bb]
            d = abs(ra[0] * rb[1] - ra[1] * rb[0]) / max(np.hypot(ra[0] - rb[0], ra[1] - rb[1]), 1e-6)
            h = max(-0.5 * (ra[2] + rb[2]), 1e-3)
            score[i] = max(score[i], d / h)
        for i in np.where(n == 1)[0]:
            a = np.where(c[i, k])[0][0]
            score[i] = max(score[i], np.hypot(r[i, k, a, 0], r[i, k, a, 1]) / max(-r[i, k, a, 2], 1e-3))
    us = 2.2 * nst_tot + 34.0 * np.minimum(score / np.maximum(np.abs(mu), 1e-3), 2.0)
    cls = np.minimum((us * 0.1).astype(int), 15)
    sol = mpcqp.MPCBatch(N=10, precision="mixed")
    dev = sol.upload(b)
    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
    it = o["iters"].cpu().numpy(); admm = it % 1000; ps = it // 1000
    cost = admm * 0.4 + ps * 12.0
    hard = admm > 200
    order = np.argsort(-cls, kind="stable")            # dearest class first (position inside a class: arrival order on the device)
    rank = np.empty(B, int); rank[order] = np.arange(B)
    line = f"seed {seed}: class sizes {np.bincount(cls, minlength=16).tolist()}; QPs with > 200 iterations: {hard.sum()}"
    for H in (64, 128, 256, 512, 1024):
        line += f" | top {H}: {int((hard & (rank < H)).sum())}"
    # an oracle order for comparison
    print(line, flush=True)
    top = np.argsort(-cost)[:12]
    print("   the 12 costliest: class", cls[top].tolist(), "rank", rank[top].tolist(), "iters", admm[top].tolist())
