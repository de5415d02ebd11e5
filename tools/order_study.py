"""Diagnostic: how much of the B = 4096 launch time is dispatch order, how much the launch form?  Solve once, then re-run the
same QPs (a) plain form (hardware dispatch in batch order) on the batch sorted by MEASURED cost, by the pre-pass's CLASS
(host emulation of cost_class) and unsorted, (b) queued form on the same three arrangements."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
B = 4096
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)


def classes(b):
    r = b["r"].astype(np.float64); c = b["contact"].astype(bool); mu = b["mu"].astype(np.float64)
    nst = c.sum(axis=2); dem = np.zeros(nst.shape)
    for i in range(len(mu)):
        for k in range(nst.shape[1]):
            st = np.nonzero(c[i, k])[0]
            if len(st) == 2:
                a, bb = st; fx, fy, fz = r[i, k, :, 0], r[i, k, :, 1], r[i, k, :, 2]
                d = abs(fx[a] * fy[bb] - fy[a] * fx[bb]) / max(np.hypot(fx[a] - fx[bb], fy[a] - fy[bb]), 1e-6)
                dem[i, k] = d / max(-0.5 * (fz[a] + fz[bb]), 1e-3)
            elif len(st) == 1:
                l = st[0]; dem[i, k] = np.hypot(r[i, k, l, 0], r[i, k, l, 1]) / max(-r[i, k, l, 2], 1e-3)
    us = 2.2 * nst.sum(1) + 34 * np.minimum(dem.max(1) / mu, 2)
    return np.minimum((us * 0.1).astype(int), 15)


for name, batch in (("bench", mpcqp.synth.config3(B)), ("seed3", mpcqp.synth.make_batch(B, 10, 0.03, 3, G, M)), ("seed5", mpcqp.synth.make_batch(B, 10, 0.03, 5, G, M))):
    def run(bt, flags):
        sol = mpcqp.MPCBatch(N=10, precision="mixed", flags=flags)
        dev = sol.upload(bt)
        ms = []
        for _ in range(8):
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
        return np.median(ms), o["iters"].cpu().numpy()
    t_q, it = run(batch, 1)
    cost = (it % 1000) + 55 * (it // 1000)
    cl = classes(batch)
    perm = {"measured cost, dearest first": np.argsort(-cost, kind="stable"), "pre-pass class, dearest first": np.argsort(-cl, kind="stable"), "batch order": np.arange(B)}
    print(f"{name}:", flush=True)
    for k, p in perm.items():
        bt = {kk: (v[p] if isinstance(v, np.ndarray) and len(v) == B else v) for kk, v in batch.items()}
        tp, _ = run(bt, 1 | 8); tq, _ = run(bt, 1)
        print(f"   {k:32s} plain {tp * 1e3:4.0f} us   queued {tq * 1e3:4.0f} us", flush=True)
