"""Diagnostic: early-rho-check threshold (env MPCQP_ADAPT_THR, read at mpcqp_create) over seeds that are NOT the bench seed,
plus the bench batch and a 65536 batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
if os.environ.get("MPCQP_LIB"):
    from mpcqp import _capi
    _capi._product = _capi.Library(os.environ["MPCQP_LIB"])
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
N = int(os.environ.get("SW_N", "10")); prec = os.environ.get("SW_PREC", "mixed")
thrs = [float(x) for x in os.environ.get("SW_THR", "3,4,5,6,8,10").split(",")]
cases = [("seed1", mpcqp.synth.make_batch(4096, N, 0.03, 1, G, M)), ("seed2", mpcqp.synth.make_batch(4096, N, 0.03, 2, G, M)),
         ("seed3", mpcqp.synth.make_batch(4096, N, 0.03, 3, G, M)), ("amble7", mpcqp.synth.make_batch(4096, N, 0.03, 7, ("amble",), M)),
         ("seed4", mpcqp.synth.make_batch(4096, N, 0.03, 4, G, M)), ("seed5", mpcqp.synth.make_batch(4096, N, 0.03, 5, G, M)), ("seed6", mpcqp.synth.make_batch(4096, N, 0.03, 6, G, M)),
         ("bench", mpcqp.synth.config3(4096) if N == 10 else mpcqp.synth.config5(4096))]
if N == 10: cases.append(("64k", mpcqp.synth.config4(65536)))
for name, batch in cases:
    B = len(batch["x0"])
    for thr in thrs:
        kw = {"polish_max": int(os.environ["SW_PM"])} if os.environ.get("SW_PM") else {}
        kw["adapt_thr"] = float(thr)
        if os.environ.get("SW_CE"): kw.update(check_every=int(os.environ["SW_CE"]), max_iter=int(os.environ.get("SW_MI", "400")))
        sol = mpcqp.MPCBatch(N=N, delta=0.03, precision=prec, **kw)
        dev = sol.upload(batch)
        ms = []
        for _ in range(9):
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
        st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
        print(f"{name:7s} N={N} {prec:5s} thr={thr:<5g} {B / np.median(ms) / 1e3:7.3f} M QP/s  ms {np.median(ms):.3f} unsolved {int(np.sum(st != 1))} iters {np.mean(it % 1000):.1f} max {np.max(it % 1000)} psteps {np.mean(it // 1000):.2f} max {np.max(it // 1000)}", flush=True)
