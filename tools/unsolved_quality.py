"""Diagnostic: how good is the iterate the engine returns with status MAX_ITER (polish never accepted)?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from gpu_check import oracle_solve, relerr
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
for N, seeds in ((10, (1, 4, 7, 9, 11, 12)), (20, (2, 3, 6))):
    for seed in seeds:
        b = mpcqp.synth.make_batch(4096, N, 0.03, seed, G if seed != 7 else ("amble",), M)
        sol = mpcqp.MPCBatch(N=N, precision="mixed", io_dtype="f64")
        dev = sol.upload(b)
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True); torch.cuda.synchronize()
        st = o["status"].cpu().numpy(); bad = np.nonzero(st != 1)[0]
        if len(bad) == 0:
            print(f"N={N} seed {seed}: all solved"); continue
        sub = {k: (v[bad] if isinstance(v, np.ndarray) and len(v) == 4096 else v) for k, v in b.items()}
        ref = oracle_solve(sub, N, 0.03)
        e = relerr(o["u"].cpu().numpy()[bad], ref["u"]); eX = np.abs(o["X"].cpu().numpy()[bad] - ref["X"]).max(axis=(1, 2))
        print(f"N={N} seed {seed}: unsolved {bad.tolist()} status {st[bad].tolist()} iters {o['iters'].cpu().numpy()[bad].tolist()} GRF rel err {np.round(e, 5).tolist()} state err {np.round(eX, 6).tolist()} oracle status {ref['status'].tolist()}", flush=True)
