"""Diagnostic: the ADMM iterate after exactly K iterations (polish off, tolerances 0, no rho adaptation inside the block) --
engine MIXED / F64 vs the CPU oracle's OSQP loop.  Equal iterates = the same iteration is being run."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from gpu_check import relerr
B = 256
for N, mk in ((10, mpcqp.synth.config3), (20, mpcqp.synth.config5)):
    batch = mk(B)
    for K in (10, 24, 25, 50, 200):
        lib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
        cfg = lib.default_config(N=N, delta=0.03, eps_abs=0.0, eps_rel=0.0, max_iter=K, check_every=K, flags=0)
        ref = mpcqp.Engine(lib, cfg).solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"])
        line = f"N={N} K={K}:"
        for prec in ("mixed", "f64"):
            sol = mpcqp.MPCBatch(N=N, delta=0.03, io_dtype="f64", precision=prec, flags=0, eps_abs=0.0, eps_rel=0.0, max_iter=K, check_every=K)
            dev = sol.upload(batch)
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
            e = relerr(o["u"].cpu().numpy(), ref["u"])
            it = o["iters"].cpu().numpy()
            line += f"  {prec}: err max {e.max():.2e} med {np.median(e):.2e} iters {np.unique(it % 1000).tolist()} (oracle {np.unique(ref['iters'] % 1000).tolist()})"
        print(line, flush=True)
