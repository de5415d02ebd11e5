"""Diagnostic: a fixed sequence of solve variants (run under rocprofv3 --pmc; tools/pmc_variants.sh maps dispatches back)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
VARIANTS = [("mixed_queued", dict(precision="mixed"), 4096), ("mixed_plain", dict(precision="mixed", flags=1 | 8), 4096),
            ("f64_queued", dict(precision="f64"), 4096), ("mixed_queued_64k", dict(precision="mixed"), 65536)]
if __name__ == "__main__":
    for name, kw, B in VARIANTS:
        batch = mpcqp.synth.config3(B)
        sol = mpcqp.MPCBatch(N=10, **kw)
        dev = sol.upload(batch)
        for _ in range(3):
            out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        print(name, "solved", float((out["status"] == 1).float().mean()), flush=True)
