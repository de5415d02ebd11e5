import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpcqp
B = 2048
N = 20
batch = mpcqp.synth.config5(B)
its = {}
for prec in ("mixed", "f64"):
    sol = mpcqp.MPCBatch(N=N, delta=0.03, precision=prec, flags=1 | 8)
    dev = sol.upload(batch)
    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
    it = o["iters"].cpu().numpy(); its[prec] = it
    c = collections.Counter((int(i % 1000), int(i // 1000)) for i in it)
    print(prec, sorted(c.items())[:14], flush=True)
pair = collections.Counter((int(a % 1000), int(a // 1000), int(b % 1000), int(b // 1000)) for a, b in zip(its["mixed"], its["f64"]))
print("mixed(it,ps) f64(it,ps):", pair.most_common(12))
