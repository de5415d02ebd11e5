"""Parity campaign of the shipped engine against the CPU checker (oracle/libmpcqp_oracle.so; test infrastructure, never on the product
path) on batches the test suite does not hold: other seeds, all three engines' horizons.  Prints solved fractions and the largest
relative force error of the solved QPs (tolerance of the tests: 1e-4).
usage: python tools/parity_campaign.py"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
cases = [(10, 0.03, 2048, sd, "mixed") for sd in (11, 12, 13, 14, 15, 16)] + [(10, 0.03, 1024, 17, "f64")] + \
        [(20, 0.03, 512, sd, "mixed") for sd in (21, 22, 23)] + [(60, 0.01, 96, sd, "mixed") for sd in (31, 32)] + [(33, 0.02, 128, 41, "mixed")]
worst = 0.0
for N, delta, B, seed, prec in cases:
    b = mpcqp.synth.make_batch(B, N, delta, seed, G, M)
    sol = mpcqp.MPCBatch(N=N, delta=delta, io_dtype="f64", precision=prec)
    dev = sol.upload(b)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
    u = out["u"].cpu().numpy().reshape(B, -1); st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    t0 = time.time()
    oeng = mpcqp.Engine(olib, olib.default_config(N=N, delta=delta, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, polish_max=30))
    ref = oeng.solve_batch_host(b["x0"], b["r"], b["contact"], b["xdes"], b["mu"])
    ur = ref["u"].reshape(B, -1); rs = ref["status"]
    ok = ((st == 1) | (st == 2)) & ((rs == 1) | (rs == 2))
    err = np.abs(u - ur).max(axis=1) / np.maximum(np.abs(ur).max(axis=1), 1.0)
    worst = max(worst, float(err[ok].max()))
    print(f"N={N:2d} delta={delta} B={B:5d} seed {seed:3d} {prec:5s}: engine solved {np.mean((st == 1) | (st == 2)):.4f} (checker {np.mean((rs == 1) | (rs == 2)):.4f}), "
          f"max rel force error of the solved {err[ok].max():.2e}, iterations mean {(it % 1000).mean():.0f} max {(it % 1000).max()}, polish steps mean {(it // 1000).mean():.2f}; checker {time.time() - t0:.1f} s", flush=True)
    del sol
print(f"worst relative force error over the campaign: {worst:.2e} (tolerance 1e-4)")
assert worst <= 1e-4
