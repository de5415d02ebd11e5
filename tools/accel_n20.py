"""Developer study: Anderson acceleration at horizon 20 (config 5: B = 4096, N = 20; history parked in LDS between extrapolations):
accel off / on x block lengths; answers compared with the plain engine's.  usage: python tools/accel_n20.py [accel,check_every,first_block,hard_x10 ...]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
batches = {"config 5": mpcqp.synth.config5(4096), "seed 1": mpcqp.synth.make_batch(4096, 20, 0.03, 1, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))}
combos = [(-1, 200, 0, 0), (5, 200, 0, 0), (5, 200, 120, 0), (5, 200, 100, 0), (5, 160, 0, 0), (5, 200, 120, 10), (5, 200, 100, 10)]
if len(sys.argv) > 1:
    combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
ref = {}
for combo in combos:
    ac, ce, fb, hx = combo[:4]
    extra = {'adapt_thr': float(combo[4])} if len(combo) > 4 and combo[4] > 0 else {}
    if os.environ.get('AS_RESTART'): extra['accel_restart'] = int(os.environ['AS_RESTART'])
    line = f"accel {ac:2d} check_every {ce} first_block {fb:3d} hard_x10 {hx:2d} {extra}:"
    for name, b in batches.items():
        sol = mpcqp.MPCBatch(N=20, precision="mixed", accel=ac, check_every=ce, first_block=fb, hard_block_x10=hx, max_iter=800, flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING, **extra)
        dev = sol.upload(b)
        for _ in range(2):
            out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ms = []
        for _ in range(6):
            ev[0].record(); out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); ev[1].record()
            torch.cuda.synchronize(); ms.append(ev[0].elapsed_time(ev[1]))
        it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy(); u = out["u"].cpu().numpy().astype(np.float64).reshape(4096, -1)
        d = 0.0
        if name in ref:
            both = (st == 1) & ref[name][1]
            d = float((np.abs(u - ref[name][0]).max(axis=1) / np.maximum(np.abs(ref[name][0]).max(axis=1), 1.0))[both].max())
        else:
            ref[name] = (u, st == 1)
        m = float(np.median(ms))
        line += f" | {name}: {m:.2f} ms = {4096 / m / 1e3:.3f} M QP/s uns {int((st != 1).sum())} it {(it % 1000).mean():.0f}/{(it % 1000).max()} ps {(it // 1000).mean():.2f}/{(it // 1000).max()} d {d:.0e}"
        del sol
    print(line, flush=True)
