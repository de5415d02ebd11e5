"""Developer study: how many non-improving polish steps should a round tolerate (MpcQpConfig.polish_patience) and how long may a
round be (polish_max), now that a step that follows another only updates S^-1?  Bench seed + four other seeds, B = 4096 and 65 536."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
allg = ("trot", "pronk", "amble", "gallop")
seeds = (20250809, 1, 2, 3, 4, 5, 6)
batches = {sd: mpcqp.synth.make_batch(4096, 10, 0.03, sd, allg, (0.3, 0.5, 0.7, 1.0)) for sd in seeds}
big = mpcqp.synth.config4(65536)
combos = [(0, 400), (2, 400), (3, 400), (4, 400), (0, 500), (3, 500), (0, 340), (3, 340)]   # last-round patience (0: unlimited), max_iter
if len(sys.argv) > 1:
    combos = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]]
for lp, mi in combos:
    sol = mpcqp.MPCBatch(N=10, precision="mixed", polish_last_patience=lp, max_iter=mi, flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING)
    line = f"last-round patience {lp} max_iter {mi}:"
    rates = []
    for sd in seeds:
        dev = sol.upload(batches[sd])
        for _ in range(3):
            out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ms = []
        for _ in range(15):
            ev[0].record(); out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); ev[1].record()
            torch.cuda.synchronize(); ms.append(ev[0].elapsed_time(ev[1]))
        it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
        m = float(np.median(ms)); rates.append(4096 / m / 1e3)
        line += f" | {sd % 100000}: {m * 1e3:.0f}us uns {int((st != 1).sum())} it {(it % 1000).mean():.0f} ps {(it // 1000).mean():.2f}"
    dev = sol.upload(big)
    for _ in range(2):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(5):
        ev[0].record(); out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); ev[1].record()
        torch.cuda.synchronize(); ms.append(ev[0].elapsed_time(ev[1]))
    st = out["status"].cpu().numpy()
    print(line + f" || mean of other seeds {np.mean(rates[1:]):.2f} M, bench seed {rates[0]:.2f} M | B=65536: {65536 / np.median(ms) / 1e3:.2f} M uns {int((st != 1).sum())}", flush=True)
    del sol
