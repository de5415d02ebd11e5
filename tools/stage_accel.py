"""Developer study: Anderson acceleration in the stage-wise engine at the reference's horizon (N = 60, delta = 0.01): the 1000 logged
ticks and a synthetic mixed batch of 1024, accel off / on x block lengths; answers compared with the plain engine's."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
from test_gpu_reference_horizon import logged_run_inputs, gpu_solve
g = {k: np.load(os.path.join(REPO, "tests", "golden", k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}
logged = logged_run_inputs(g, 60, np.arange(1000))
synth = mpcqp.synth.make_batch(1024, 60, 0.01, 11, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
ref = {}
combos = [(-1, 120, 0), (5, 120, 0), (5, 100, 0), (5, 80, 0), (5, 120, 60), (5, 100, 50), (8, 120, 0), (3, 120, 0)]
if len(sys.argv) > 1:
    combos = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for combo in combos:
    ac, ce, fb = combo[:3]
    extra = {}
    if os.environ.get('AS_RESTART'): extra['accel_restart'] = int(os.environ['AS_RESTART'])
    if len(combo) > 3 and combo[3] > 0: extra['adapt_thr'] = float(combo[3])
    if len(combo) > 4 and combo[4] > 0: extra['hard_block_x10'] = int(combo[4])
    line = f"accel {ac:2d} check_every {ce} first_block {fb} {extra}:"
    for prec in ("mixed", "f64"):
        for name, b in (("logged", logged), ("synthetic", synth)):
            o = gpu_solve(b, 60, 0.01, prec, alpha=1e-2, check_every=ce, max_iter=2400, polish_max=8, accel=ac, first_block=fb, **extra)
            it = o["iters"] % 1000; ps = o["iters"] // 1000
            u = o["u"].reshape(len(it), -1)
            key = (prec, name)
            d = 0.0
            if key in ref:
                both = (o["status"] == 1) & ref[key][1]
                d = float((np.abs(u - ref[key][0]).max(axis=1) / np.maximum(np.abs(ref[key][0]).max(axis=1), 1.0))[both].max())
            else:
                ref[key] = (u, o["status"] == 1)
            line += f" | {prec} {name}: {o['ms']:.1f} ms = {len(it) / o['ms']:.1f} k QP/s, uns {int(np.sum(o['status'] != 1))}, it {it.mean():.0f}/{it.max()}, ps {ps.mean():.2f}/{ps.max()} d {d:.0e}"
    print(line, flush=True)
