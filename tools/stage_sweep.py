"""Developer study: ADMM block length of the stage-wise engine at the reference's horizon (N = 60): synthetic mixed batch + the 1000 logged ticks."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
from test_gpu_reference_horizon import logged_run_inputs, gpu_solve
g = {k: np.load(os.path.join(REPO, "tests", "golden", k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}
logged = logged_run_inputs(g, 60, np.arange(1000))
synth = mpcqp.synth.make_batch(1024, 60, 0.01, 11, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
for ce, mi, pm in ((200, 2400, 8), (150, 2400, 8), (120, 2400, 8), (100, 2400, 8), (80, 2400, 8), (60, 2400, 8), (150, 2400, 4)):
    line = f"check_every {ce} max_iter {mi} polish_max {pm}:"
    for name, b in (("logged", logged), ("synthetic", synth)):
        o = gpu_solve(b, 60, 0.01, "mixed", alpha=1e-2, check_every=ce, max_iter=mi, polish_max=pm)
        it = o["iters"] % 1000; ps = o["iters"] // 1000
        line += f" | {name}: {o['ms']:.1f} ms = {len(it) / o['ms']:.1f} k QP/s, solved {np.mean(o['status'] == 1):.4f}, iters mean {it.mean():.0f} max {it.max()}, polish mean {ps.mean():.2f} max {ps.max()}"
    print(line, flush=True)
