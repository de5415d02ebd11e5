"""Developer study: warm-started solves of consecutive logged ticks on the stage-wise engine (N = 60), one robot, host-synchronised per solve:
iterations and wall time per tick against cold solves of the same ticks.  env AS_LIB: a variant build."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp
if os.environ.get("AS_LIB"):
    from mpcqp import _capi
    _capi._product = _capi.Library(os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc", os.environ["AS_LIB"]))
from test_gpu_reference_horizon import logged_run_inputs
g = {k: np.load(os.path.join(REPO, "tests", "golden", k + ".npz")) for k in ("ref_log", "planner_golden", "qp_inputs", "qp_optima")}
ticks = np.arange(100, 300)
run = logged_run_inputs(g, 60, ticks)
for warm in (False, True):
    ws = mpcqp.MPCBatch(N=60, delta=0.01, io_dtype="f64", precision="mixed", warm_start=warm, warm_shift=warm)
    its, ps, ms, bad = [], [], [], 0
    for rep in range(2):
        its, ps, ms, bad = [], [], [], 0
        for i in range(len(ticks)):
            one = {k: v[i:i + 1] for k, v in run.items()}
            d = ws.upload(one)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o = ws.solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"]); torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            it = int(o["iters"][0]); its.append(it % 1000); ps.append(it // 1000); bad += int(o["status"][0]) != 1
    print(f"{'warm (shifted previous tick)' if warm else 'cold':30s}: iterations mean {np.mean(its[1:]):.1f} max {max(its[1:])}, polish steps mean {np.mean(ps[1:]):.2f}, {np.mean(ms[1:]):.3f} ms per tick = {1e3 / np.mean(ms[1:]):.0f} solves/s, unsolved {bad}", flush=True)
