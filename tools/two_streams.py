"""Developer tool: two independent batches of 4096 QPs on two handles / two HIP streams (a handle serves one stream at a time,
include/mpcqp.h).  A launch of 4096 is as long as its longest QPs; with a second batch in flight on another
stream the tail of one launch is filled by the head of the next, and the rate approaches the large-batch figure.  Secondary
figure only: bench.py's `value` is one batch per step on one stream."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp

B, N, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 10, 50
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
flags = mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING
sols = [mpcqp.MPCBatch(N=N, delta=0.03, io_dtype="f32", precision="mixed", flags=flags) for _ in range(2)]
devs = [sols[i].upload(mpcqp.synth.make_batch(B, N, 0.03, 20250809 + i, G, M)) for i in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def solve(i, stream=None):
    d = devs[i]
    return sols[i].solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"], want_X=False, stream=stream)


def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for i in range(2):
            o = solve(i, streams[i] if two else None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * STEPS)


for two in (False, True, False, True):
    run(two)                      # warm-up of the form
    t = run(two)
    outs = [solve(i, streams[i] if two else None) for i in range(2)]
    torch.cuda.synchronize()
    ok = np.mean([float(((o["status"] == 1) | (o["status"] == 2)).float().mean().item()) for o in outs])
    print(f"B={B} x 2 batches, {'two streams' if two else 'one stream '}: {t * 1e3:.3f} ms per batch = {B / t / 1e6:.2f} M QP/s  solved {ok:.4f}", flush=True)
