"""Developer study (stamps build): what ends a launch of 4096 QPs?  Per seed: launch time, per-QP durations and end times,
how many QPs are still running in the last 50 / 40 / 30 / 20 / 10 % of the launch and what they are (gait, mu, rounds, polish
steps), and how much of the launch is the single longest QP."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from mpcqp import _capi
lib = _capi.Library(os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc", os.environ.get("TL_LIB", "libmpcqp_stamps.so")))   # TL_LIB=libmpcqp_timeline.so: the timeline without the phase stamps
_capi._product = lib
B = int(os.environ.get("TL_B", "4096"))
TICK = 100.0   # s_memtime ticks per us (100 MHz constant clock on gfx950)
names = ["trot", "pronk", "amble", "gallop"]
allg = ("trot", "pronk", "amble", "gallop")
lib.lib.mpcqp_debug_read_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int64]
for seed in [int(x) for x in os.environ.get('TL_SEEDS', '20250809,1,2,3,4').split(',')]:
    batch = mpcqp.synth.make_batch(B, 10, 0.03, seed, allg, (0.3, 0.5, 0.7, 1.0))
    sol = mpcqp.MPCBatch(N=10, precision="mixed", accel=int(os.environ.get("TL_ACCEL", "0")), first_block=int(os.environ.get("TL_FIRST", "0")), hard_block_x10=int(os.environ.get("TL_HARD", "0")))
    dev = sol.upload(batch)
    for _ in range(3):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    ms = sol.last_kernel_ms()
    buf = (ctypes.c_ulonglong * (3 * B))()
    assert lib.lib.mpcqp_debug_read_timeline(buf, B) == 0
    t = np.array(list(buf), dtype=np.uint64).reshape(B, 3)
    t0 = t[:, 0].astype(np.float64); t1 = t[:, 1].astype(np.float64)
    xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
    admm = it % 1000; ps = it // 1000
    dur = (t1 - t0)
    # the tick unit: calibrate on the launch itself (longest span on any XCD ~ kernel time)
    tick_per_us = TICK                                   # s_memrealtime: one clock for the whole device
    end = (t1 - t0.min()) / tick_per_us; start = (t0 - t0.min()) / tick_per_us
    dur_us = dur / tick_per_us
    T = end.max()
    print(f"seed {seed}: kernel {ms:.3f} ms (stamps build), ticks/us {tick_per_us:.1f}, solved {np.mean((st == 1) | (st == 2)):.4f}, "
          f"mean iters {admm.mean():.1f}, mean polish {ps.mean():.2f}, work sum {dur_us.sum() / 2048:.1f} us per slot")
    print(f"   per-QP us: p50 {np.median(dur_us):.0f} p90 {np.percentile(dur_us, 90):.0f} p99 {np.percentile(dur_us, 99):.0f} max {dur_us.max():.0f}; launch span {T:.0f} us")
    for frac in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
        live = (end > frac * T) & (start <= frac * T)
        print(f"   at {frac:.0%} of the launch ({frac * T:4.0f} us): {live.sum():4d} QPs running, {(start > frac * T).sum():4d} not yet started")
    order = np.argsort(-end)[:12]
    print("   last finishers: (end us, start us, dur us, admm iters, polish steps, gait, mu)")
    for i in order:
        print(f"      {end[i]:6.0f} {start[i]:6.0f} {dur_us[i]:6.0f} {admm[i]:4d} {ps[i]:3d} {names[batch['gait_ids'][i]]:7s} {batch['mu'][i]:.1f}")
    # what would the launch be if every QP longer than X were 2x faster?
    late = np.argsort(-dur_us)[:64]
    print(f"   64 longest QPs: mean dur {dur_us[late].mean():.0f} us, mean start {start[late].mean():.0f} us, max start {start[late].max():.0f} us")
    hist_r = np.bincount(np.minimum((admm + 99) // 100, 6), minlength=7)
    print("   ADMM iterations /100 histogram:", hist_r.tolist(), " polish steps hist:", np.bincount(np.minimum(ps, 15)).tolist())
    del sol
