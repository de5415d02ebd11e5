"""Developer study: does the best ADMM block length depend on the friction-demand class of a QP?"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
B = 4096
b = mpcqp.synth.config3(B)
r, c, mu = b["r"], b["contact"], b["mu"]
N = 10
score = np.zeros(B)
for i in range(B):
    for k in range(N):
        st = np.nonzero(c[i, k])[0]
        if len(st) == 2:
            ra, rb = r[i, k, st[0]], r[i, k, st[1]]
            d = abs(ra[0] * rb[1] - ra[1] * rb[0]) / max(np.hypot(*(ra[:2] - rb[:2])), 1e-9)
            score[i] = max(score[i], d / (-0.5 * (ra[2] + rb[2])))
        elif len(st) == 1:
            ra = r[i, k, st[0]]; score[i] = max(score[i], np.hypot(ra[0], ra[1]) / (-ra[2]))
score /= mu
bucket = np.clip((2 * score).astype(int), 0, 7)
# cost model (ticks, profiles/r01_h_phase_stamps.txt): ADMM block = desc+build+sweep 70k + 810/iteration; polish step 95k
for K in (50, 60, 70, 80, 100, 120):
    sol = mpcqp.MPCBatch(N=10, precision="mixed", check_every=K, max_iter=4 * K)
    dev = sol.upload(b)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
    admm = it % 1000; pol = it // 1000
    blocks = np.ceil(admm / K)
    cost = blocks * 70e3 + admm * 810.0 + pol * 95e3
    print(f"K={K:3d} solved {np.mean((st==1)|(st==2)):.4f} kernel {sol.last_kernel_ms():.3f} ms | " +
          " ".join(f"c{k}:{cost[bucket == k].mean() / 1e3:5.0f}k/{pol[bucket == k].mean():.2f}" for k in range(8)))
print("class sizes", np.bincount(bucket, minlength=8))
