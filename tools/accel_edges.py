"""Developer check: odd settings of the acceleration (period 1 / 2 / 7 / none, blocks shorter than a period, restarts every iteration) on both
engines: finite answers, solved QPs within 1e-4 of the checker."""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import mpcqp
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
b = mpcqp.synth.config3(256)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
oeng = mpcqp.Engine(olib, olib.default_config(N=10, delta=0.03, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
ref = oeng.solve_batch_host(b["x0"], b["r"], b["contact"], b["xdes"], b["mu"])["u"].reshape(256, -1)
for N, flags in ((10, mpcqp.FLAG_POLISH), (10, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL)):
    for ac in (1, 2, 7, 1000, 5):
        for ce, fb, rs in ((3, 0, 0), (10, 0, 0), (100, 7, 0), (100, 0, 1), (100, 0, 7), (100, 60, 26)):
            sol = mpcqp.MPCBatch(N=N, precision="mixed", io_dtype="f64", accel=ac, check_every=ce, first_block=fb, accel_restart=rs, max_iter=400, flags=flags)
            dev = sol.upload(b)
            o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
            u = o["u"].cpu().numpy().reshape(256, -1); st = o["status"].cpu().numpy()
            assert np.all(np.isfinite(u)), (ac, ce, fb, rs)
            ok = st == 1
            err = (np.abs(u - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0))[ok].max() if ok.any() else 0.0
            assert err <= 1e-4, (ac, ce, fb, rs, err)
            print(f"engine {'stage' if flags & mpcqp.FLAG_STAGE_KERNEL else 'dense'} accel {ac:4d} check_every {ce:3d} first_block {fb:2d} restart {rs:2d}: solved {ok.mean():.3f} max err {err:.1e}", flush=True)
            del sol
print("edge cases ok")
