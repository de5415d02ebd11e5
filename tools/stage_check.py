"""Developer check of the stage-wise engine (mpcqp_stage.h): against the oracle and the dense wrench-space engine at N = 10 / 20
(MPCQP_FLAG_STAGE_KERNEL), and against the committed optima of the reference's own N = 60 ticks (tests/golden)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import mpcqp

def rel_err(u, ur):
    u = u.reshape(len(u), -1); ur = ur.reshape(len(ur), -1)
    return np.abs(u - ur).max(axis=1) / np.maximum(np.abs(ur).max(axis=1), 1.0)

def gpu(b, N, delta, precision, flags, **kw):
    sol = mpcqp.MPCBatch(N=N, delta=delta, io_dtype="f64", precision=precision, flags=flags, **kw)
    dev = sol.upload(b)
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
    torch.cuda.synchronize()
    t0 = time.time()
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
    res["ms"] = sol.last_kernel_ms()
    return res

olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
which = sys.argv[1:] or ["n10", "n60", "rate"]
if "n10" in which:
    for N, b in ((10, mpcqp.synth.config3(64)), (20, mpcqp.synth.config5(32))):
        oeng = mpcqp.Engine(olib, olib.default_config(N=N, delta=0.03, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
        ref = oeng.solve_batch_host(b["x0"], b["r"], b["contact"], b["xdes"], b["mu"])
        dense = gpu(b, N, 0.03, "mixed", mpcqp.FLAG_POLISH)
        for prec in ("f64", "mixed"):
            o = gpu(b, N, 0.03, prec, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL)
            ok = o["status"] == 1
            e = rel_err(o["u"], ref["u"]); eX = np.abs(o["X"] - ref["X"]).reshape(len(e), -1).max(axis=1)
            print(f"N={N} stage {prec}: solved {ok.sum()}/{len(ok)} statuses {np.bincount(o['status'] + 1).tolist()} max rel err u (solved) {e[ok].max() if ok.any() else float('nan'):.2e} "
                  f"X {eX[ok].max() if ok.any() else float('nan'):.2e}; worst overall {e.max():.2e}; iters mean {np.mean(o['iters'] % 1000):.0f} polish {np.mean(o['iters'] // 1000):.2f} "
                  f"(dense engine: {np.mean(dense['iters'] % 1000):.0f} / {np.mean(dense['iters'] // 1000):.2f}); kernel {o['ms']:.2f} ms (dense {dense['ms']:.3f})", flush=True)
if "n60" in which:
    q = np.load(os.path.join(REPO, "tests", "golden", "qp_inputs.npz")); opt = np.load(os.path.join(REPO, "tests", "golden", "qp_optima.npz"))
    N = 60
    b = {"x0": q[f"N{N}_x0"], "r": q[f"N{N}_r"], "contact": q[f"N{N}_contact"], "xdes": q[f"N{N}_xdes"], "mu": np.full(len(q["ticks"]), float(q["mu"]))}
    for prec in ("f64", "mixed"):
        for alpha, tag in ((1e-2, "a1e-2"), (1e-4, "a1e-4")):
            o = gpu(b, N, float(q["delta"]), prec, mpcqp.FLAG_POLISH, alpha=alpha)
            ok = o["status"] == 1
            e = rel_err(o["u"], opt[f"N{N}_{tag}_u"]); eX = np.abs(o["X"] - opt[f"N{N}_{tag}_X"]).reshape(len(e), -1).max(axis=1)
            print(f"N=60 golden ticks {prec} alpha {alpha}: status {o['status'].tolist()} iters {o['iters'].tolist()} rel err u {np.array2string(e, precision=1)} X {eX.max():.2e} kernel {o['ms']:.2f} ms", flush=True)
if "rate" in which:
    for B in (256, 1024):
        b = mpcqp.synth.make_batch(B, 60, 0.01, 11, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
        for prec in ("mixed", "f64"):
            o = gpu(b, 60, 0.01, prec, mpcqp.FLAG_POLISH)
            it = o["iters"] % 1000 + 0; ps = o["iters"] // 1000
            print(f"N=60 B={B} {prec}: {o['ms']:.1f} ms = {B / o['ms']:.1f} k QP/s, solved {np.mean(o['status'] == 1):.3f}, iters mean {np.mean(it):.0f} max {it.max()} hist(/100) {np.bincount(it // 100).tolist()} "
                  f"polish mean {np.mean(ps):.1f} max {ps.max()}; unsolved: {[(int(i), int(o['iters'][i]), float(b['mu'][i]), int(b['gait_ids'][i])) for i in np.where(o['status'] != 1)[0][:6]]}", flush=True)
if "prof" in which:
    q = np.load(os.path.join(REPO, "tests", "golden", "qp_inputs.npz"))
    for N in (10, 20, 60):
        b = {"x0": q[f"N{N}_x0"], "r": q[f"N{N}_r"], "contact": q[f"N{N}_contact"], "xdes": q[f"N{N}_xdes"], "mu": np.full(len(q["ticks"]), float(q["mu"]))}
        for prec in ("mixed", "f64"):
            ms = {}
            for K in (50, 100, 200):
                o = gpu(b, N, float(q["delta"]), prec, mpcqp.FLAG_STAGE_KERNEL, max_iter=K, check_every=K, eps_abs=0.0, eps_rel=0.0)
                ms[K] = o["ms"]
            per_it = (ms[200] - ms[100]) / 100 * 1e3
            print(f"N={N} {prec} ADMM only: K=50/100/200 -> {ms[50]:.3f} / {ms[100]:.3f} / {ms[200]:.3f} ms; {per_it:.2f} us per iteration, fixed part {ms[100] - 0.1 * per_it:.3f} ms", flush=True)
            o = gpu(b, N, float(q["delta"]), prec, mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL, max_iter=25, check_every=25, polish_max=1)
            print(f"      25 iterations + 1 polish step: {o['ms']:.3f} ms (status {o['status'][:4].tolist()}, iters {o['iters'][:4].tolist()})", flush=True)
