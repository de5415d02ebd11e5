"""Developer check on a GPU box: HIP engine vs the CPU oracle on synthetic batches; prints error / status stats."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp  # noqa: E402


def oracle_solve(batch, N, delta, **kw):
    lib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
    cfg = lib.default_config(N=N, delta=delta, eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30, **kw)
    eng = mpcqp.Engine(lib, cfg)
    return eng.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"])


def relerr(u, ur):
    u = u.reshape(len(u), -1); ur = ur.reshape(len(ur), -1)
    sc = np.maximum(np.abs(ur).max(axis=1), 1.0)
    return np.abs(u - ur).max(axis=1) / sc


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out = {}
    for name, mk, N in (("config3", mpcqp.synth.config3, 10), ("config5", mpcqp.synth.config5, 20)):
        batch = mk(B)
        t = time.time(); ref = oracle_solve(batch, N, 0.03); t_or = time.time() - t
        print(f"[{name}] oracle: {t_or:.1f}s status {np.bincount(ref['status'] + 1)}", flush=True)
        precs = ("mixed", "f32", "f64")
        for prec in precs:
            for io in ("f32", "f64"):
                try:
                    sol = mpcqp.MPCBatch(N=N, delta=0.03, io_dtype=io, precision=prec)
                    dev = sol.upload(batch)
                    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
                    torch.cuda.synchronize()
                    ms = sol.last_kernel_ms()
                    u = o["u"].cpu().numpy().astype(np.float64); X = o["X"].cpu().numpy().astype(np.float64)
                    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
                    e = relerr(u, ref["u"]); eX = np.abs(X - ref["X"]).max()
                    solved = (st == 1) | (st == 2)
                    rec = {"ms": ms, "status_hist": np.bincount(st + 1, minlength=5).tolist(),
                           "err_solved_max": float(e[solved].max()) if solved.any() else None,
                           "err_solved_med": float(np.median(e[solved])) if solved.any() else None,
                           "err_all_max": float(e.max()), "n_err_gt_1e-4": int((e > 1e-4).sum()),
                           "n_solved_err_gt_1e-4": int((e[solved] > 1e-4).sum()), "X_err_max": float(eX),
                           "admm_iters_mean": float((it % 1000).mean()), "polish_steps_mean": float((it // 1000).mean())}
                    out[f"{name}/{prec}/{io}"] = rec
                    print(f"[{name}] prec={prec} io={io}: {json.dumps(rec)}", flush=True)
                except Exception as ex:  # noqa: BLE001
                    print(f"[{name}] prec={prec} io={io}: FAILED {type(ex).__name__}: {ex}", flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "gpu_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
