"""Developer check on a GPU box: wrench-space engine vs the CPU oracle and vs the stage-wise engine at the same horizon (config 3)."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp  # noqa: E402
from gpu_check import oracle_solve, relerr  # noqa: E402


def run(batch, ref, label, reps=5, **kw):
    sol = mpcqp.MPCBatch(N=10, delta=0.03, **kw)
    dev = sol.upload(batch)
    ms = []
    for _ in range(reps):
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=True)
        torch.cuda.synchronize()
        ms.append(sol.last_kernel_ms())
    u = o["u"].cpu().numpy().astype(np.float64); X = o["X"].cpu().numpy().astype(np.float64)
    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
    e = relerr(u, ref["u"]); eX = np.abs(X - ref["X"]).reshape(len(u), -1).max(axis=1)
    solved = (st == 1) | (st == 2)
    rec = {"ms_min": min(ms), "ms_all": [round(m, 3) for m in ms], "status_hist": np.bincount(st + 1, minlength=5).tolist(),
           "err_solved_max": float(e[solved].max()) if solved.any() else None, "err_all_max": float(e.max()),
           "X_err_solved_max": float(eX[solved].max()) if solved.any() else None,
           "admm_iters_mean": float((it % 1000).mean()), "polish_steps_mean": float((it // 1000).mean()),
           "qps_M": len(u) / min(ms) / 1e3}
    print(f"[{label}] {json.dumps(rec)}", flush=True)
    return rec


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    batch = mpcqp.synth.config3(B)
    t = time.time(); ref = oracle_solve(batch, 10, 0.03); print(f"oracle {time.time() - t:.1f}s status {np.bincount(ref['status'] + 1)}", flush=True)
    out = {}
    out["wrench/mixed/f32"] = run(batch, ref, "wrench mixed f32", io_dtype="f32", precision="mixed")
    out["wrench/mixed/f64"] = run(batch, ref, "wrench mixed f64", io_dtype="f64", precision="mixed")
    out["wrench/f64/f64"] = run(batch, ref, "wrench f64 f64", io_dtype="f64", precision="f64")
    out["stage/mixed/f32"] = run(batch, ref, "stage mixed f32", io_dtype="f32", precision="mixed", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_STAGE_KERNEL)
    for K in (30, 50, 70):
        out[f"wrench/K{K}"] = run(batch, ref, f"wrench mixed f32 K={K}", io_dtype="f32", precision="mixed", check_every=K)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "gpurun_out", "wrench_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
