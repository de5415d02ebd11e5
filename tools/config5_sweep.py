"""BASELINE config 5 (B = 4096, N = 20, seed 20250811): fp32-tile vs all-fp64 arithmetic over the iteration / tolerance knobs,
against the fp64 CPU oracle.  Writes profiles/r02_config5_sweep.json.

Two families of cells ("Config 5"):
  admm_only   polish off, OSQP's termination test: iteration cap K in {25..400} x eps in {1e-3..1e-6}  (what the reference runs:
              OSQP with default tolerances, src/mpc.py:51-55)
  polish      the engine's mode: ADMM blocks of K iterations + active-set polish, cap 4 K
Per cell: QP solves/s (B / event time of one solve_batch), solved fraction, max / median relative GRF error vs the oracle over
ALL QPs and over the solved ones (error definition: SURVEY.md 8(c)).  The oracle runs on the first `NREF` QPs."""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from gpu_check import oracle_solve, relerr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NREF = min(B, 512)
N = 20
batch = mpcqp.synth.config5(B)
sub = {k: (v[:NREF] if isinstance(v, np.ndarray) and len(v) == B else v) for k, v in batch.items()}
t0 = time.time(); ref = oracle_solve(sub, N, 0.03); print(f"oracle on {NREF} QPs: {time.time() - t0:.1f} s", flush=True)
cells = []


def run(label, precision, **kw):
    sol = mpcqp.MPCBatch(N=N, delta=0.03, io_dtype="f32", precision=precision, **kw)
    dev = sol.upload(batch)
    ms = []
    for _ in range(3):
        o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize(); ms.append(sol.last_kernel_ms())
    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
    u = o["u"].cpu().numpy().astype(np.float64)[:NREF]
    e = relerr(u, ref["u"]); ok = ((st == 1) | (st == 2))
    okr = ok[:NREF]
    rec = dict(label, precision=precision, qp_per_s=B / (min(ms) * 1e-3), ms=min(ms), solved_fraction=float(ok.mean()),
               err_max_all=float(e.max()), err_median_all=float(np.median(e)),
               err_max_solved=float(e[okr].max()) if okr.any() else None, err_median_solved=float(np.median(e[okr])) if okr.any() else None,
               admm_iters_mean=float((it % 1000).mean()), polish_steps_mean=float((it // 1000).mean()))
    cells.append(rec)
    print(json.dumps(rec), flush=True)


for precision in ("mixed", "f64"):
    for K in (25, 50, 100, 200, 400):
        run({"mode": "polish", "K": K}, precision, check_every=K, max_iter=4 * K)
    for K in (25, 50, 100, 200, 400):
        for eps in (1e-3, 1e-4, 1e-5, 1e-6):
            run({"mode": "admm_only", "K": K, "eps": eps}, precision, flags=0, check_every=K, max_iter=K, eps_abs=eps, eps_rel=eps)
out = {"workload": f"BASELINE configs[4]: B={B}, N=20, delta=0.03, mixed gaits + mu sweep, seed 20250811, alpha=1e-2; oracle on the first {NREF} QPs",
       "engine": "mpcqp_wrench_solve (N = 20: one QP per 4-wave workgroup); precision mixed = fp32 ADMM tiles + fp64 polish / residuals, f64 = all fp64",
       "error": "per-QP |u - u_oracle|_inf / max(|u_oracle|_inf, 1) over the whole horizon", "cells": cells}
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "profiles", "r02_config5_sweep.json"), "w"), indent=1)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "r02_config5_sweep.json"), "w"), indent=1)
