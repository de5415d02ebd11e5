"""Diagnostic: per-QP measured work (ADMM iterations, polish steps) of the wrench engine on batches of several seeds, saved with
the operator tuples' support-pattern features -> gpurun_out/cost_data.npz (analysed offline for the dispatch-order model)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
G, M = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
recs = {}
for seed in (20250809, 1, 2, 3, 4, 5):
    b = mpcqp.synth.make_batch(4096, 10, 0.03, seed, G, M)
    sol = mpcqp.MPCBatch(N=10, precision="mixed")
    dev = sol.upload(b)
    o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
    recs[f"iters_{seed}"] = o["iters"].cpu().numpy()
    recs[f"r_{seed}"] = b["r"].astype(np.float32); recs[f"contact_{seed}"] = b["contact"]; recs[f"mu_{seed}"] = b["mu"].astype(np.float32)
    recs[f"x0_{seed}"] = b["x0"].astype(np.float32); recs[f"xdes_{seed}"] = b["xdes"].astype(np.float32)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(REPO, "gpurun_out", "cost_data.npz"), **recs)
print("saved")
