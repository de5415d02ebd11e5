import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import mpcqp
b = mpcqp.synth.config3(4096)
sol = mpcqp.MPCBatch(N=10, precision="mixed")
dev = sol.upload(b)
out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
torch.cuda.synchronize()
np.save("/root/repo/gpurun_out/iters_c3.npy", out["iters"].cpu().numpy())
