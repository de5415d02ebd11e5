import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
B = 4096
batch = mpcqp.synth.config3(B)
sol = mpcqp.MPCBatch(N=10, precision="mixed")
dev = sol.upload(batch)
out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
torch.cuda.synchronize()
it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy()
rounds = (it % 1000) // 100; ps = it // 1000
print("rounds hist", np.bincount(rounds), "polish steps hist", np.bincount(ps))
names = ["trot", "pronk", "amble", "gallop"]
for r in (2, 3, 4):
    m = rounds >= r
    print(f"needing >= {r} rounds: {m.sum()}  by gait", {names[g]: int((batch['gait_ids'][m] == g).sum()) for g in range(4)},
          "by mu", {float(mu): int((batch['mu'][m] == mu).sum()) for mu in (0.3, 0.5, 0.7, 1.0)})
print("all by gait", {names[g]: int((batch['gait_ids'] == g).sum()) for g in range(4)})
# number of stance legs at stage 0, and fraction of all-stance stages
c = batch["contact"]
ns = c.sum(axis=(1, 2))
for r in (1, 2, 3):
    m = rounds == r
    print(f"rounds == {r}: mean stance leg-stages {ns[m].mean():.1f}, first polish steps mean {ps[m].mean():.2f}")
np.save(os.path.join(REPO, "gpurun_out", "straggler_idx.npy"), np.where(rounds >= 2)[0])
