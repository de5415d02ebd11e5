"""Diagnostic (stamps build): the hardest QPs of the bench batch, each solved ALONE (B = 1) -- duration and phase shares."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from mpcqp import _capi
prod = _capi.product() if hasattr(_capi, "product") else None
_seed = os.environ.get("HARD_SEED")
_gaits = tuple(os.environ.get("HARD_GAITS", "trot,pronk,amble,gallop").split(","))
batch = mpcqp.synth.config3(4096) if _seed is None else mpcqp.synth.make_batch(4096, 10, 0.03, int(_seed), _gaits, (0.3, 0.5, 0.7, 1.0))
sol = mpcqp.MPCBatch(N=10, precision="mixed")
dev = sol.upload(batch)
o = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"]); torch.cuda.synchronize()
it = o["iters"].cpu().numpy()
cost = (it % 1000) + 60 * (it // 1000)
order = np.argsort(-cost)[:6]
print("hardest:", [(int(b), int(it[b])) for b in order], "contact stance legs/stage:", [batch["contact"][b].sum(axis=1).tolist() for b in order[:2]], "mu", [float(batch["mu"][b]) for b in order])
names = ["setup", "admm-E", "admm-tile-init", "admm-sweep", "admm-iters", "rho-check", "polish-solve+kkt", "polish-publish", "output",
         "polish-E", "polish-tile-init", "polish-sweep"]
lib = _capi.Library(os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc", "libmpcqp_stamps.so"))
for which, L in (("product", None), ("stamps", lib)):
    if L is not None: _capi._product = L
    for b in order[:3]:
        one = {k: (v[b:b + 1] if isinstance(v, np.ndarray) and len(v) == 4096 else v) for k, v in batch.items()}
        s1 = mpcqp.MPCBatch(N=10, precision="mixed")
        d1 = s1.upload(one)
        ms = []
        for _ in range(4):
            o1 = s1.solve_batch(d1["x0"], d1["r"], d1["contact"], d1["xdes"], d1["mu"]); torch.cuda.synchronize(); ms.append(s1.last_kernel_ms())
        line = f"{which} QP {int(b)} alone: {min(ms) * 1e3:.0f} us iters {int(o1['iters'][0])}"
        if L is not None:
            buf = (ctypes.c_ulonglong * 32)()
            L.lib.mpcqp_debug_read_stamps(buf)
            v = np.array(list(buf), dtype=np.float64)
            line += "  cycles: " + ", ".join(f"{n} {v[i]:.0f} ({v[16 + i]:.0f}x)" for i, n in enumerate(names) if v[i] > 0)
        print(line, flush=True)
        if L is not None:
            dbg = (ctypes.c_double * 2048)()
            if L.lib.mpcqp_debug_read_wdbg(dbg) == 0:
                d = np.array(list(dbg))[1300:1300 + 8 * 80].reshape(80, 8)
                nst = int(o1["iters"][0]) // 1000
                for r in d[:nst]:
                    print(f"      kind/round {int(r[0]):3d} step {int(r[1])} stat {r[2]:.2e} prim {r[3]:.2e} dual {r[4]:.2e} rho {r[5]:.3g} iters {int(r[6])} ok {int(r[7])}")
