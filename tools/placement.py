"""Diagnostic (stamps build, plain form): where the hardware's dispatcher puts workgroup k of a one-wave-per-workgroup launch."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
from mpcqp import _capi
lib = _capi.Library(os.path.join(REPO, "mpc-for-dynamic-locomotion-in-the-mit-cheetah-3_amd", "csrc", "libmpcqp_stamps.so"))
_capi._product = lib
B = 4096
batch = mpcqp.synth.config3(B)
sol = mpcqp.MPCBatch(N=10, precision="mixed", flags=1 | 8)
dev = sol.upload(batch)
for _ in range(2):
    out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * B))()
lib.lib.mpcqp_debug_read_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.lib.mpcqp_debug_read_timeline(buf, B) == 0
t = np.array(list(buf), dtype=np.uint64).reshape(B, 3)
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
wave = hw & 0xf; simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print("k: xcc se sh cu simd wave")
for k in list(range(0, 40)) + list(range(248, 272)) + [512, 513, 1024, 1025, 2040, 2047]:
    print(k, xcc[k], se[k], sh[k], cu[k], simd[k], wave[k])
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
first = key[:2048]
ids, cnt = np.unique(first, return_counts=True)
print("distinct SIMDs among the first 2048 workgroups:", len(ids), "waves per SIMD min/max", cnt.min(), cnt.max())
# partner structure: for k < 2048, which other k' < 2048 shares its SIMD?
partner = {}
for k in range(2048):
    partner.setdefault(int(first[k]), []).append(k)
d = [abs(v[1] - v[0]) for v in partner.values() if len(v) == 2]
print("index distance between the two first-fill workgroups of a SIMD: ", np.unique(d, return_counts=True))
