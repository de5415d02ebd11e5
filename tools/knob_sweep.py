"""Developer tool: throughput / solved fraction / accuracy over solver knobs on the bench workload."""
import itertools, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp
B = 4096
batch = mpcqp.synth.config3(B)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
oeng = mpcqp.Engine(olib, olib.default_config(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000, polish_max=30))
sub = {k: batch[k][:512] for k in ("x0", "r", "contact", "xdes", "mu")}
ref = oeng.solve_batch_host(sub["x0"], sub["r"], sub["contact"], sub["xdes"], sub["mu"], want_X=False)["u"].reshape(512, -1)
CES = tuple(int(x) for x in os.environ.get('KNOB_CE', '40,50,60,80,100,130').split(','))
RHOS = tuple(float(x) for x in os.environ.get('KNOB_RHO', '1.0,2.0').split(','))
PMS = tuple(int(x) for x in os.environ.get('KNOB_PM', '3,4').split(','))
for ce, rho, pm in itertools.product(CES, RHOS, PMS):
    sol = mpcqp.MPCBatch(N=10, precision="mixed", check_every=ce, max_iter=int(os.environ.get("KNOB_MIF", "4")) * ce, rho=rho, polish_max=pm)
    dev = sol.upload(batch)
    for _ in range(2):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    u = out["u"].cpu().numpy().astype(np.float64).reshape(B, -1)[:512]
    ok = ((st == 1) | (st == 2))
    e = np.abs(u - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)
    print(f"check_every {ce:3d} rho {rho:4.1f} polish_max {pm}: {ms:6.3f} ms  {B / ms * 1e3:9,.0f} QP/s  solved {ok.mean():.4f}  "
          f"admm {np.mean(it % 1000):6.1f}  polish {np.mean(it // 1000):4.2f}  maxerr(solved) {e[ok[:512]].max():.1e}", flush=True)
