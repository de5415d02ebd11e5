import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import mpcqp, qp_spec as S
G = os.path.join(REPO, "tests", "golden")
q = np.load(os.path.join(G, "qp_inputs.npz")); opt = np.load(os.path.join(G, "qp_optima.npz"))
bg = {"x0": q["N10_x0"], "r": q["N10_r"], "contact": q["N10_contact"], "xdes": q["N10_xdes"], "mu": np.full(10, float(q["mu"]))}
cfgg = S.QPConfig(N=10, delta=float(q["delta"]), alpha=0.0)
B = 256
batch = mpcqp.synth.config3(B)
olib = mpcqp.Library(os.path.join(REPO, "oracle", "libmpcqp_oracle.so"))
o0 = mpcqp.Engine(olib, olib.default_config(alpha=0.0, rho=0.3, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, polish_max=30))
r0 = o0.solve_batch_host(batch["x0"], batch["r"], batch["contact"], batch["xdes"], batch["mu"])
cfg0 = S.QPConfig(N=10, delta=0.03, alpha=0.0)
W0 = np.array([S.net_wrench(r0["u"][i], batch["r"][i], batch["contact"][i], cfg0) for i in range(B)])
for floor in ("1e-5", "3e-6", "1e-6"):
    sol = mpcqp.MPCBatch(N=10, delta=float(q["delta"]), precision="mixed", io_dtype="f64", alpha=0.0, max_iter=800, alpha_floor=float(floor))
    d = sol.upload(bg); o = sol.solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"], want_X=True); torch.cuda.synchronize()
    st = o["status"].cpu().numpy(); u = o["u"].cpu().numpy(); X = o["X"].cpu().numpy()
    dJ = [abs(S.objective(X[i], u[i], bg["xdes"][i], cfgg) - opt["N10_a0_J"][i]) / max(1, abs(opt["N10_a0_J"][i])) for i in range(10)]
    dW = [np.abs(S.net_wrench(u[i], bg["r"][i], bg["contact"][i], cfgg) - opt["N10_a0_wrench"][i]).max() / max(1, np.abs(opt["N10_a0_wrench"][i]).max()) for i in range(10)]
    dX = np.abs(X - opt["N10_a0_X"]).reshape(10, -1).max(axis=1)
    print(f"floor {floor} golden ticks: status {st.tolist()} dJ max {max(dJ):.1e} dX max {dX.max():.1e} dW max {max(dW):.1e}", flush=True)
    sol = mpcqp.MPCBatch(N=10, precision="mixed", io_dtype="f64", alpha=0.0, max_iter=800)
    d = sol.upload(batch); o = sol.solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"], want_X=True); torch.cuda.synchronize()
    st = o["status"].cpu().numpy(); u = o["u"].cpu().numpy(); X = o["X"].cpu().numpy(); it = o["iters"].cpu().numpy()
    ok = (st == 1) & (r0["status"] != 3)
    dW = np.array([np.abs(S.net_wrench(u[i], batch["r"][i], batch["contact"][i], cfg0) - W0[i]).max() / max(1, np.abs(W0[i]).max()) for i in range(B)])
    dX = np.abs(X - r0["X"]).reshape(B, -1).max(axis=1)
    print(f"floor {floor} config3: solved {(st == 1).mean():.3f} dX max {dX[ok].max():.1e} dW max {dW[ok].max():.1e} polish {np.mean(it // 1000):.1f} {sol.last_kernel_ms():.2f} ms", flush=True)
