import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import mpcqp
for name, mk, N in (("config4 B=65536", lambda: mpcqp.synth.config4(65536), 10), ("config5 B=4096 N=20", lambda: mpcqp.synth.config5(4096), 20)):
    b = mk()
    sol = mpcqp.MPCBatch(N=N, precision="mixed")
    dev = sol.upload(b)
    for _ in range(2):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = out["status"].cpu().numpy()
    print(f"{name}: {dt*1e3:.2f} ms  {len(st)/dt:,.0f} QP/s  solved {np.mean((st==1)|(st==2)):.4f}")
