import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import mpcqp
b = mpcqp.synth.config5(1024)
for K, mi in ((100, 400), (150, 600), (200, 800), (300, 900), (400, 1200)):
    sol = mpcqp.MPCBatch(N=20, precision="mixed", check_every=K, max_iter=mi)
    dev = sol.upload(b)
    for _ in range(2):
        out = sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    print(f"K={K}: {sol.last_kernel_ms():.2f} ms  {1024 / sol.last_kernel_ms() * 1e3:,.0f} QP/s solved {np.mean((st==1)|(st==2)):.4f} admm {np.mean(it % 1000):.0f} polish {np.mean(it // 1000):.2f}")
