"""Developer tool: the PCIe-inclusive rate of the headline workload -- never bench.py's `value`.
The boundary takes device pointers; a caller whose operands live in host memory pays an upload of the operator tuple (1 148 B/QP
fp32, or ~100 B/QP through the gait entry point) and a download of the stage-0 forces + status (52 B/QP) around every solve.
Measured here with pinned host buffers and non-blocking copies on the solve's stream: resident / tuple upload / gait upload."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import mpcqp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, STEPS = 10, 50
batch = mpcqp.synth.config3(B)
sol = mpcqp.MPCBatch(N=N, delta=0.03, io_dtype="f32", precision="mixed", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING)
dev = sol.upload(batch)
keys = ("x0", "r", "contact", "xdes", "mu")
host = {k: dev[k].cpu().pin_memory() for k in keys}
u0_host = torch.empty((B, 12), dtype=torch.float32).pin_memory()
st_host = torch.empty(B, dtype=torch.int32).pin_memory()


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS


def resident():
    return sol.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=False)


def tuple_upload():
    for k in keys:
        dev[k].copy_(host[k], non_blocking=True)
    out = resident()
    u0_host.copy_(out["u"][:, 0, :], non_blocking=True)
    st_host.copy_(out["status"], non_blocking=True)


t_res, t_up = timed(resident), timed(tuple_upload)
nbytes = sum(host[k].numel() * host[k].element_size() for k in keys)
print(f"B={B} N={N}: resident {t_res * 1e3:.3f} ms = {B / t_res / 1e6:.2f} M QP/s;  host operands (pinned, {nbytes / B:.0f} B/QP up, 52 B/QP down) "
      f"{t_up * 1e3:.3f} ms = {B / t_up / 1e6:.2f} M QP/s  (transfer share {(t_up - t_res) / t_up * 100:.0f} %, {nbytes / (t_up - t_res) / 1e9:.1f} GB/s effective)")

if hasattr(mpcqp.synth, "make_gait_batch"):
    g = mpcqp.synth.make_gait_batch(B, N, 0.03, 20250809)
    gd = sol.upload_gait(g)
    gkeys = ("x0", "ref", "feet0", "footholds", "gait", "feet_id", "mu")
    ghost = {k: gd[k].cpu().pin_memory() for k in gkeys}

    def gait_resident():
        return sol.solve_batch_gait(gd["x0"], gd["ref"], gd["feet0"], gd["footholds"], gd["gait"], gd["feet_id"], gd["mu"])

    def gait_upload():
        for k in gkeys:
            gd[k].copy_(ghost[k], non_blocking=True)
        out = gait_resident()
        u0_host.copy_(out["u"][:, 0, :], non_blocking=True)
        st_host.copy_(out["status"], non_blocking=True)

    tg_res, tg_up = timed(gait_resident), timed(gait_upload)
    gbytes = sum(ghost[k].numel() * ghost[k].element_size() for k in gkeys)
    print(f"gait entry point: resident {tg_res * 1e3:.3f} ms = {B / tg_res / 1e6:.2f} M QP/s;  host descriptors ({gbytes / B:.0f} B/QP up) "
          f"{tg_up * 1e3:.3f} ms = {B / tg_up / 1e6:.2f} M QP/s")
