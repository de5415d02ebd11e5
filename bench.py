#!/usr/bin/env python3
"""Headline benchmark: QP solves/sec (horizon=10, 4-contact Lite3) at batch=4096 per GPU.

One "step" = one pass of the hot path (raw operator tuple -> solved ground-reaction forces: model build,
discretisation, condensing, matrix inversion, ADMM, polish) over one resident batch of 4096 synthetic QPs
(mixed gaits + friction sweep, seed 20250809 + rank).  Inputs and outputs stay
in HBM for the whole timed region.  N>1: the batch axis is sharded, one rank per GPU, no data-path collective.  Two modes:
  weak   (default)            4096 QPs per GPU, every rank its own draw (seed + rank); `--same-shards`: every rank the configured batch
  strong (--global-batch G)   BASELINE configs[3]: G QPs of the config-4 distribution (seed 20250810) sharded contiguously over the
                              ranks (mpcqp.dist.shard_bounds), value = G x steps / time
`--allgather` adds the optional RCCL all-gather of stage-0 GRFs.  The per-rank times (min / mean / max) are reported next to the
job time (= the slowest rank), so the spread between draws is visible.

Prints ONE JSON line (rank 0).  `roofline` prices the solve kernel against the packed-fp32 vector peak (the roof that
binds: the path is neither HBM- nor MFMA-shaped) with the ALGORITHMIC flop count of SURVEY.md section
8(d); `cpu_baseline` times the fp64 CPU oracle on a bounded sample of the same workload on the host cores, on one core and
on all of them (a reported baseline, not the target; the reference's own CasADi + OSQP path cannot run offline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import mpcqp  # noqa: E402

PEAK_FP32_TFLOPS = 157.3     # MI355X fp32 vector = fp32 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0        # HBM3E spec (6290 measured float4 copy)
TRAFFIC_FILE = "r03f_hbm_traffic.json"     # PMC passes of the current kernel (profiles/), FETCH_SIZE corrected by the round-3 calibration
SQ_FILE = "r03f_pmc.txt"                     # SQ counter passes of the same kernel and workload (tools/pmc_run.sh)
FLOPS_FILE = "r03f_flops.json"               # floating-point instruction mix of the same kernel and workload (tools/pmc_flops.sh)
PEAK_VALU_ISSUE = 1024 * 2.4e9 / 4          # wave-instructions/s: 1024 SIMDs, one wave64 vector instruction per 4 cycles (an fp64 FMA: 8)


def algorithmic_flops(N, K):
    """n^2 s + n^3/3 + K (2 n^2 + 60 n), n = 12N, s = 13N."""
    n, s = 12 * N, 13 * N
    return n * n * s + n ** 3 / 3.0 + K * (2 * n * n + 60 * n)


def algorithmic_bytes(N, elt=4):
    """operator tuple in + forces/status out (fp32: 1636 B at N=10)."""
    return (13 + 12 * N + 13 * (N + 1) + 1) * elt + 4 * N + 12 * N * elt + 8


def _host_cores():
    cores = len(os.sched_getaffinity(0))               # the threads OpenMP will actually get ...
    try:                                               # ... capped by the container's CPU quota (cgroup v2), if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return min(cores, 64)


def cpu_baseline(batch, cfg_kw, sample):
    """fp64 CPU oracle (oracle/, OpenMP over the batch) on the first `sample` QPs; same solver settings.  All host cores
    (the headline `value`) and one core; about 8 s of wall each."""
    path = os.path.join(REPO, "oracle", "libmpcqp_oracle.so")
    if not os.path.exists(path):
        return None
    import ctypes
    cores = _host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)         # read by libgomp when the library is loaded below
    lib = mpcqp.Library(path)
    eng = mpcqp.Engine(lib, lib.default_config(**cfg_kw))
    try:
        set_threads = ctypes.CDLL("libgomp.so.1").omp_set_num_threads
    except OSError:
        set_threads = None

    def timed(n_threads, n_qp, budget):
        if set_threads is not None:
            set_threads(int(n_threads))
        sub = {k: batch[k][:n_qp] for k in ("x0", "r", "contact", "xdes", "mu")}
        eng.solve_batch_host(sub["x0"][:n_threads], sub["r"][:n_threads], sub["contact"][:n_threads], sub["xdes"][:n_threads],
                             sub["mu"][:n_threads], want_X=False)
        reps, dt, st = 0, 0.0, None
        t0 = time.perf_counter()
        while dt < budget and reps < 64:
            out = eng.solve_batch_host(sub["x0"], sub["r"], sub["contact"], sub["xdes"], sub["mu"], want_X=False)
            reps += 1
            dt = time.perf_counter() - t0
            st = out["status"]
        return reps * n_qp / dt, reps, dt, float(((st == 1) | (st == 2)).mean())

    v_all, reps, dt, solved = timed(cores, sample, 8.0)
    one = None
    if set_threads is not None:
        v_one, reps1, dt1, _ = timed(1, max(32, sample // 16), 8.0)
        one = {"value": v_one, "cores": 1, "sample": f"first {max(32, sample // 16)} QPs x {reps1} passes, {dt1:.1f} s wall"}
    return {"value": v_all, "unit": "QP solves/s", "cores": cores, "kind": "port", "one_core": one,
            "reference_osqp_path": "unavailable offline: casadi / osqp / dartpy are absent from the image and cannot be installed "
                                   "(SURVEY.md section 8c); the reference's own published figure is 61 Hz per solve at N = 60 (BASELINE.md)",
            "sample": f"first {sample} QPs of the same batch x {reps} passes, fp64 condensed OSQP-style ADMM + polish (same rho/sigma/"
                      f"relax/iteration cap as the GPU run), OpenMP over the batch, {dt:.1f} s wall; solved fraction {solved:.3f}"}


def gait_breakdown(solver, N, delta, B, steps=5, precision="mixed"):
    """Secondary figures on the same engine: single-gait batches, a true 4-contact batch (all feet down on all stages), the
    headline workload drawn with other seeds, and a batch that fills the device many times over."""
    out = {}
    allg = ("trot", "pronk", "amble", "gallop")
    cases = [(g, (g,), None, 20250809) for g in allg] + [("all_stance", ("trot",), 1, 20250809)]
    # the workload of `value` drawn with other seeds: a launch of 4096 is as long as its few longest QPs, so the rate moves with
    # the draw (the solver's thresholds were chosen on such batches, not on the one `value` is quoted on)
    cases += [(f"mixed_seed_{sd}", allg, None, sd) for sd in (1, 2, 3, 4)]
    note = lambda m: print(f"[bench breakdown] {m}", file=sys.stderr, flush=True)   # (progress on stderr: a fault names its piece)
    for name, gaits, force_contact, seed in cases:
        note(name)
        b = mpcqp.synth.make_batch(B, N, delta, seed, gaits, (0.3, 0.5, 0.7, 1.0))
        if force_contact is not None:
            b["contact"][:] = 1
        dev = solver.upload(b)
        for _ in range(2):
            o = solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            o = solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        e1.record()
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy()
        out[name] = {"qp_per_s": B * steps / (e0.elapsed_time(e1) * 1e-3), "solved_fraction": float(((st == 1) | (st == 2)).mean())}
    # the same engine on a batch that fills the device many times over (BASELINE config 4's 65 536 QPs on ONE GPU): the rate when
    # the launch is not as long as its longest QPs
    big = 65536
    note("mixed_batch_65536")
    b = mpcqp.synth.make_batch(big, N, delta, 20250810, ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0))
    dev = solver.upload(b)
    for _ in range(2):
        o = solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        o = solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
    e1.record()
    torch.cuda.synchronize()
    st = o["status"].cpu().numpy()
    out["mixed_batch_65536"] = {"qp_per_s": big * 3 / (e0.elapsed_time(e1) * 1e-3), "solved_fraction": float(((st == 1) | (st == 2)).mean())}
    # the reference's arithmetic is all-fp64: the same workload with every tile, vector and residual in fp64
    note("f64_b4096")
    try:
        s64 = mpcqp.MPCBatch(N=N, delta=delta, device=solver.device.index, io_dtype="f32", precision="f64", flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING)
        b = mpcqp.synth.make_batch(B, N, delta, 20250809, allg, (0.3, 0.5, 0.7, 1.0))
        dev = s64.upload(b)
        for _ in range(2):
            o = s64.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            o = s64.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        e1.record()
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy()
        out["f64_b4096"] = {"qp_per_s": B * steps / (e0.elapsed_time(e1) * 1e-3), "solved_fraction": float(((st == 1) | (st == 2)).mean()),
                            "note": "all-fp64 arithmetic (MPCQP_PREC_F64), same batch as `value`"}
        del s64
    except Exception as e:
        out["f64_b4096"] = {"error": repr(e)}
    # the reference's OWN configuration (N = 60, delta = 0.01, src/main.py:37,41) on the stage-wise engine: secondary, a different
    # horizon than the metric's -- the reference's figure for this setting is 61 solves/s on its CPU
    note("reference_horizon_n60_b1024")
    try:
        Bn = 1024
        s60 = mpcqp.MPCBatch(N=60, delta=0.01, device=solver.device.index, io_dtype="f32", precision=precision, flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING)
        b = mpcqp.synth.make_batch(Bn, 60, 0.01, 20250809, allg, (0.3, 0.5, 0.7, 1.0))
        dev = s60.upload(b)
        o = s60.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            o = s60.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"])
        e1.record()
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy()
        out["reference_horizon_n60_b1024"] = {"qp_per_s": Bn * 2 / (e0.elapsed_time(e1) * 1e-3), "solved_fraction": float(((st == 1) | (st == 2)).mean()),
                                              "note": "N = 60, delta = 0.01, mixed gaits + mu sweep, stage-wise engine (csrc/mpcqp_stage.h)"}
        del s60
        # ... and ONE robot, the reference's own use (its log: 61.2 solves/s on its CPU, src/main.py:194-202): solve latency of a batch of
        # one, cold and warm-started from the previous solution (src/mpc.py:270-271), host-synchronised after every solve
        one = {k: (v[:1] if isinstance(v, np.ndarray) and len(v) == Bn else v) for k, v in b.items()}
        for tag, kw in (("cold", {}), ("warm", {"warm_start": True})):
            s1 = mpcqp.MPCBatch(N=60, delta=0.01, device=solver.device.index, io_dtype="f32", precision=precision, **kw)
            d1 = s1.upload(one)
            for _ in range(3):
                o = s1.solve_batch(d1["x0"], d1["r"], d1["contact"], d1["xdes"], d1["mu"]); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                o = s1.solve_batch(d1["x0"], d1["r"], d1["contact"], d1["xdes"], d1["mu"]); torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t0) / 10
            out["reference_horizon_n60_single_robot_" + tag] = {"solves_per_s": 1.0 / dt1, "ms_per_solve": dt1 * 1e3, "status": int(o["status"][0].item()),
                                                                "note": "batch of one, wall clock incl. launch and synchronisation"}
            del s1
    except Exception as e:
        out["reference_horizon_n60_b1024"] = {"error": repr(e)}
    # two independent batches of B in flight: two handles on two HIP streams (a handle serves one stream at a time,
    # include/mpcqp.h).  The tail of one launch is filled by the head of the other -- what a caller with more than one fleet gets per
    # batch of B.  NOT `value`: that is one batch per step on one stream.
    note("two_batches_two_streams")
    try:
        sols = [solver, mpcqp.MPCBatch(N=N, delta=delta, device=solver.device.index, io_dtype="f32", precision=precision,
                                       flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING)]
        devs = [s_.upload(mpcqp.synth.make_batch(B, N, delta, 20250809 + i, allg, (0.3, 0.5, 0.7, 1.0))) for i, s_ in enumerate(sols)]
        streams = [torch.cuda.Stream(device=solver.device) for _ in range(2)]
        torch.cuda.synchronize()
        reps = 4 * steps

        def both():
            for s_, d, st_ in zip(sols, devs, streams):
                o_ = s_.solve_batch(d["x0"], d["r"], d["contact"], d["xdes"], d["mu"], want_X=False, stream=st_)
            return o_
        for _ in range(3):
            both()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            o = both()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        st = o["status"].cpu().numpy()
        out["two_batches_two_streams"] = {"qp_per_s": 2 * B * reps / dt2, "solved_fraction": float(((st == 1) | (st == 2)).mean()),
                                          "note": "secondary: two handles / two HIP streams, wall clock over both; not the headline"}
    except Exception as e:   # a secondary figure must not take the bench line down
        out["two_batches_two_streams"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="QPs per GPU")
    ap.add_argument("--precision", default="mixed", choices=["mixed", "f64"])
    ap.add_argument("--allgather", action="store_true", help="all-gather stage-0 GRFs over RCCL every step")
    ap.add_argument("--distinct-shards", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--same-shards", action="store_true", help="N > 1, weak mode: every rank solves the configured batch (seed 20250809) instead of its own draw")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many QPs in total (config-4 distribution), sharded contiguously over the ranks")
    ap.add_argument("--first-block", type=int, default=0, help="MpcQpConfig.first_block (0: the engine's default; tools/first_block_sweep.sh)")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the secondary per-gait / all-stance figures")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    device_index = local_rank % ndev        # one rank per GPU on a real node; wraps only in single-GPU rehearsals
    torch.cuda.set_device(device_index)
    dist = None
    backend = os.environ.get("MPCQP_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" for 1-GPU rehearsals
    # (MPCQP_BENCH_DIST1=1 under `torch.distributed.run --nproc-per-node 1`: the process-group branch with one rank -- how the
    #  RCCL calls below are rehearsed on a one-GPU box, tests/test_gpu_configs45.py)
    if world > 1 or os.environ.get("MPCQP_BENCH_DIST1") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    sync_device = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")

    N, delta, B = 10, 0.03, args.batch
    gaits, mus = ("trot", "pronk", "amble", "gallop"), (0.3, 0.5, 0.7, 1.0)
    # this rank's shard.  Weak scaling: 4096 QPs per GPU, each rank its own draw of the workload (seed + rank; rank 0 = the configured
    # batch) -- a launch of 4096 is as long as its longest QPs, so the ranks' times differ by the draw (reported below), and the job
    # is as slow as the unluckiest one.  Strong scaling (--global-batch): BASELINE configs[3], the batch sharded contiguously.
    strong = args.global_batch > 0
    if strong:
        from mpcqp.dist import shard_bounds
        lo, hi = shard_bounds(args.global_batch, world, rank)
        full = mpcqp.synth.make_batch(args.global_batch, N, delta, 20250810, gaits, mus)
        batch = {k: (v[lo:hi] if isinstance(v, np.ndarray) and len(v) == args.global_batch else v) for k, v in full.items()}
        B = hi - lo
        del full
    else:
        batch = mpcqp.synth.make_batch(B, N, delta, 20250809 + (0 if args.same_shards else rank), gaits, mus)
    # (MPCQP_FLAG_NO_TIMING: the engine's own per-call event pair is a diagnostic; the timed region below is bracketed by this
    #  script's events on the same stream)
    solver = mpcqp.MPCBatch(N=N, delta=delta, device=device_index, io_dtype="f32", precision=args.precision,
                            flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_NO_TIMING, first_block=args.first_block)
    dev = solver.upload(batch)
    gathered = None
    if args.allgather and dist is not None:   # RCCL: device buffers over xGMI; gloo (1-GPU rehearsals): through host memory
        gathered = torch.empty((world * B, 12), dtype=torch.float32, device=solver.device if backend == "nccl" else "cpu")

    ragged = strong and args.global_batch % world != 0

    def step():
        out = solver.solve_batch(dev["x0"], dev["r"], dev["contact"], dev["xdes"], dev["mu"], want_X=False)
        if gathered is not None:
            if ragged:       # unequal shards: padded gather (mpcqp.dist)
                mpcqp.dist.all_gather_stage0(out["u"] if backend == "nccl" else out["u"].cpu(), args.global_batch)
            else:
                u0 = out["u"][:, 0, :].contiguous()
                dist.all_gather_into_tensor(gathered, u0 if backend == "nccl" else u0.cpu())
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()                    # same stream the engine launches on (torch's current stream)
    for i in range(args.steps):
        out = step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = evs[0].elapsed_time(evs[-1]) / args.steps       # average launch duration over the timed region
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    rank_ms = [kernel_ms]
    total_qps = B
    if dist:
        mine = torch.tensor([dt, kernel_ms, float(B), float(((out["status"] == 1) | (out["status"] == 2)).sum().item())], dtype=torch.float64, device=sync_device)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        dt, kernel_ms = float(allr[:, 0].max()), float(allr[:, 1].max())     # the job is as slow as its slowest rank
        rank_ms = [float(v) for v in allr[:, 1]]
        total_qps = int(allr[:, 2].sum())
        solved_all = float(allr[:, 3].sum()) / total_qps

    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    solved = float(((status == 1) | (status == 2)).mean())
    if dist:
        solved = solved_all
    k_mean = float((iters % 1000).mean())
    if rank == 0:
        value = total_qps * args.steps / dt
        flops = algorithmic_flops(N, k_mean)
        achieved = flops * B / (kernel_ms * 1e-3) / 1e12
        hbm = algorithmic_bytes(N) * B / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch: NOT measured in this run (PMC counters need their own rocprofv3 passes, tools/pmc_hbm.sh); the
        # figure is read from the committed summary of those passes on this same workload and kernel, and says so
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", TRAFFIC_FILE)))
            if B == 4096 and args.precision == "mixed":
                traffic, traffic_src = tj["hbm_bytes_per_launch"], "profiles/" + TRAFFIC_FILE
        except (OSError, ValueError, KeyError):
            pass
        # what actually binds the kernel ("Roofline that binds"): vector-instruction issue.  SQ_INSTS_VALU per
        # launch comes from the committed counter passes (not measured in this run, like `traffic`), the duration is this run's.
        issue = None
        try:
            if B == 4096 and args.precision == "mixed":
                for ln in open(os.path.join(REPO, "profiles", SQ_FILE)):
                    if ln.startswith("SQ_INSTS_VALU "):
                        n_valu = float(ln.split()[-2])
                        rate = n_valu / (kernel_ms * 1e-3)
                        issue = {"achieved": rate / 1e9, "peak": PEAK_VALU_ISSUE / 1e9, "unit": "G wave-instructions/s", "frac": rate / PEAK_VALU_ISSUE,
                                 "valu_instructions_per_launch": n_valu, "source": "profiles/" + SQ_FILE,
                                 "note": "a launch of 4096 QPs ends with its longest QPs alone on their SIMDs (DESIGN.md section 5); fp64 FMAs of the polish issue at half this rate"}
                        break
        except (OSError, ValueError, IndexError):
            pass
        # executed work next to the algorithmic count (round-2 review): the wrench-space form executes fewer flops than the condensed
        # formulation SURVEY 8(d) prices, so `frac` above is notional; the instruction mix comes from committed counter passes
        executed = None
        try:
            if B == 4096 and args.precision == "mixed":
                fj = json.load(open(os.path.join(REPO, "profiles", FLOPS_FILE)))
                ex_min, ex_max = fj["executed_flops_per_qp_min"], fj["executed_flops_per_qp_max"]
                executed = {"executed_flops_per_qp": {"min": ex_min, "max": ex_max},
                            "executed_over_algorithmic": {"min": ex_min / flops, "max": ex_max / flops},
                            "executed_tflops": {"min": ex_min * B / (kernel_ms * 1e-3) / 1e12, "max": ex_max * B / (kernel_ms * 1e-3) / 1e12},
                            "fp_share_of_valu_issue": fj["fp_share_of_valu_issue"], "fma_share_of_valu_issue": fj["fma_share_of_valu_issue"],
                            "source": "profiles/" + FLOPS_FILE, "note": fj["note"]}
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "QP solves/sec (horizon=10, 4-contact Lite3) at batch=4096", "value": value, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": median_ms,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "rank_kernel_ms": {"min": min(rank_ms), "mean": sum(rank_ms) / len(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
            "dtype": {"mixed": "f32 tiles + f64 residuals", "f64": "f64"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[2]: batch=4096/GPU mixed gaits (trot/pronk/amble/gallop) + mu sweep, horizon=10, "
                                   "dt=0.03, Lite3 constants, alpha=1e-2, euler", "batch_per_gpu": B, "global_batch": total_qps,
                       "shards": ("contiguous shards of one batch of %d (seed 20250810)" % args.global_batch) if strong else
                                 ("every rank solves the configured batch" if args.same_shards else "one draw per rank (seed 20250809 + rank)"), "horizon": N,
                       "precision": args.precision, "admm_block": int(solver.cfg.check_every), "max_iter": int(solver.cfg.max_iter),
                       "polish": bool(solver.cfg.flags & 1), "allgather": bool(gathered is not None),
                       "solved_fraction": solved, "admm_iters_mean": k_mean, "polish_steps_mean": float((iters // 1000).mean())},
            "roofline": {"bound": "valu_fp32", "bound_note": "packed-fp32 vector peak (v_pk_fma_f32; numerically the fp32 MFMA peak of the "
                         "guide): the kernel has no GEMM-shaped work and 1.6 KB of compulsory HBM traffic per QP",
                         "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "mpcqp_wrench_solve<double,float,double> (one launch per solve_batch, after a 5 us ordering pre-pass; both inside kernel_ms)" if args.precision == "mixed"
                         else "mpcqp_wrench_solve<double,double,double>",
                         "kernel_ms": kernel_ms,
                         "algorithmic_flops_per_qp": flops,
                         "achieved_is": "notional: ALGORITHMIC flops of the condensed formulation (SURVEY 8d) over the measured duration; the engine executes fewer (see `executed`)",
                         "executed": executed,
                         "hbm": {"achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS,
                                 "algorithmic_bytes_per_qp": algorithmic_bytes(N), "note": "non-binding roof (SURVEY 8d)"},
                         "valu_issue": issue},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(batch, dict(N=N, delta=delta, rho=solver.cfg.rho, sigma=solver.cfg.sigma, relax=solver.cfg.relax,
                                          max_iter=int(solver.cfg.max_iter), check_every=int(solver.cfg.check_every),
                                          eps_abs=solver.cfg.eps_abs, eps_rel=solver.cfg.eps_rel,
                                          polish_max=int(solver.cfg.polish_max), flags=int(solver.cfg.flags)),
                              min(args.cpu_sample, B))
            if cb:
                line["cpu_baseline"] = cb
        if world == 1 and not args.no_breakdown:
            line["breakdown"] = gait_breakdown(solver, N, delta, B, precision=args.precision)
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
