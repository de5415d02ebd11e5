"""Copies one run of tools/refresh_profiles_r3.sh (gpurun_out/<tag>/) into profiles/<prefix>_* and recomputes the two JSON summaries
bench.py reads (HBM traffic with the calibrated FETCH_SIZE correction, floating-point instruction mix).
usage: python tools/install_profiles.py <tag> [prefix = r03f]"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; prefix = sys.argv[2] if len(sys.argv) > 2 else "r03f"
src = os.path.join(R, "gpurun_out", tag) + "/"; dst = os.path.join(R, "profiles", prefix + "_")
for f in ("bench.json", "bench_under_rocprof.json", "kernel_stats.csv", "hbm_traffic_pmc.txt", "pmc.txt", "flops_pmc.txt", "phase_stamps.txt", "stage_rate.txt",
          "stage_kernel_stats.csv", "all_configs.txt", "tail_timeline.txt", "hardest_qps.txt", "rollout_rate.txt"):
    if os.path.exists(src + f): shutil.copy(src + f, dst + f)


def val(path, key, who=None):
    for l in open(path):
        p = l.split()
        if who is None and p and p[0] == key: return float(p[-2])
        if who is not None and len(p) > 1 and p[0] == who and p[1] == key: return float(p[-2])
    raise KeyError(key)


h = src + "hbm_traffic_pmc.txt"
fs, fo, ws, wo = val(h, "FETCH_SIZE", "solve"), val(h, "FETCH_SIZE", "order"), val(h, "WRITE_SIZE", "solve"), val(h, "WRITE_SIZE", "order")
alg = 6701056   # 1 636 B/QP x 4096
tj = json.load(open(dst + "hbm_traffic.json"))
tj.update({"FETCH_SIZE_KB": {"solve": fs, "order": fo}, "WRITE_SIZE_KB": {"solve": ws, "order": wo}, "hbm_bytes_per_launch_uncorrected": (fs + fo + ws + wo) * 1024,
           "hbm_bytes_per_launch": (2 * fs + 2 * fo + ws + wo) * 1024, "ratio": (2 * fs + 2 * fo + ws + wo) * 1024 / alg})
json.dump(tj, open(dst + "hbm_traffic.json", "w"), indent=1)
fp = src + "flops_pmc.txt"
g = lambda k: int(val(fp, k))
w = {"SQ_INSTS_VALU": g("SQ_INSTS_VALU"), "FMA_F32": g("SQ_INSTS_VALU_FMA_F32"), "ADD_F32": g("SQ_INSTS_VALU_ADD_F32"), "MUL_F32": g("SQ_INSTS_VALU_MUL_F32"),
     "FMA_F64": g("SQ_INSTS_VALU_FMA_F64"), "ADD_F64": g("SQ_INSTS_VALU_ADD_F64"), "MUL_F64": g("SQ_INSTS_VALU_MUL_F64"), "TRANS_F32": g("SQ_INSTS_VALU_TRANS_F32"),
     "TRANS_F64": g("SQ_INSTS_VALU_TRANS_F64"), "CVT": g("SQ_INSTS_VALU_CVT"), "INT32": g("SQ_INSTS_VALU_INT32"), "INT64": g("SQ_INSTS_VALU_INT64"), "MFMA": g("SQ_INSTS_MFMA")}
fps = sum(w[k] for k in ("FMA_F32", "ADD_F32", "MUL_F32", "FMA_F64", "ADD_F64", "MUL_F64", "TRANS_F32", "TRANS_F64"))
f32 = 2 * w["FMA_F32"] + w["ADD_F32"] + w["MUL_F32"]; f64 = 2 * w["FMA_F64"] + w["ADD_F64"] + w["MUL_F64"]
mn, mx = 64 * (f32 + f64), 64 * (2 * f32 + f64)
fj = json.load(open(dst + "flops.json"))
fj.update({"wave_instructions": w, "fp_share_of_valu_issue": fps / w["SQ_INSTS_VALU"], "fma_share_of_valu_issue": (w["FMA_F32"] + w["FMA_F64"]) / w["SQ_INSTS_VALU"],
           "executed_flops_per_launch_min": mn, "executed_flops_per_launch_max": mx, "executed_flops_per_qp_min": mn / 4096, "executed_flops_per_qp_max": mx / 4096})
json.dump(fj, open(dst + "flops.json", "w"), indent=1)
print("traffic", tj["hbm_bytes_per_launch"], round(tj["ratio"], 3), "VALU", w["SQ_INSTS_VALU"], round(fj["fp_share_of_valu_issue"], 3), round(fj["fma_share_of_valu_issue"], 3), mn / 4096, mx / 4096)
for l in open(src + "kernel_stats.csv"):
    if "wrench_solve" in l or "order_kernel" in l: print(l.split('",')[1:4])
d = json.loads(open(src + "bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["config"]["admm_iters_mean"])
for k, v in d["breakdown"].items(): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "note"})
