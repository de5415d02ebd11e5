#!/bin/bash
# Executed work of the solve kernel (round-2 review, item 4): floating-point instruction mix from the SQ counters, in separate --pmc
# passes (kernel-trace only), plus the FETCH_SIZE / WRITE_SIZE calibration at 4 / 8 / 16 B per lane on a copy of known size.
# usage (on the GPU box): bash tools/pmc_flops.sh <outdir under gpurun_out>
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1 || rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32" \
           "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_INT64" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown > $OUT/pass$i.log 2>&1 || echo "pass $i failed (a counter of this set may not exist on gfx950: see $OUT/avail.txt)"
done
j=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  j=$((j+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/calib$j -- $GRAFT_REPO_ROOT/tools/ub_r3/fetch_calib > $OUT/calib$j.log 2>&1 || echo "calibration pass $j failed"
done
python3 - <<PY
import csv, glob, collections
print("== solve kernel, per dispatch")
for p in sorted(glob.glob("$OUT/pass*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "mpcqp_" in row["Kernel_Name"] and "order_kernel" not in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} per-dispatch mean {sum(v)/len(v):16.0f}  (n={len(v)})")
print("== calibration copies (1 GiB read + 1 GiB written per launch)")
for p in sorted(glob.glob("$OUT/calib*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "copy_kernel" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:60s} {k[1]:12s} per-dispatch mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
