#!/bin/bash
# Instruction-cache counters of the solve kernel (separate --pmc pass, kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "mpcqp_wrench" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} per-dispatch mean {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
