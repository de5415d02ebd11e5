#!/bin/bash
# HBM traffic of the solve kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "mpcqp_" in row["Kernel_Name"]:
            kn = "order" if "order_kernel" in row["Kernel_Name"] else "solve"
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:6s} {k[1]:16s} per-dispatch mean {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
