#!/bin/bash
# HBM traffic of the horizon-20 kernels (they spill inside the solve: 72 VGPRs MIXED, 120 F64; round-2 review, weak #5): FETCH_SIZE /
# WRITE_SIZE passes on BASELINE config 5 (B = 4096), both precisions.  usage (GPU box): bash tools/pmc_n20.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
for prec in mixed f64; do
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${prec}_$set -- python3 $GRAFT_REPO_ROOT/tools/all_configs.py config5 --precision $prec > $OUT/${prec}_$set.log 2>&1 || echo "$prec $set failed"
  done
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "mpcqp_wrench_solve" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{p.split('/')[-3]:18s} {k[0]:60s} {k[1]:12s} per-dispatch mean {sum(v)/len(v):14.1f} KB (n={len(v)})")
print("algorithmic: 3156 B/QP x 4096 = 12.9 MB per launch (fp32 buffers); FETCH_SIZE reads half of the bytes (profiles/r03_hbm_traffic.json)")
PY
