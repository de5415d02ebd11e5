#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per dispatch of tools/pmc_variants.py (separate --pmc passes, kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/pmc_variants.py > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob
for p in sorted(glob.glob("$OUT/pass*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(p)) if "mpcqp_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        print(r["Dispatch_Id"], r["Kernel_Name"][:70], r["Counter_Name"], r["Counter_Value"], "grid", r.get("Grid_Size"), "scratch", r.get("Scratch_Size", r.get("Private_Segment_Size")))
PY
