#!/bin/bash
# One GPU call: roll-out rate + rocprofv3 kernel stats of the roll-out (expand / order / solve / advance launches per tick).
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/rollout_rate.py 4096 100 > $OUT/rollout_rate.txt 2>&1
echo "rate done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/rollout_rate.py 4096 100 > $OUT/rollout_under_rocprof.txt 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/rollout_kernel_stats.csv
echo "stats done"
