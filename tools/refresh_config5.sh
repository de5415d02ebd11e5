#!/bin/bash
# One GPU call for the horizon-20 evidence of the current kernel: rocprofv3 kernel stats of config 5 (MIXED and F64),
# throughput of every BASELINE configuration, the config-5 K x eps x precision sweep.
# usage (on the GPU box): bash tools/refresh_config5.sh <tag>      -> gpurun_out/<tag>/...
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in mixed f64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$P -- python3 $R/tools/all_configs.py config5 --precision $P > $OUT/config5_$P.txt 2> $OUT/stats_$P.log
  cp $(ls $OUT/stats_$P/*/*kernel_stats.csv | head -1) $OUT/config5_${P}_kernel_stats.csv
  echo "stats $P done"
done
python3 $R/tools/all_configs.py > $OUT/all_configs.txt 2>&1
echo "all configs done"
python3 $R/tools/config5_sweep.py > $OUT/config5_sweep.log 2>&1
echo "sweep done"
