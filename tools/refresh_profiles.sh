#!/bin/bash
# One GPU call that regenerates every committed measurement of the current kernel: rocprofv3 kernel stats of the bench command,
# the bench line itself (with cpu_baseline and the per-gait breakdown), HBM / SQ counter passes, phase stamps.
# usage (on the GPU box): bash tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-breakdown > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "stats done"
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.log
echo "bench done"
bash $R/tools/pmc_hbm.sh $TAG/hbm > $OUT/hbm_traffic_pmc.txt 2>&1
echo "hbm done"
bash $R/tools/pmc_run.sh $TAG/sq > $OUT/pmc.txt 2>&1
echo "sq done"
python3 $R/tools/stamps.py 4096 mixed 10 > $OUT/phase_stamps.txt 2>&1
python3 $R/tools/stamps.py 65536 mixed 10 >> $OUT/phase_stamps.txt 2>&1
echo "stamps done"
