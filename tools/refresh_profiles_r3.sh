#!/bin/bash
# One GPU call that regenerates the committed measurements of round 3: rocprofv3 kernel stats of the bench command, the bench line,
# HBM / SQ / floating-point-mix counter passes + the FETCH_SIZE calibration, phase stamps, the stage-wise engine's kernel stats.
# usage (on the GPU box): bash tools/refresh_profiles_r3.sh <tag>      -> gpurun_out/<tag>/...
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.log
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-breakdown > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "stats done"
bash $R/tools/pmc_hbm.sh $TAG/hbm > $OUT/hbm_traffic_pmc.txt 2>&1
echo "hbm done"
bash $R/tools/pmc_run.sh $TAG/sq > $OUT/pmc.txt 2>&1
echo "sq done"
bash $R/tools/pmc_flops.sh $TAG/flops > $OUT/flops_pmc.txt 2>&1
echo "flops done"
python3 $R/tools/stamps.py 4096 mixed 10 > $OUT/phase_stamps.txt 2>&1
python3 $R/tools/stamps.py 65536 mixed 10 >> $OUT/phase_stamps.txt 2>&1
echo "stamps done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stage_stats -- python3 $R/tools/stage_check.py rate > $OUT/stage_rate.txt 2> $OUT/stage_stats.log
cp $(ls $OUT/stage_stats/*/*kernel_stats.csv | head -1) $OUT/stage_kernel_stats.csv
echo "stage done"
python3 $R/tools/all_configs.py > $OUT/all_configs.txt 2>&1
echo "all configs done"
