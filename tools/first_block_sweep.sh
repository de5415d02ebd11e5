#!/bin/bash
# Sensitivity of the rate to the length of a cold solve's first ADMM block (MpcQpConfig.first_block through bench.py --first-block; default 0.7 check_every = 70):
# the bench line's headline + breakdown (other seeds, single gaits, B = 65 536) per setting.  -> gpurun_out/<tag>/first_block_<v>.json
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for V in 50 60 70 80 100; do
  python3 $R/bench.py --first-block $V --no-cpu-baseline --steps 30 > $OUT/first_block_$V.json 2> $OUT/first_block_$V.log || exit 1
  echo "first_block $V done"
done
