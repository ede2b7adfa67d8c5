#!/bin/bash
# tools/stats_stair.sh [bench args]: per-kernel rocprofv3 stats of the config-5 (staircase) bench
REPO=$(pwd); OUT=$REPO/gpurun_out/stats_stair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --truncate-kernels -d $OUT -o t --output-format csv -- python $REPO/bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg "$@" > $OUT/log.txt 2>&1
tail -c 400 $OUT/log.txt; echo; head -10 $OUT/t_kernel_stats.csv
