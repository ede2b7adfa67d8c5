#!/bin/bash
# tools/stats.sh <tag> [bench args]: rocprofv3 --kernel-trace --stats for one bench run; prints kernel stats
TAG=$1; shift; REPO=$(pwd); OUT=$REPO/gpurun_out/stats_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --truncate-kernels -d $OUT -o t --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
cat $OUT/t_kernel_stats.csv
