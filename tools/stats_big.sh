#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/stats_big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --truncate-kernels -d $OUT -o t --output-format csv -- python $REPO/tools/big_scene.py "$@" > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt; head -8 $OUT/t_kernel_stats.csv
