#!/bin/bash
# tools/profile.sh <tag> [bench args...] — kernel-trace stats + PMC passes for bench.py on the GPU box.
# Outputs summaries under gpurun_out/prof_<tag>/ (copy what should be judged into profiles/;
# `python tools/pmc_summary.py --merge gpurun_out/prof_<tag>/traffic.json [--section staircase]` updates profiles/traffic.json).
# RENDERS=<n>: renders per command (per-render sums for the wavefront kernels); STEPS / WARMUP override the bench loop.
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps ${STEPS:-2} --warmup ${WARMUP:-1} --no-cpu-baseline --no-extra-configs $@"
rocprofv3 --kernel-trace --stats --truncate-kernels -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$name -o pmc --output-format csv -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/pmc_summary.py $OUT $TAG --renders ${RENDERS:-0}
