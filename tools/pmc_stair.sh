#!/bin/bash
# cache behaviour of the wavefront kernels on the config-5 geometry
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_stair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --scene staircase --steps 1 --warmup 0 --no-cpu-baseline --no-scatter-leg $@"
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
cd $REPO
python - <<PY
import csv, glob, collections, re
aggs=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"k_(fused|wf_[a-z]+)", r["Kernel_Name"])
        if m: aggs[m.group(0)][r["Counter_Name"]]+=float(r["Counter_Value"])
for kn,agg in aggs.items():
    print("==", kn)
    for k,v in sorted(agg.items()): print(f"   {k:36s}{v:.4g}")
PY
