#!/bin/bash
# cache / issue behaviour of the wavefront kernels on the config-5 geometry (one --pmc group per pass)
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_stair; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --scene staircase --width 720 --height 1280 --bins 400 --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-scatter-leg $@"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o pmc --output-format csv -- $CMD > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
cd $REPO
python - <<PY
import csv, glob, collections, re
aggs=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"k_(fused|wf_[a-z_]+)", r["Kernel_Name"])
        if m: aggs[m.group(0)][r["Counter_Name"]]+=float(r["Counter_Value"])
for kn,agg in aggs.items():
    print("==", kn)
    for k,v in sorted(agg.items()): print(f"   {k:36s}{v:.4g}")
    if agg.get("SQ_ACTIVE_INST_VALU") and agg.get("SQ_THREAD_CYCLES_VALU"):
        print("   avg active lanes per VALU inst:", agg["SQ_THREAD_CYCLES_VALU"]/agg["SQ_ACTIVE_INST_VALU"])
PY
