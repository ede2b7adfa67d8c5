#!/bin/bash
# tools/pmc.sh <lib.so> <tag>: quick SQ counter pass for the path kernel
LIB=$(pwd)/$1; TAG=$2; shift; shift; EXTRA="$@"; REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA"
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  MITRANSIENT_AMD_LIB=$LIB rocprofv3 --kernel-trace --pmc $grp -d $OUT/$name -o pmc --output-format csv -- $CMD > $OUT/$name.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
import re
aggs=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r"k_(fused|wf_[a-z]+)", r["Kernel_Name"])
        if m: aggs[m.group(0)][r["Counter_Name"]]+=float(r["Counter_Value"])
for kn,agg in aggs.items():
    print("==", kn)
    for k,v in sorted(agg.items()): print(f"   {k:26s}{v:.4g}")
    if agg.get("SQ_ACTIVE_INST_VALU"):
        print("   avg active lanes per VALU inst:", agg["SQ_THREAD_CYCLES_VALU"]/agg["SQ_ACTIVE_INST_VALU"])
PY
