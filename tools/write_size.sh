#!/bin/bash
# tools/write_size.sh lib.so ... — HBM write / fetch counters of k_fused per launch for kernel variants (one rocprofv3 --pmc pass each)
REPO=$(pwd)
for lib in "$@"; do
  OUT=$REPO/gpurun_out/ws_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && MITRANSIENT_AMD_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs > $OUT/log 2>&1 )
  python - <<PY
import csv, glob
tot={}; n={}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            k=r["Counter_Name"]; tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
print("$lib", {k: "%.3g" % (tot[k]/n[k]) for k in tot})
PY
done
