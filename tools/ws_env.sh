#!/bin/bash
# tools/ws_env.sh "<ENV=1 ...>" <tag> — WRITE_SIZE / L2 counters of k_fused for one bench variant selected by environment variables
ENVS=$1; TAG=$2; REPO=$(pwd); OUT=$REPO/gpurun_out/ws_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && env $ENVS timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs > $OUT/log 2>&1 )
python - <<PY
import csv, glob
tot={}; n={}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            k=r["Counter_Name"]; tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
print("$TAG", {k: "%.4g" % (tot[k]/n[k]) for k in tot}, "(WRITE_SIZE in KiB per launch)")
PY
