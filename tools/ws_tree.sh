#!/bin/bash
# tools/ws_tree.sh "<bench args>" tree ... — HBM write / read request counters per kernel (one rocprofv3 --pmc pass per checkout of the repository)
args=$1; shift
REPO=$(pwd)
for t in "$@"; do
  tag=$(echo $t | tr '/.' '__'); OUT=$REPO/gpurun_out/ws_tree_$tag; rm -rf $OUT; mkdir -p $OUT
  # (WRITE_SIZE and FETCH_SIZE do not fit one pass: "Request exceeds the capabilities of the hardware to collect" — and the aborted profiler then hangs until its timeout)
  ( export TMPDIR=/tmp && cd $REPO/$t && timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc -o pmc --output-format csv -- python bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-scatter-leg --no-extra-configs > $OUT/log 2>&1
    timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc --output-format csv -- python bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-scatter-leg --no-extra-configs > $OUT/log2 2>&1 )
  python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_(fused|wf_[a-z_]+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            name = m.group(0)
            if "k_wf_trace" in name: name = "k_wf_trace<ANY>" if re.search(r"k_wf_trace<\d+, (true|false), true", r["Kernel_Name"]) else "k_wf_trace<closest>"
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for kn, a in sorted(agg.items()):
    print("$t", kn, {k: "%.4g" % v for k, v in sorted(a.items())})
PY
done
