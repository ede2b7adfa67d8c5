#!/bin/bash
# tools/ws_libs.sh "<bench args>" lib.so ... — WRITE_SIZE / TCC hits per kernel for library variants (one rocprofv3 --pmc pass each), then FETCH_SIZE
args=$1; shift
REPO=$(pwd)
for lib in "$@"; do
  tag=$(basename $lib .so); OUT=$REPO/gpurun_out/ws_$tag; rm -rf $OUT; mkdir -p $OUT
  for grp in "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    name=$(echo $grp | cut -d' ' -f1)
    ( export TMPDIR=/tmp; MITRANSIENT_AMD_LIB=$REPO/$lib timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $OUT/$name -o pmc --output-format csv -- python bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-scatter-leg --no-extra-configs > $OUT/$name.log 2>&1 )
  done
  python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_(fused|wf_[a-z_]+)", r["Kernel_Name"])
        if m:
            name = m.group(0)
            if name == "k_wf_trace": name += "<ANY>" if re.search(r"k_wf_trace<\d+, (true|false), true", r["Kernel_Name"]) else "<closest>"
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for kn, a in sorted(agg.items()):
    print("$tag", kn, {k: "%.4g" % v for k, v in sorted(a.items())})
PY
done
