#!/bin/bash
O=$(pwd)/gpurun_out/r4e; mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -d $O/trace -o trace --output-format csv -- python $R/tools/splat_bench.py 28 > $O/log.txt 2>&1
cd $R
python - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), "avg %.3f ms" % (float(r["AverageNs"]) / 1e6), "total %.1f ms" % (float(r["TotalDurationNs"]) / 1e6))
PY
