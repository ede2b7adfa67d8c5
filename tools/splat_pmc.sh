#!/bin/bash
# tools/splat_pmc.sh [log2 S = 28] — HBM counters per kernel of tools/splat_bench.py (one rocprofv3 --pmc pass)
O=$(pwd)/gpurun_out/splat_pmc; rm -rf $O; mkdir -p $O
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc -o pmc --output-format csv -- python $R/tools/splat_bench.py ${1:-28} > $O/log.txt 2>&1 )
python - <<PY
import csv, glob
tot={}; n={}
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]); tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
names=sorted({k[0] for k in tot})
for nm in names:
    if "part" in nm or "splat" in nm:
        g=lambda c: tot.get((nm,c),0)/max(n.get((nm,c),1),1)
        print("%-42s launches %3d  FETCH %.3f GB (x2 corr: %.3f)  WRITE %.3f GB  L2 hit %.1f%%" % (nm, n.get((nm,"WRITE_SIZE"),0), g("FETCH_SIZE")*1024/1e9, 2*g("FETCH_SIZE")*1024/1e9, g("WRITE_SIZE")*1024/1e9, 100*g("TCC_HIT_sum")/max(g("TCC_HIT_sum")+g("TCC_MISS_sum"),1)))
PY
