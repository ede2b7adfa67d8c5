#!/bin/bash
# tools/splat_pmc.sh [log2 S = 28] — counters per kernel of tools/splat_bench.py (separate rocprofv3 --pmc passes)
O=$(pwd)/gpurun_out/splat_pmc; rm -rf $O; mkdir -p $O
R=$(pwd)
i=0
for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc --output-format csv -- python $R/tools/splat_bench.py ${1:-28} > $O/log$i.txt 2>&1 )
done
python - <<PY
import csv, glob
tot={}; n={}
for f in glob.glob("$O/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]); tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
names=sorted({k[0] for k in tot})
for nm in names:
    if "part" in nm or "splat" in nm:
        g=lambda c: tot.get((nm,c),0)/max(n.get((nm,c),1),1)
        print("%-36s n %3d FETCHx2 %.2f GB WRITE %.2f GB L2hit %.0f%% | VALU %.3g LDS %.3g VMEM %.3g bankconf/ldsact %.2f | wait_any %.2f wait_inst %.2f active_any %.2f (of wave cycles %.3g)" % (
            nm, n.get((nm,"WRITE_SIZE"),0), 2*g("FETCH_SIZE")*1024/1e9, g("WRITE_SIZE")*1024/1e9, 100*g("TCC_HIT_sum")/max(g("TCC_HIT_sum")+g("TCC_MISS_sum"),1),
            g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM"), g("SQ_LDS_BANK_CONFLICT")/max(g("SQ_ACTIVE_INST_LDS"),1),
            g("SQ_WAIT_ANY")/max(g("SQ_WAVE_CYCLES"),1), g("SQ_WAIT_INST_ANY")/max(g("SQ_WAVE_CYCLES"),1), g("SQ_ACTIVE_INST_ANY")/max(g("SQ_WAVE_CYCLES"),1), g("SQ_WAVE_CYCLES")))
PY
