#!/bin/bash
# tools/profile.sh <tag> [bench args...] — kernel-trace stats + PMC passes for bench.py on the GPU box.
# Outputs summaries under gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --truncate-kernels -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$name -o pmc --output-format csv -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, os, collections
out="$OUT"
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read()[:3000])
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        cnt[(k,r["Counter_Name"])]+=1
with open(out+"/pmc_summary.txt","w") as fh:
    for k,v in agg.items():
        fh.write(k+"\n")
        for c,val in sorted(v.items()):
            n=cnt[(k,c)]
            fh.write(f"   {c:28s} total {val:.6g}  dispatches {n}  per-dispatch {val/n:.6g}\n")
print(open(out+"/pmc_summary.txt").read())
import json, re
traffic={}
for k,v in agg.items():
    m=re.search(r"k_(fused|wf_[a-z]+|develop_[a-z]+)", k)
    if not m or "FETCH_SIZE" not in v: continue
    nf=cnt[(k,"FETCH_SIZE")]; nw=cnt.get((k,"WRITE_SIZE"),1)
    fetch_kb=v["FETCH_SIZE"]/nf; write_kb=v.get("WRITE_SIZE",0.0)/max(1,nw)
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md §HBM)
    traffic[m.group(0)]={"fetch_size_kib_per_launch":fetch_kb,"write_size_kib_per_launch":write_kb,
        "hbm_bytes_per_launch":(2.0*fetch_kb+write_kb)*1024.0,"dispatches_profiled":nf,
        "note":"FETCH_SIZE doubled (gfx950 wide-read correction); WRITE_SIZE uncalibrated"}
    if v.get("SQ_ACTIVE_INST_VALU"):
        nv=cnt[(k,"SQ_INSTS_VALU")]
        traffic[m.group(0)].update({"valu_insts_per_launch":v["SQ_INSTS_VALU"]/nv,
            "valu_lanes_per_inst":v["SQ_THREAD_CYCLES_VALU"]/v["SQ_ACTIVE_INST_VALU"],
            "lds_insts_per_launch":v.get("SQ_INSTS_LDS",0.0)/nv, "salu_insts_per_launch":v.get("SQ_INSTS_SALU",0.0)/nv,
            "wait_any_frac":v.get("SQ_WAIT_ANY",0.0)/max(1.0,v.get("SQ_WAVE_CYCLES",1.0)),
            "wait_inst_any_frac":v.get("SQ_WAIT_INST_ANY",0.0)/max(1.0,v.get("SQ_WAVE_CYCLES",1.0)),
            "profile":"$TAG"})
json.dump(traffic, open(out+"/traffic.json","w"), indent=1)
print(json.dumps(traffic, indent=1))
PY
