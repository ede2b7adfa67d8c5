#!/bin/bash
# round-4 batch Y: derived path state in the NLOS loop: GPU suite, config 4's share A/B against HEAD (time, HBM write counter)
O=gpurun_out/r4y; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
for rep in 1 2; do for lib in ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 200 python bench.py --scene nlos --steps 20 --warmup 3 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'nlos ms/step %.3f' % r['ms_per_step'], 'kernel %.3f' % r['roofline'].get('avg_launch_ms', 0))
" | tee -a $O/ab_c4.txt
done; done
REPO=$(pwd)
for lib in ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so; do
  OUT=$REPO/gpurun_out/ws4_$(basename $lib .so); rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && MITRANSIENT_AMD_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc -o pmc --output-format csv -- python $REPO/bench.py --scene nlos --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg > $OUT/log 2>&1 )
  python - <<PY | tee -a $O/write_size_c4.txt
import csv, glob
tot={}; n={}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused" in r["Kernel_Name"]:
            k=r["Counter_Name"]; tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
print("$lib", {k: "%.3g" % (tot[k]/n[k]) for k in tot})
PY
done
