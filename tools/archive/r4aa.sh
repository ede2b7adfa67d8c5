#!/bin/bash
# round-4 batch AA: k_fused with the unused walkers of traverse() folded away (assume), 8-wide-only k_wf_trace: GPU suite, config 2 / 4 / 5 A/B against HEAD, WRITE_SIZE
O=gpurun_out/r4aa; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -3 $O/gputests.log
bash tools/ab.sh ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/ab_c2.txt
for lib in ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 200 python bench.py --scene nlos --steps 20 --warmup 3 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'nlos ms/step %.3f' % r['ms_per_step'], 'kernel %.3f' % r['roofline'].get('avg_launch_ms', 0))
" | tee -a $O/ab_c4.txt
done
bash tools/write_size.sh mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/write_size.txt
