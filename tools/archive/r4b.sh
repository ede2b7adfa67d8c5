#!/bin/bash
# round-4 measurement batch B (GPU box): test suite, kernel-variant A/B, WRITE_SIZE, reserved CUs, splat bench at 2^30
O=gpurun_out/r4b; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
bash tools/ab.sh ab/libs4/lib_base.so ab/libs4/lib_trk.so ab/libs4/lib_noeps.so 2>&1 | tee $O/ab_c2.txt
for lib in ab/libs4/lib_base.so ab/libs4/lib_trk.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'staircase ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'))
" | tee -a $O/ab_c5.txt
done
bash tools/write_size.sh ab/libs4/lib_base.so ab/libs4/lib_trk.so 2>&1 | tee $O/write_size.txt
for k in 0 8 16 32; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs --reserve-cus $k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('reserve_cus $k', 'ms/step %.2f' % r['ms_per_step'])
" | tee -a $O/reserve_cus.txt
done
timeout 900 python tools/splat_bench.py 30 2>&1 | tail -4 | tee $O/splat_2p30.txt
