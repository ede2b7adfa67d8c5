#!/bin/bash
# round-4 batch L: config 5 with QNode8 padded to one 128-byte line per node (A/B)
O=gpurun_out/r4l; mkdir -p $O
for rep in 1 2; do
for lib in mitransient_amd/csrc/libmitransient_amd.so ab/exp/libs/lib_q8pad.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'staircase ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'))
" | tee -a $O/ab_c5.txt
done
done
