#!/bin/bash
# round-4 batch Z2: waves per SIMD of the two k_wf_trace instantiations (config 5 at full size)
O=gpurun_out/r4z2; mkdir -p $O
for lib in mitransient_amd/csrc/libmitransient_amd.so ab/exp/libs/lib_any7.so ab/exp/libs/lib_any8.so ab/exp/libs/lib_cl5.so ab/exp/libs/lib_cl4.so mitransient_amd/csrc/libmitransient_amd.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'staircase ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'))
" | tee -a $O/sweep.txt
done
