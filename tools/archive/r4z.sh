#!/bin/bash
# round-4 batch Z: k_wf_trace split by any-hit: wavefront parity tests, config 5 A/B against HEAD
O=gpurun_out/r4z; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -3 $O/gputests.log
for rep in 1 2; do for lib in ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'staircase ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'))
" | tee -a $O/ab_c5.txt
done; done
