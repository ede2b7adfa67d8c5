#!/bin/bash
# tools/ab_stair.sh lib1.so ... — config 5 at 256 spp per library variant (each twice, interleaved)
for rep in 1 2; do
for lib in "$@"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py --scene staircase --spp ${SPP:-256} --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.1f' % r['ms_per_step'], 'k_wf_trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'Mray/s %.0f' % r['value'])
"
done
done
