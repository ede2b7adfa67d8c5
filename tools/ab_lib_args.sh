#!/bin/bash
# tools/ab_lib_args.sh "<bench args>" lib.so ... — one bench.py invocation per library variant (each twice, interleaved)
args=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'k_wf_trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'Mray/s %.0f' % r['value'])
"
done
done
