#!/bin/bash
# tools/ab_wf.sh lib... — config 2 in the wavefront organisation per library variant (each twice)
for rep in 1 2; do
for lib in "$@"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs --mode wavefront 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'Mray/s %.0f' % r['value'])
"
done
done
