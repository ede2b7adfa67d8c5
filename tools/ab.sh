#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... — A/B the config-2 render time of kernel variants on the GPU box (each twice, interleaved)
for rep in 1 2; do
for lib in "$@"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 180 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline'].get('avg_launch_ms', 0), 'Mray/s %.0f' % r['value'])
"
done
done
