#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... — A/B the config-2 render time of kernel variants on the GPU box
for lib in "$@"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib timeout 180 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline']['avg_launch_ms'], 'Mray/s %.0f' % r['value'])
"
done
