#!/bin/bash
# tools/ab_all.sh lib1.so lib2.so ... — config-2 fused + wavefront and the staircase (720x1280x400, 64 spp) per library
for lib in "$@"; do
  for args in "" "--mode wavefront" "--scene staircase --width 720 --height 1280 --bins 400 --spp 64"; do
    MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', '[$args]', 'ms/step %.2f' % r['ms_per_step'], 'Mray/s %.0f' % r['value'])
"
  done
done
