#!/bin/bash
# tools/bench_stair_small.sh [lib ...]: config-5 geometry at the profile size (720x1280 x 400 bins x 64 spp), wavefront auto mode
for lib in "${@:-mitransient_amd/csrc/libmitransient_amd.so}"; do
  MITRANSIENT_AMD_LIB=$(pwd)/$lib python bench.py --scene staircase --width 720 --height 1280 --bins 400 --spp 64 --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lib', 'ms/step %.2f' % r['ms_per_step'], 'Mray/s %.0f' % r['value'])
"
done
