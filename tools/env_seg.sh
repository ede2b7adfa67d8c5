#!/bin/bash
for seg in 3072 4096 6144; do
  for args in "--mode wavefront" "--scene staircase --width 720 --height 1280 --bins 400 --spp 64"; do
    MTR_WF_SEG=$seg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('seg $seg', '[$args]', 'ms/step %.2f' % r['ms_per_step'])
"
  done
done
