#!/bin/bash
# tools/ab_env_stair.sh "<ENV=..>" ... — config 5 (reduced to 256 spp) per environment setting: ms per render, k_wf_trace ms
for v in "$@"; do
  env $v python bench.py --scene staircase --spp ${SPP:-256} --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', 'ms/step %.1f' % r['ms_per_step'], 'k_wf_trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'Mray/s %.0f' % r['value'], r['counters_per_step']['rays_closest'])
"
done
