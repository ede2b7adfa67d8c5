#!/bin/bash
# tools/ab_env_c2.sh "<ENV=..>" ... — config 2 per environment setting
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline'].get('avg_launch_ms', 0))
"
done
done
