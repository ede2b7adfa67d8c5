#!/bin/bash
# tools/ab_env_nlos.sh "<ENV=..>" ... — config 4's share (bench.py --scene nlos) per environment setting (each twice, interleaved): ms per render
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --scene nlos --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', 'ms/step %.3f' % r['ms_per_step'], 'Mray/s %.0f' % r['value'])
"
done
done
