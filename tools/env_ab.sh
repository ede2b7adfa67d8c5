#!/bin/bash
# tools/env_ab.sh VAR v1 v2 ... : A/B an environment knob
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$VAR=$v', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline']['avg_launch_ms'], 'Mray/s %.0f' % r['value'])
"
done
