#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
for rep in 1 2; do
timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>$O/stair.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('staircase ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'), 'scatter', r.get('scatter_add', {}).get('kernel_ms_per_render'))
" | tee -a $O/c5.txt
done
tail -3 $O/stair.err
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('config2 ms/step %.2f' % r['ms_per_step'])
" | tee -a $O/c2.txt
