#!/bin/bash
# round 5 batch I: staged tree top (56 nodes) + shallow-tree rule of MTR_MODE_AUTO: GPU suite, size sweep, config 5 at full size
O=gpurun_out/r5i; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
timeout 1700 python tools/size_sweep.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/size_sweep.txt
python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): r = json.loads(l); print('config 5 full: ms/step %.1f' % r['ms_per_step'], 'k_wf_trace %.1f' % r.get('k_wf_trace_ms_per_step', 0))" | tee $O/c5.txt
