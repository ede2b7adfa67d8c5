#!/bin/bash
# round 5 batch E: single advance per walk iteration + kTrLeafPair — full GPU suite, A/B against the traits-only library
O=gpurun_out/r5e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
bash tools/ab.sh ab/libs/lib_traits.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/ab_c2.txt
python bench.py --scene nlos --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): r = json.loads(l); print('nlos ms/step %.3f' % r['ms_per_step'])" | tee $O/nlos.txt
python bench.py --mode wavefront --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): r = json.loads(l); print('c2 wavefront ms/step %.2f' % r['ms_per_step'])" | tee $O/wf.txt
