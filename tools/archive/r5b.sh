#!/bin/bash
# round 5 batch B: full GPU suite of the current tree (ADVICE fixes, TEA64 seeding flag, lagged live-count polling) + the driver-form bench line
O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r5b/bench.json') if l.startswith('{')][-1])
print('ms/step', r['ms_per_step'], 'value', r['value'])
print('splat_microbench', json.dumps(r.get('splat_microbench'))[:1500])
e = r.get('extra_configs', {})
for k in e:
    if isinstance(e[k], dict): print(k, e[k].get('ms'), e[k].get('mode'))
print('scatter', r.get('scatter_add', {}).get('render_ms_wavefront'))
PY
