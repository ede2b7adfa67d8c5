#!/bin/bash
# tools/ab_tree.sh "<bench args>" tree1 tree2 ... — the same bench.py invocation in several checkouts of the repository (each with its own
# library AND Python layer: for A/Bs across an ABI change), each twice, interleaved.  tree "." = this tree.
args=$1; shift
for rep in 1 2; do
for t in "$@"; do
  (cd $t && python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$t', 'ms/step %.2f' % r['ms_per_step'], 'k_wf_trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'k_wf_shade %.1f' % r.get('k_wf_shade_ms_per_step', 0), 'Mray/s %.0f' % r['value'])
")
done
done
