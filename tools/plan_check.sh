#!/bin/bash
# tools/plan_check.sh — the residency-first plan of k_fused against the rule of rounds 1 - 5 (MTR_FUSED_OLD_PLAN=1; experiments library)
export MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd_exp.so
run() {  # <env> <bench args>
  env $1 MTR_FUSED_VERBOSE=1 python bench.py $2 --steps 5 --warmup 2 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2> /tmp/plan.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$1 | $2 |', 'ms/step %.3f' % r['ms_per_step'])
"
  grep fused_plan /tmp/plan.err | tail -1
}
for a in "--bins 128 --spp 256" "--bins 2048 --spp 256" "--spp 256" "" "--scene nlos"; do
  run "MTR_FUSED_OLD_PLAN=1" "$a"; run "X=0" "$a"
done
for a in "--spp 256" ""; do
  run "MTR_BENCH_DETERMINISTIC=1 MTR_FUSED_OLD_PLAN=1" "$a"; run "MTR_BENCH_DETERMINISTIC=1" "$a"
done
