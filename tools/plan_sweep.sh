#!/bin/bash
# tools/plan_sweep.sh — k_fused's plan (row slots G against workgroups per CU) on path-tracing workloads: config 2's scene at other bin counts, and
# with deterministic rows (experiments library: MTR_FUSED_G, MTR_FUSED_PER_CU, MTR_FUSED_VERBOSE)
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
run "X=0" "--bins 128 --spp 256"
run "MTR_FUSED_PER_CU=4" "--bins 128 --spp 256"
run "X=0" "--bins 2048 --spp 256"
run "MTR_FUSED_G=1" "--bins 2048 --spp 256"
run "X=0" "--bins 4096 --spp 256"
run "MTR_BENCH_DETERMINISTIC=1" "--spp 256"
run "MTR_BENCH_DETERMINISTIC=1 MTR_FUSED_G=1" "--spp 256"
run "MTR_BENCH_DETERMINISTIC=1 MTR_FUSED_G=1 MTR_FUSED_PER_CU=2" "--spp 256"
