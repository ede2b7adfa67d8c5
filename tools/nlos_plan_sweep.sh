#!/bin/bash
# tools/nlos_plan_sweep.sh — config 4's share (bench.py --scene nlos) over the row slots (MTR_FUSED_G) and workgroups per CU (MTR_FUSED_PER_CU)
# of k_fused<NLOS>'s plan (experiments library)
export MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd_exp.so
for env in "MTR_NO_GREY=1" "MTR_FUSED_G=3" "MTR_FUSED_G=2" "MTR_FUSED_G=2 MTR_FUSED_PER_CU=2" "MTR_FUSED_G=1" "MTR_FUSED_G=1 MTR_FUSED_PER_CU=3" "MTR_FUSED_G=1 MTR_FUSED_PER_CU=2"; do
  env $env MTR_FUSED_VERBOSE=1 python bench.py --scene nlos --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2> /tmp/plan.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$env', 'ms/step %.3f' % r['ms_per_step'])
"
  grep fused_plan /tmp/plan.err | tail -1
done
