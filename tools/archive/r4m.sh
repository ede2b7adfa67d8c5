#!/bin/bash
# round-4 batch M: config 5 at full size, knobs re-measured after the deferred commit (segment size, waves per SIMD, refill threshold)
O=gpurun_out/r4m; mkdir -p $O
run() {  # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py --scene staircase --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg 2>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lbl', 'ms/step %.1f' % r['ms_per_step'], 'trace %.1f' % r.get('k_wf_trace_ms_per_step', 0), 'shade', r.get('roofline_shade', {}).get('kernel_ms_per_render'))
" | tee -a $O/sweep.txt
}
E=$(pwd)/ab/exp/libs
run base MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd.so
for seg in 4096 16384 32768; do run seg$seg MITRANSIENT_AMD_LIB=$E/lib_exp.so MTR_WF_SEG=$seg; done
for v in tw5 tw8 sw3 sw6 rf8 rf32; do run $v MITRANSIENT_AMD_LIB=$E/lib_$v.so; done
run base2 MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd.so
run bins64 MITRANSIENT_AMD_LIB=$E/lib_exp.so MTR_BVH_BINS=64
run leaf3 MITRANSIENT_AMD_LIB=$E/lib_exp.so MTR_BVH_LEAF=3
