#!/bin/bash
# round-4 batch V: k_fused knobs re-measured after the derived path state (pixels per ticket, row slots in flight)
O=gpurun_out/r4v; mkdir -p $O
run() { lbl=$1; shift
  env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$lbl', 'ms/step %.2f' % r['ms_per_step'], 'kernel %.2f' % r['roofline'].get('avg_launch_ms', 0))
" | tee -a $O/sweep.txt; }
E=$(pwd)/ab/exp/libs
run base MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd.so
for c in 8 16 64 128; do run chunk$c MITRANSIENT_AMD_LIB=$E/lib_exp.so MTR_FUSED_CHUNK=$c; done
run exp MITRANSIENT_AMD_LIB=$E/lib_exp.so
run seg1k MITRANSIENT_AMD_LIB=$E/lib_seg1k.so
run seg4k MITRANSIENT_AMD_LIB=$E/lib_seg4k.so
run base2 MITRANSIENT_AMD_LIB=$(pwd)/mitransient_amd/csrc/libmitransient_amd.so
