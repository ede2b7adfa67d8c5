#!/bin/bash
# round-4 batch J: splat tests + per-kernel times (parallel scan, tile of 2048 as a variant), splat counters, reserved-CU sweep
O=gpurun_out/r4j; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "splat or film_add" > $O/splat_tests.log 2>&1; echo "pytest rc=$?" >> $O/splat_tests.log); tail -3 $O/splat_tests.log
bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels.txt
MITRANSIENT_AMD_LIB=$(pwd)/ab/exp/libs/lib_stage8.so bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels_stage8.txt
bash tools/splat_pmc.sh 28 2>&1 | tee $O/splat_pmc.txt
for k in 0 8 16 32; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-scatter-leg --no-extra-configs --reserve-cus $k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('reserve_cus $k', 'ms/step %.2f' % r['ms_per_step'])
" | tee -a $O/reserve_cus.txt
done
