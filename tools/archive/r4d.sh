#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "splat" > $O/splat_tests.log 2>&1; echo "pytest rc=$?" >> $O/splat_tests.log); tail -3 $O/splat_tests.log
timeout 200 python tools/splat_bench.py 28 2>&1 | tail -6 | tee $O/splat_2p28.txt
timeout 400 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt; bash tools/r4e.sh | head -14
