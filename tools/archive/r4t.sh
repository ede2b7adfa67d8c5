#!/bin/bash
# round-4 batch T: batched k_splat_rows: splat tests, bench at 2^30 / 2^28
O=gpurun_out/r4t; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "splat or film_add" > $O/splat_tests.log 2>&1; echo "pytest rc=$?" >> $O/splat_tests.log); tail -3 $O/splat_tests.log
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
timeout 300 python tools/splat_bench.py 28 2>&1 | tail -6 | tee $O/splat_2p28.txt
