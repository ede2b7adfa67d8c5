#!/bin/bash
# round-4 batch K: pipelined splat scatters: tests, per-kernel times, bench at 2^30
O=gpurun_out/r4k; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "splat or film_add" > $O/splat_tests.log 2>&1; echo "pytest rc=$?" >> $O/splat_tests.log); tail -3 $O/splat_tests.log
bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels.txt
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
