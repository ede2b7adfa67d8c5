#!/bin/bash
# round 5 batch C: deferred iteration (two rays per lane in one walk) in k_fused — parity with the variant library, then A/B
O=gpurun_out/r5c; mkdir -p $O
MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_def.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
bash tools/ab.sh ab/libs/lib_traits.so ab/libs/lib_def.so 2>&1 | tee $O/ab_c2.txt
