#!/bin/bash
# round 5 batch A: scene-trait specialised k_fused (diffuse + one rectangle emitter) — parity suite, then A/B against round 4's library
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
bash tools/ab.sh ab/libs/lib_r4head.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/ab_c2.txt
