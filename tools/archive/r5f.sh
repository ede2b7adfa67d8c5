#!/bin/bash
# round 5 batch F: cold kernel arguments from the kernarg segment — full GPU suite + A/B
O=gpurun_out/r5f; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
bash tools/ab.sh ab/libs/lib_traits.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/ab_c2.txt
