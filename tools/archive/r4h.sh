#!/bin/bash
# round-4 batch H: GPU suite with the packed derived path state + LDS-staged splat scatters; A/B of the k_fused variants; splat bench
O=gpurun_out/r4h; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
L="ab/exp/libs/lib_base.so ab/exp/libs/lib_derive.so ab/exp/libs/lib_v3.so mitransient_amd/csrc/libmitransient_amd.so"
bash tools/ab.sh $L 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh ab/exp/libs/lib_v3.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/write_size.txt
bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels.txt
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
