#!/bin/bash
# round-4 batch I: GPU suite (pure-derive k_fused, batched row kernel), A/B incl. the power-of-two fastdiv, splat counters and bench
O=gpurun_out/r4i; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
L="ab/exp/libs/lib_base.so ab/exp/libs/lib_pow2.so mitransient_amd/csrc/libmitransient_amd.so"
bash tools/ab.sh $L 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh ab/exp/libs/lib_pow2.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/write_size.txt
bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels.txt
bash tools/splat_pmc.sh 28 2>&1 | tee $O/splat_pmc.txt
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
