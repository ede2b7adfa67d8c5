#!/bin/bash
# round-4 batch W: float-reciprocal divisions in the derived path state: parity (fused tests), A/B against HEAD, WRITE_SIZE
O=gpurun_out/r4w; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
bash tools/ab.sh ab/exp/libs/lib_head.so mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh mitransient_amd/csrc/libmitransient_amd.so 2>&1 | tee $O/write_size.txt
