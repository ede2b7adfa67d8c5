#!/bin/bash
# round-4 batch G: the GPU suite after the run-table fix, derived-path-state A/B (time + WRITE_SIZE), per-kernel times of the partitioned splat
O=gpurun_out/r4g; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
L="ab/exp/libs/lib_base.so ab/exp/libs/lib_derive.so ab/exp/libs/lib_derive_cam.so ab/exp/libs/lib_derive_cam_le.so mitransient_amd/csrc/libmitransient_amd.so"
bash tools/ab.sh $L 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh $L 2>&1 | tee $O/write_size.txt
bash tools/r4e.sh 2>&1 | head -16 | tee $O/splat_kernels.txt
