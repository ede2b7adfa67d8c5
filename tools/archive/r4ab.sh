#!/bin/bash
# round-4 batch AB3: with all three code removals in, is the coordinate packing still needed?  time and WRITE_SIZE
O=gpurun_out/r4ab; mkdir -p $O
L="ab/exp/libs/lib_nopack.so mitransient_amd/csrc/libmitransient_amd.so"
bash tools/ab.sh $L 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh ab/exp/libs/lib_nopack.so 2>&1 | tee $O/write_size.txt
