#!/bin/bash
# round-4 batch AB2: light forms of the derived path state (film coordinates packed in one register; + the row slot from q): time and WRITE_SIZE
O=gpurun_out/r4ab; mkdir -p $O
L="ab/exp/libs/lib_noderive.so ab/exp/libs/lib_xy.so ab/exp/libs/lib_xyslot.so mitransient_amd/csrc/libmitransient_amd.so"
bash tools/ab.sh $L 2>&1 | tee $O/ab_c2.txt
bash tools/write_size.sh ab/exp/libs/lib_xy.so ab/exp/libs/lib_xyslot.so 2>&1 | tee $O/write_size.txt
