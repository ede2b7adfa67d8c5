#!/bin/bash
# round 5 batch D: where does the deferred iteration lose?  SIMT counters at 1024 spp, wave-clock sections, occupancy; base = traits kernel
O=gpurun_out/r5d; mkdir -p $O
{
for v in base def; do
  echo "== $v SIMT (1024 spp)"; MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_${v}_simt.so timeout 300 python tools/simt.py 1024 2>&1 | tail -3
  echo "== $v sections"; MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_${v}_cyc.so timeout 300 python tools/cycles.py 2>&1 | tail -7
done
echo "== def occupancy"; MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_def_occ.so timeout 300 python tools/occ.py 2>&1 | tail -1
bash tools/ab.sh ab/libs/lib_base_c2.so ab/libs/lib_def_c2.so
} 2>&1 | tee $O/out.txt
