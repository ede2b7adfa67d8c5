#!/bin/bash
# tools/simt_sections.sh <tag> — in-kernel SIMT counters, wave-clock sections and wave occupancy of k_fused on config 2,
# from the three instrumented builds (ab/libs/lib_prof_{SIMT,CYCLES,OCC}.so: the tree built with -DMTR_PROFILE_<x>)
tag=${1:-rXX}
out=gpurun_out/${tag}_fused_simt_sections.txt
mkdir -p gpurun_out
{
echo "# $tag build: in-kernel SIMT counters (-DMTR_PROFILE_SIMT, tools/simt.py 1024), wave-clock sections (-DMTR_PROFILE_CYCLES, tools/cycles.py), wave occupancy (-DMTR_PROFILE_OCC, tools/occ.py); config 2"
echo "== SIMT"
MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_prof_SIMT.so timeout 300 python tools/simt.py 1024 2>&1 | tail -4
echo "== sections"
MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_prof_CYCLES.so timeout 300 python tools/cycles.py 2>&1 | tail -8
echo "== occupancy"
MITRANSIENT_AMD_LIB=$(pwd)/ab/libs/lib_prof_OCC.so timeout 300 python tools/occ.py 2>&1 | tail -2
} > $out
cat $out
