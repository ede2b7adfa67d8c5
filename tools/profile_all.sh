#!/bin/bash
# tools/profile_all.sh <round-tag> — the profile passes behind profiles/traffic.json: config 2 (k_fused + the wavefront leg) and
# config 5 (staircase, one full-size render per pass), config 4's per-GPU share (NLOS, k_fused<NLOS>).  Afterwards, in the authoring container:
#   python tools/pmc_summary.py --merge gpurun_out/prof_<tag>_c2/traffic.json
#   python tools/pmc_summary.py --merge gpurun_out/prof_<tag>_c5/traffic.json --section staircase
#   python tools/pmc_summary.py --merge gpurun_out/prof_<tag>_c4/traffic.json --section nlos
TAG=$1
tools/profile.sh ${TAG}_c2 > gpurun_out/prof_${TAG}_c2.log 2>&1
STEPS=1 WARMUP=0 RENDERS=1 tools/profile.sh ${TAG}_c5 --scene staircase --no-scatter-leg > gpurun_out/prof_${TAG}_c5.log 2>&1
tools/profile.sh ${TAG}_c4 --scene nlos --no-scatter-leg > gpurun_out/prof_${TAG}_c4.log 2>&1
# ... and config 5 as its file describes it (GGX lobes, vertex normals, bitmaps): the "staircase_rough" section
STEPS=1 WARMUP=0 RENDERS=1 tools/profile.sh ${TAG}_c5r --scene staircase --materials rough --no-scatter-leg > gpurun_out/prof_${TAG}_c5r.log 2>&1
tail -n 3 gpurun_out/prof_${TAG}_c5r.log
tail -n 3 gpurun_out/prof_${TAG}_c2.log; tail -n 3 gpurun_out/prof_${TAG}_c5.log; tail -n 3 gpurun_out/prof_${TAG}_c4.log
