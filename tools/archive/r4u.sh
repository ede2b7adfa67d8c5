#!/bin/bash
O=gpurun_out/r4u; mkdir -p $O
timeout 900 python tools/soak_splat.py 1 > $O/soak_splat_1.txt 2>&1; grep -v amdgpu.ids $O/soak_splat_1.txt | tail -8
timeout 900 python tools/soak_splat.py 2 > $O/soak_splat_2.txt 2>&1; grep -v amdgpu.ids $O/soak_splat_2.txt | tail -8
