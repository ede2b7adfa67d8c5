#!/bin/bash
# round 5 batch G: the size sweep (tools/size_sweep.py)
O=gpurun_out/r5g; mkdir -p $O
timeout 1700 python tools/size_sweep.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/size_sweep.txt
