#!/bin/bash
# round-4 batch R: GPU suite with anisotropic roughconductors and the partition's edge cases
O=gpurun_out/r4r; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -15 $O/gputests.log
