#!/bin/bash
# round-4 batch N (final sources): the whole GPU suite, the driver-form bench line, the three profile passes, the splat bench at 2^30 and its kernel trace
O=gpurun_out/r4n; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
bash tools/profile_all.sh r04b
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
timeout 300 python tools/splat_bench.py 28 2>&1 | tail -6 | tee $O/splat_2p28.txt
bash tools/r4e.sh 2>&1 | head -18 | tee $O/splat_kernels_2p28.txt
timeout 300 python bench.py --scene staircase --steps 3 --warmup 1 > $O/staircase_bench.json 2> $O/staircase.err
timeout 300 python bench.py --scene nlos > $O/nlos_bench.json 2> $O/nlos.err
