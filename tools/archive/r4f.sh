#!/bin/bash
# round-4 measurement batch F (GPU box): the whole GPU suite, the driver-form bench line, the three profile passes, the splat bench at 2^30
O=gpurun_out/r4f; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash tools/profile_all.sh r04a
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
timeout 300 python bench.py > $O/bench_after.json 2> $O/bench_after.err
