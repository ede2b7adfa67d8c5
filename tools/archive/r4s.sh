#!/bin/bash
# round-4 batch S2 (final sources): the three profile passes, the counters merged on the box, the three bench lines, the splat bench, the 1-rank RCCL path
O=gpurun_out/r4s; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -3 $O/gputests.log
bash tools/profile_all.sh r04i
bash tools/merge_profiles.sh r04i > $O/merge.log 2>&1; tail -2 $O/merge.log
cp profiles/traffic.json $O/traffic.json
timeout 600 python bench.py > $O/config2_bench.json 2> $O/bench.err
timeout 300 python bench.py --scene staircase --steps 3 --warmup 1 > $O/staircase_bench.json 2> $O/staircase.err
timeout 300 python bench.py --scene nlos > $O/nlos_bench.json 2> $O/nlos.err
timeout 600 python tools/splat_bench.py 30 2>&1 | tail -6 | tee $O/splat_2p30.txt
timeout 300 python tools/rccl_one_rank.py > $O/rccl_one_rank.txt 2>&1; tail -3 $O/rccl_one_rank.txt
tail -c 300 $O/config2_bench.json
timeout 300 python bench.py --scene staircase --materials rough --steps 2 --warmup 1 --no-cpu-baseline --no-scatter-leg > $O/staircase_rough_bench.json 2> $O/staircase_rough.err
