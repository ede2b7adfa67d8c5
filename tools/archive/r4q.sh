#!/bin/bash
# round-4 batch Q (final sources): the three profile passes, then — the counters merged on the box — the three bench lines
O=gpurun_out/r4q; mkdir -p $O
bash tools/profile_all.sh r04c
bash tools/merge_profiles.sh r04c > $O/merge.log 2>&1; tail -2 $O/merge.log
cp profiles/traffic.json $O/traffic.json
timeout 600 python bench.py > $O/config2_bench.json 2> $O/bench.err
timeout 300 python bench.py --scene staircase --steps 3 --warmup 1 > $O/staircase_bench.json 2> $O/staircase.err
timeout 300 python bench.py --scene nlos > $O/nlos_bench.json 2> $O/nlos.err
tail -c 300 $O/config2_bench.json
