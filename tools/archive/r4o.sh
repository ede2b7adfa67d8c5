#!/bin/bash
# round-4 batch O: the three bench lines after profiles/traffic.json was merged (same sources: the counter-based roofline blocks are printed)
O=gpurun_out/r4o; mkdir -p $O
timeout 600 python bench.py > $O/config2_bench.json 2> $O/bench.err
timeout 300 python bench.py --scene staircase --steps 3 --warmup 1 > $O/staircase_bench.json 2> $O/staircase.err
timeout 300 python bench.py --scene nlos > $O/nlos_bench.json 2> $O/nlos.err
tail -c 400 $O/config2_bench.json
