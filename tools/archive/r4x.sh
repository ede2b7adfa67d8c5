#!/bin/bash
# round-4 batch X: what the driver runs at round end, in its order: smoke, the GPU suite, the bench line
O=gpurun_out/r4x; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
(timeout 1800 python -m pytest tests/ -x -q -m gpu > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -3 $O/gputests.log
SECONDS=0; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall: $SECONDS s"; tail -c 200 $O/bench.json
