#!/bin/bash
# round-4 batch P: GPU suite with the Beckmann lobes; config 2 / config 5 timing check; rough staircase bench
O=gpurun_out/r4p; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc=$?" >> $O/gputests.log); tail -4 $O/gputests.log
timeout 600 python bench.py --no-cpu-baseline > $O/config2_bench.json 2> $O/bench.err
python - <<PY
import json
for l in open("$O/config2_bench.json"):
    if l.startswith("{"):
        r = json.loads(l); print("config2 ms/step %.2f kernel %.2f" % (r["ms_per_step"], r["roofline"].get("avg_launch_ms", 0)), {k: (v.get("ms_per_step"), v.get("roofline", {}).get("kernel_ms_per_render")) for k, v in r.get("extra_configs", {}).items() if isinstance(v, dict)})
PY
