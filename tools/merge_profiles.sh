#!/bin/bash
# tools/merge_profiles.sh <tag> — after `gpurun -- bash tools/profile_all.sh <tag>`: merge the three counter fragments into
# profiles/traffic.json and copy the summaries the judge reads into profiles/<round>_*   usage: tools/merge_profiles.sh <tag> [round = r04]
TAG=$1; RND=${2:-r04}
python tools/pmc_summary.py --merge gpurun_out/prof_${TAG}_c2/traffic.json
python tools/pmc_summary.py --merge gpurun_out/prof_${TAG}_c5/traffic.json --section staircase
python tools/pmc_summary.py --merge gpurun_out/prof_${TAG}_c4/traffic.json --section nlos
python tools/pmc_summary.py --merge gpurun_out/prof_${TAG}_c5r/traffic.json --section staircase_rough
for c in c2 c5 c4 c5r; do
  cp gpurun_out/prof_${TAG}_$c/pmc_summary.txt profiles/${RND}_${c}_pmc_summary.txt
  cp gpurun_out/prof_${TAG}_$c/trace/trace_kernel_stats.csv profiles/${RND}_${c}_kernel_stats.csv
done
python -c "
import json, bench
t = json.load(open('profiles/traffic.json'))
print('sources', bench.source_hash(), '| config 2', t['source_hash'], '| staircase', t['staircase']['source_hash'], '| nlos', t['nlos']['source_hash'], '| staircase_rough', t.get('staircase_rough', {}).get('source_hash'))"
