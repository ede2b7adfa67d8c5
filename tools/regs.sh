#!/bin/bash
# tools/regs.sh [file.hip] [pattern] — VGPRs / spills / occupancy per kernel (cross-compiles, no GPU needed)
f=${1:-mtr_kernels.hip}; pat=${2:-.}
cd "$(dirname "$0")/../mitransient_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function $EXTRA \
  -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c "
import sys, re
cur = None; rows = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', l)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    if re.search('$pat', k): print('%-70s vgpr %3d spill %3d scratch %4d occ %d' % (k[:70], r.get('VGPRs', -1), r.get('VGPRs Spill', -1), r.get('ScratchSize', -1), r.get('Occupancy', -1)))
"
