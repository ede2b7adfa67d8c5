#!/bin/bash
# tools/regs_c2.sh [extra flags] — VGPRs / spills / scratch of the ONE k_fused instantiation config 2 runs (cross-compiles, ~30 s)
cd "$(dirname "$0")/../mitransient_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function -DMTR_ONLY_C2 "$@" \
  -Rpass-analysis=kernel-resource-usage -c mtr_kernels.hip -o /dev/null 2>&1 | grep -A12 "k_fusedILb1ELb1ELb0ELi4ELb0ELb0ELb0ELj15E" | grep -E "VGPRs:|Spill|Scratch|Occupancy" | tr '\n' ' '; echo
