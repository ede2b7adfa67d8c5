#!/bin/bash
# tools/build_variant.sh <name> [git-ref|-] [extra hipcc flags]
# Builds libmitransient_amd into ab/libs/lib_<name>.so (ab/ is git-ignored but travels to the GPU box), from the working
# tree ("-") or from a git ref (sources extracted to a temporary directory), for A/B runs with tools/ab.sh.
set -e
name=$1; ref=${2:--}; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/ab/libs"
src="$root"
if [ "$ref" != "-" ]; then
  src=$(mktemp -d); (cd "$root" && git archive "$ref" mitransient_amd/csrc include | tar -x -C "$src")
fi
cd "$src/mitransient_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function "$@" \
  -shared -o "$root/ab/libs/lib_$name.so" mtr_api.hip mtr_kernels.hip mtr_wavefront.hip mtr_splat.hip mtr_bvh.cpp mtr_scene_host.cpp
[ "$ref" != "-" ] && rm -rf "$src"
echo "built ab/libs/lib_$name.so"
