#!/bin/bash
# tools/build_variant.sh <name> [git-ref|-] [extra hipcc flags]
# Builds libmitransient_amd into ab/libs/lib_<name>.so (ab/ is git-ignored but travels to the GPU box), from the working
# tree ("-") or from a git ref (sources extracted to a temporary directory), for A/B runs with tools/ab.sh.
# With -DMTR_ONLY_C2 among the flags only mtr_kernels.hip is recompiled (one k_fused instantiation: the one config 2 runs) and
# linked against objects of the other sources cached under ab/obj/<source hash>/: a minute per variant.
set -e
name=$1; ref=${2:--}; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/ab/libs"
src="$root"
if [ "$ref" != "-" ]; then
  src=$(mktemp -d); (cd "$root" && git archive "$ref" mitransient_amd/csrc include | tar -x -C "$src")
fi
cd "$src/mitransient_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wno-unused-function"
if [[ " $* " == *" -DMTR_ONLY_C2 "* ]]; then
  h=$(cat mtr_api.hip mtr_wavefront.hip mtr_splat.hip mtr_bvh.cpp mtr_scene_host.cpp *.h ../../include/mitransient_amd.h | sha256sum | cut -c1-16)
  od="$root/ab/obj/$h"; mkdir -p "$od"
  for f in mtr_api.hip mtr_wavefront.hip mtr_splat.hip mtr_bvh.cpp mtr_scene_host.cpp; do
    [ -f "$od/$f.o" ] || /opt/rocm/bin/hipcc $FL -c $f -o "$od/$f.o" &
  done
  /opt/rocm/bin/hipcc $FL "$@" -c mtr_kernels.hip -o "$od/kernels_$name.o" &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$root/ab/libs/lib_$name.so" "$od"/mtr_api.hip.o "$od"/mtr_wavefront.hip.o "$od"/mtr_splat.hip.o "$od"/mtr_bvh.cpp.o "$od"/mtr_scene_host.cpp.o "$od/kernels_$name.o"
else
  /opt/rocm/bin/hipcc $FL "$@" -shared -o "$root/ab/libs/lib_$name.so" mtr_api.hip mtr_kernels.hip mtr_wavefront.hip mtr_splat.hip mtr_bvh.cpp mtr_scene_host.cpp
fi
[ "$ref" != "-" ] && rm -rf "$src"
echo "built ab/libs/lib_$name.so"
