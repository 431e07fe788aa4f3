#!/bin/bash
# Build an experimental variant of the library next to the product one, for
# same-process A/B timing with tools/ab_libs.py.
# usage: tools/build_variant.sh <name> [extra hipcc flags ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${SRC:-$root/landhydrology.jl_amd/csrc}
out=$root/landhydrology.jl_amd/lib/variants
tmp=$(mktemp -d /tmp/lh_variant.XXXXXX)
mkdir -p "$out"
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-function"
for f in "$src"/*.hip; do
    /opt/rocm/bin/hipcc $flags "$@" -c "$f" -o "$tmp/$(basename "$f" .hip).o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/liblandhydro_hip_$name.so" "$tmp"/*.o -L/opt/rocm/lib -lrccl -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
rm -rf "$tmp"
echo "$out/liblandhydro_hip_$name.so"
