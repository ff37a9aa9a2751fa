#!/bin/bash
# Builds libptgpu.so for gfx950 in-tree (path_tracer_amd/libptgpu.so).
# -ffp-contract=off: the kernel reproduces the reference's unfused f32 arithmetic;
# explicit fmaf() is used only in the conservative LDS filter.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function"
$HIPCC $FLAGS "$@" -o ../libptgpu.so ptgpu.hip ptgpu_multi.cpp -ldl -pthread
echo "built $(cd .. && pwd)/libptgpu.so"
