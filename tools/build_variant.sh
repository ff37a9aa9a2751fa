#!/bin/bash
# Experimental build of libptgpu with extra compiler flags -> path_tracer_amd/libptgpu_<name>.so
# usage: tools/build_variant.sh <name> [extra hipcc flags...]; select with PTGPU_LIB=path_tracer_amd/libptgpu_<name>.so
# (timing experiments only: tools/perf_probe.py; never loaded by tests, smoke() or bench.py).
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../path_tracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -o ../libptgpu_${name}.so ptgpu.hip ptgpu_multi.cpp -ldl -pthread
echo built libptgpu_${name}.so
