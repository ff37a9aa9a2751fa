#!/bin/bash
# Diagnostic build of libptgpu with per-phase s_memtime stamps -> path_tracer_amd/libptgpu_prof.so
# (used only by tools/; select it with PTGPU_LIB=path_tracer_amd/libptgpu_prof.so).
set -euo pipefail
cd "$(dirname "$0")/../path_tracer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -DPT_PHASE_TIMING -o ../libptgpu_prof.so ptgpu.hip ptgpu_multi.cpp -ldl -pthread
echo built libptgpu_prof.so
