#!/bin/bash
# Builds the C++ host mirror (libpthost.so) and the pt_render CLI against libptgpu.so.
set -euo pipefail
cd "$(dirname "$0")"
CXX=${CXX:-g++}
FLAGS="-O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math"
$CXX $FLAGS -shared -o ../libpthost.so pt_host.cpp -L.. -lptgpu -Wl,-rpath,'$ORIGIN'
$CXX $FLAGS -o ../pt_render pt_render_main.cpp pt_host.cpp -L.. -lptgpu -Wl,-rpath,'$ORIGIN'
echo "built $(cd .. && pwd)/libpthost.so and pt_render"
