#!/bin/bash
# builds build/gemm_tune (native GEMM A/B harness); run from the repo root
set -e
mkdir -p build
make -C wire_amd/csrc -j4 >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -c tools/gemm_tune.hip -o build/gemm_tune.o   # as wire_gemmx3.hip is built
hipcc --offload-arch=gfx950 build/gemm_tune.o build/csrc/wire_point.o -o build/gemm_tune
