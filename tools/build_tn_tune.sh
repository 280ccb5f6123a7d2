#!/bin/bash
# builds build/gemm_tn_tune (A/B of the split-bf16 weight-gradient kernels); run from the repo root
set -e
mkdir -p build
make -C wire_amd/csrc -j4 >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -c tools/wire_gemmx3g.hip -o build/wire_gemmx3g.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_tn_tune.hip -o build/gemm_tn_tune.o
hipcc --offload-arch=gfx950 build/gemm_tn_tune.o build/csrc/wire_gemm.o build/csrc/wire_gemm3m.o build/csrc/wire_gemmx3.o \
      build/wire_gemmx3g.o build/csrc/wire_gemmx3h.o -o build/gemm_tn_tune
# the same harness on a build with the weight-gradient kernel's ablation switches (WIRE_TN_ABL=1|2|4|8, timing only)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWIRE_ABLATE_TN -c wire_amd/csrc/wire_gemmx3.hip -o build/wire_gemmx3_abltn.o
hipcc --offload-arch=gfx950 build/gemm_tn_tune.o build/csrc/wire_gemm.o build/csrc/wire_gemm3m.o build/wire_gemmx3_abltn.o \
      build/wire_gemmx3g.o build/csrc/wire_gemmx3h.o -o build/gemm_tn_tune_abl
