#!/bin/bash
# builds build/gemm_x2_tune (2 x fp16 split NT / TN GEMMs against the 3 x bf16 ones); run from the repo root
# X2_EXTRA=-DX2_PROBE_3M X2_TAG=_3m: the 3-multiplication timing probe as build/gemm_x2_tune_3m
set -e
mkdir -p build
make -C wire_amd/csrc -j8 >/dev/null
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -c tools/wire_gemmx3g.hip -o build/wire_gemmx3g.o
# the 2 x fp16 kernels once more with the tiled-A addressing probe compiled in (harness only)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWIRE_X2_TILED_PROBE -DWIRE_X2_EXPERIMENTS $X2_EXTRA -c tools/wire_gemmx2h_probe.hip -o build/wire_gemmx2h_probe$X2_TAG.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_x2_tune.hip -o build/gemm_x2_tune.o
hipcc --offload-arch=gfx950 build/gemm_x2_tune.o build/csrc/wire_gemm.o build/csrc/wire_gemm3m.o build/csrc/wire_gemmx3.o \
      build/wire_gemmx3g.o build/csrc/wire_gemmx3h.o build/wire_gemmx2h_probe$X2_TAG.o -o build/gemm_x2_tune$X2_TAG
