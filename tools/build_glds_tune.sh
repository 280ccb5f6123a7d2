#!/bin/bash
# builds build/gemm_glds_tune (A/B of the split-bf16 NT GEMM editions); run from the repo root
set -e
mkdir -p build
make -C wire_amd/csrc -j4 >/dev/null
# the LDS-DMA kernels once more with their ablation switches compiled in (harness only)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWIRE_ABLATE_G -c tools/wire_gemmx3g.hip -o build/wire_gemmx3g_abl.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_glds_tune.hip -o build/gemm_glds_tune.o
hipcc --offload-arch=gfx950 build/gemm_glds_tune.o build/csrc/wire_gemm.o build/csrc/wire_gemm3m.o build/csrc/wire_gemmx3.o \
      build/wire_gemmx3g_abl.o build/csrc/wire_gemmx3h.o -o build/gemm_glds_tune
# probe: the ablation build once more with 16x16x32 MFMAs in place of 32x32x16 (timing only)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWIRE_ABLATE_G -DWIRE_ABLATE_G16 -c tools/wire_gemmx3g.hip -o build/wire_gemmx3g_abl16.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWIRE_PROBE16 -c wire_amd/csrc/wire_gemmx3.hip -o build/wire_gemmx3_p16.o
hipcc --offload-arch=gfx950 build/gemm_glds_tune.o build/csrc/wire_gemm.o build/csrc/wire_gemm3m.o build/wire_gemmx3_p16.o \
      build/wire_gemmx3g_abl16.o build/csrc/wire_gemmx3h.o -o build/gemm_glds_tune16
