// gemm_tune.cpp -- native A/B harness for the fp32-MFMA GEMM kernels (no Python,
// no torch): times launch_gemm_nt / launch_gemm_tn variants with HIP events,
// interleaved rounds in ONE process (cdna guide rule 24), random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_tune.cpp build/csrc/wire_gemm.o \
//         build/csrc/wire_point.o -o build/gemm_tune && ./build/gemm_tune [N] [P]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define WIRE_ABLATE 1
#include "../wire_amd/csrc/wire_gemm.hip"   // compiled in with the ablation hooks

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
static float* dalloc(size_t n, unsigned seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 262144;
  const int P = argc > 2 ? atoi(argv[2]) : 512;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  float* A = dalloc((size_t)N * P, 1, 1.0f);
  float* Bt = dalloc((size_t)P * P, 2, 0.06f);
  float* bias = dalloc(P, 3, 0.06f);
  float* lin = dalloc((size_t)N * P, 4, 0.5f);
  float* out = dalloc((size_t)N * P, 5, 1.0f);
  float* o0 = dalloc((size_t)N * P, 6, 0.f);
  float* o1 = dalloc((size_t)N * P, 7, 0.f);
  const int S = gemm_tn_splits(N, P, P, 64);
  float* slab = dalloc((size_t)S * P * P, 8, 0.f);
  float* bslab = dalloc((size_t)S * P, 9, 0.f);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;

  struct Var { const char* name; int kind; int epi; int bk; int abl = 0; };
  std::vector<Var> vars = {
      {"nt store      bk32", 0, EPI_STORE, 32},     {"nt store      bk16", 0, EPI_STORE, 16},
      {"nt gabor_fwd  bk32", 0, EPI_GABOR_FWD, 32}, {"nt gabor_fwd  bk16", 0, EPI_GABOR_FWD, 16},
      {"nt gabor_bwd  bk32", 0, EPI_GABOR_BWD, 32}, {"nt gabor_bwd  bk16", 0, EPI_GABOR_BWD, 16},
      {"nt relu_fwd   bk16", 0, EPI_RELU_FWD, 16},  {"tn wgrad", 1, 0, 0},
      {"store bk32 noglobal", 0, EPI_STORE, 32, 1}, {"store bk32 nog+nolds", 0, EPI_STORE, 32, 3},
      {"store bk32 nobarrier", 0, EPI_STORE, 32, 4}, {"store bk32 mfma only", 0, EPI_STORE, 32, 7},
      {"store bk16 noglobal", 0, EPI_STORE, 16, 1}, {"store bk16 mfma only", 0, EPI_STORE, 16, 7},
      {"store bk16 prio", 0, EPI_STORE, 16, 8}, {"store bk32 prio", 0, EPI_STORE, 32, 8},
      {"gabor_fwd bk16 prio", 0, EPI_GABOR_FWD, 16, 8}, {"gabor_fwd bk32 prio", 0, EPI_GABOR_FWD, 32, 8},
      {"gabor_bwd bk16 prio", 0, EPI_GABOR_BWD, 16, 8},
  };
  std::vector<double> best(vars.size(), 1e30), sum(vars.size(), 0);
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t v = 0; v < vars.size(); ++v) {
      const Var& V = vars[v];
      GemmEpiParams ep;
      ep.bias = bias; ep.o0 = o0; ep.o1 = o1; ep.i0 = lin; ep.i1 = out; ep.ld0 = P; ep.ld1 = P;
      ep.omega = 20.f; ep.scale = 30.f; ep.kvalid = P / 2; ep.ablate = V.abl;
      if (V.epi == EPI_RELU_FWD) ep.kvalid = P;
      CK(hipEventRecord(e0, 0));
      if (V.kind == 0) {
        gemm_tune_set("nt_bk", V.bk);
        CK(launch_gemm_nt(0, V.epi, A, P, Bt, P, N, P, P, ep));
      } else {
        CK(launch_gemm_tn(0, A, P, out, P, N, P, P, S, slab, bslab));
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
    }
  }
  printf("N=%lld P=%d  (%.1f GFLOP per launch; fp32 MFMA peak 157.3 TF)\n", (long long)N, P, flop / 1e9);
  for (size_t v = 0; v < vars.size(); ++v)
    printf("  %-20s mean %7.3f ms  min %7.3f ms  -> %6.1f TF (%.1f%% of peak)\n", vars[v].name,
           sum[v] / rounds, best[v], flop / (sum[v] / rounds * 1e-3) / 1e12,
           100.0 * flop / (sum[v] / rounds * 1e-3) / 1e12 / 157.3);
  return 0;
}
