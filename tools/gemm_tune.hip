// gemm_tune.hip -- native A/B harness for the fp32-MFMA GEMM kernels (no Python,
// no torch): times kernel variants with HIP events, interleaved rounds in ONE
// process (cdna guide rule 24), random operands.  The kernels are compiled in
// with WIRE_ABLATE so single costs (global loads, LDS writes, barriers) can be
// switched off (results are then wrong; only the timing matters).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_tune.hip -o build/gemm_tune.o
//   hipcc --offload-arch=gfx950 build/gemm_tune.o build/csrc/wire_point.o -o build/gemm_tune
//   ./build/gemm_tune [N] [P] [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define WIRE_ABLATE 1
#include "../wire_amd/csrc/wire_gemm.hip"     // 4-multiplication kernels
#include "../wire_amd/csrc/wire_gemm3m.hip"   // 3-multiplication kernels

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
  const int Kp = P / 2;
  float* A = dalloc((size_t)N * P, 1, 1.0f);
  float* Bt = dalloc((size_t)P * P, 2, 0.06f);
  float* bias = dalloc(P, 3, 0.06f);
  float* lin = dalloc((size_t)N * P, 4, 0.5f);
  float* out = dalloc((size_t)N * P, 5, 1.0f);
  float* o0 = dalloc((size_t)N * P, 6, 0.f);
  float* o1 = dalloc((size_t)N * P, 7, 0.f);
  const int S = gemm_tn_splits(N, P, P, 64);
  const int S3 = gemm3m_tn_splits(N, Kp, Kp, 64);
  float* slab = dalloc((size_t)64 * P * P, 8, 0.f);
  float* bslab = dalloc((size_t)64 * P, 9, 0.f);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;    // algorithmic: 8 flop per complex MAC

  // kind: 0 = 4M NT, 1 = 4M TN, 2 = 3M NT, 3 = 3M TN
  struct Var { const char* name; int kind; int epi; int bk; int abl; };
  std::vector<Var> vars = {
      {"4M nt gabor_fwd bk16", 0, EPI_GABOR_FWD, 16, 0}, {"4M nt gabor_bwd bk16", 0, EPI_GABOR_BWD, 16, 0},
      {"4M tn wgrad", 1, 0, 0, 0},
      {"3M nt store", 2, EPI_STORE, 0, 0},          {"3M nt gabor_fwd", 2, EPI_GABOR_FWD, 0, 0},
      {"3M nt gabor_bwd", 2, EPI_GABOR_BWD, 0, 0},  {"3M tn wgrad", 3, 0, 0, 0},
      {"3M store noglobal", 2, EPI_STORE, 0, 1},    {"3M store nog+nolds", 2, EPI_STORE, 0, 3},
      {"3M store nobarrier", 2, EPI_STORE, 0, 4},   {"3M store mfma only", 2, EPI_STORE, 0, 7},
      {"3M fwd 3WG/CU", 2, EPI_GABOR_FWD, 0, 12 << 8}, {"3M fwd 2WG/CU", 2, EPI_GABOR_FWD, 0, 36 << 8},
      {"3M fwd 1WG/CU", 2, EPI_GABOR_FWD, 0, 100 << 8}, {"3M mfma-only 1WG/CU", 2, EPI_STORE, 0, (100 << 8) | 7},
      {"3M mfma-only 2WG/CU", 2, EPI_STORE, 0, (36 << 8) | 7},

  };
  std::vector<double> best(vars.size(), 1e30), sum(vars.size(), 0);
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t v = 0; v < vars.size(); ++v) {
      const Var& V = vars[v];
      GemmEpiParams ep;
      ep.bias = bias; ep.o0 = o0; ep.o1 = o1; ep.i0 = lin; ep.i1 = out; ep.ld0 = P; ep.ld1 = P;
      ep.omega = 20.f; ep.scale = 30.f; ep.kvalid = P / 2; ep.ablate = V.abl;
      CK(hipEventRecord(e0, 0));
      if (V.kind == 0) {
        gemm_tune_set("nt_bk", V.bk);
        CK(launch_gemm_nt(0, V.epi, A, P, Bt, P, N, P, P, ep));
      } else if (V.kind == 1) {
        CK(launch_gemm_tn(0, A, P, out, P, N, P, P, S, slab, bslab));
      } else if (V.kind == 2) {
        CK(launch_gemm3m_nt(0, V.epi, A, P, Bt, P, N, Kp, Kp, ep));
      } else {
        CK(launch_gemm3m_tn(0, A, P, out, P, N, Kp, Kp, S3, slab, bslab));
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
    }
  }
  printf("N=%lld P=%d  (%.1f algorithmic GFLOP per launch; fp32 MFMA peak 157.3 TF; 3M executes 0.75x)\n",
         (long long)N, P, flop / 1e9);
  for (size_t v = 0; v < vars.size(); ++v) {
    const double tf = flop / (sum[v] / rounds * 1e-3) / 1e12;
    const double mf = (vars[v].kind >= 2 ? 0.75 : 1.0) * tf;   // 3M kinds execute 0.75x the flops
    printf("  %-22s mean %7.3f ms  min %7.3f ms  -> %6.1f alg TF (%5.1f%%)   MFMA busy %5.1f%%\n",
           vars[v].name, sum[v] / rounds, best[v], tf, 100.0 * tf / 157.3, 100.0 * mf / 157.3);
  }
  return 0;
}
