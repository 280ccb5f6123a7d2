// gemm_tune.hip -- native A/B harness for the fp32-MFMA GEMM kernels (no Python,
// no torch): times kernel variants with HIP events, interleaved rounds in ONE
// process (cdna guide rule 24), random operands.  The kernels are compiled in
// with WIRE_ABLATE so single costs (global loads, LDS writes, barriers) can be
// switched off (results are then wrong; only the timing matters).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/gemm_tune.hip -o build/gemm_tune.o
//   hipcc --offload-arch=gfx950 build/gemm_tune.o build/csrc/wire_point.o -o build/gemm_tune
//   ./build/gemm_tune [N] [P] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define WIRE_ABLATE 1
#include "../wire_amd/csrc/wire_gemm.hip"     // 4-multiplication kernels
#include "../wire_amd/csrc/wire_gemm3m.hip"   // 3-multiplication kernels
#include "../wire_amd/csrc/wire_gemmx3.hip"   // split-bf16 kernels

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
__global__ void add_kernel(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += v;
}
__global__ void zero_some_kernel(float* p, size_t n, int pct) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2246822519u) ^ 0x9e3779b9u;
  x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
  if ((int)(x % 100u) < pct) p[i] = 0.f;
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
  // argv[5] = 1: constant operands (every element 0.75 / 0.0625: the m and l planes are zero, nothing toggles) --
  // same instruction stream, a fraction of the switching power: separates stalls from the power limit
  const bool constant = argc > 5 && atoi(argv[5]) == 1;
  float* A = dalloc((size_t)N * P, 1, constant ? 0.f : 1.0f);
  float* Bt = dalloc((size_t)P * P, 2, constant ? 0.f : 0.06f);
  // argv[6] = percentage of the activation elements set to exactly 0 (pseudo-random positions): do zero
  // operands cost the matrix pipe less power?  (WIRE activations at s0 = 30 are mostly < 1e-30.)
  const int zero_pct = argc > 6 ? atoi(argv[6]) : 0;
  if (zero_pct > 0) zero_some_kernel<<<(unsigned)(((size_t)N * P + 255) / 256), 256>>>(A, (size_t)N * P, zero_pct);
  if (constant) {
    add_kernel<<<(unsigned)(((size_t)N * P + 255) / 256), 256>>>(A, (size_t)N * P, 0.75f);
    add_kernel<<<(unsigned)(((size_t)P * P + 255) / 256), 256>>>(Bt, (size_t)P * P, 0.0625f);
  }
  float* bias = dalloc(P, 3, 0.06f);
  float* lin = dalloc((size_t)N * P, 4, 0.5f);
  float* out = dalloc((size_t)N * P, 5, 1.0f);
  float* o0 = dalloc((size_t)N * P, 6, 0.f);
  float* o1 = dalloc((size_t)N * P, 7, 0.f);
  const int S = gemm_tn_splits(N, P, P, 64);
  const int S3 = gemm3m_tn_splits(N, Kp, Kp, 64);
  float* slab = dalloc((size_t)64 * P * P, 8, 0.f);
  float* bslab = dalloc((size_t)64 * P, 9, 0.f);
  const int SX = gemmx3_tn_splits(N, P, P, 64);
  float* Bx3; CK(hipMalloc(&Bx3, (size_t)gemmx3_b_image_floats(P, P) * 4));
  CK(launch_x3_split_b(0, Bt, P, P, P, Bx3));
  CK(hipDeviceSynchronize());
  // ---- correctness of the split-bf16 kernels against the fp32-MFMA ones (same operands)
  {
    const int64_t Nc = N < 8192 ? N : 8192;
    GemmEpiParams ep; ep.o0 = o0; ep.ld0 = P; ep.ld1 = P;
    CK(launch_gemm_nt(0, EPI_STORE, A, P, Bt, P, Nc, P, P, ep));
    ep.o0 = o1;
    CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, Nc, P, P, ep));
    std::vector<float> c0((size_t)Nc * P), c1((size_t)Nc * P);
    CK(hipMemcpy(c0.data(), o0, c0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), o1, c1.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, md = 0;
    for (size_t i = 0; i < c0.size(); ++i) { mx = fmax(mx, fabs(c0[i])); md = fmax(md, fabs((double)c0[i] - c1[i])); }
    printf("check NT  x3 vs fp32-MFMA: max|diff| %.3e  max|C| %.3e  rel %.3e\n", md, mx, md / mx);
    float* slab2; CK(hipMalloc(&slab2, (size_t)64 * P * P * 4));
    float* bslab2; CK(hipMalloc(&bslab2, (size_t)64 * P * 4));
    CK(launch_gemm_tn(0, A, P, out, P, Nc - 5, P, P, 4, slab, bslab));
    gemmx3_tune_set("x3_tn_tall", argc > 4 ? atoi(argv[4]) : 0);
    CK(launch_gemmx3_tn(0, A, P, out, P, Nc - 5, P, P, 8, slab2, bslab2));   // ragged row count
    std::vector<float> s0((size_t)4 * P * P), s1((size_t)8 * P * P), b0(4 * P), b1(8 * P);
    CK(hipMemcpy(s0.data(), slab, s0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(s1.data(), slab2, s1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b0.data(), bslab, b0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b1.data(), bslab2, b1.size() * 4, hipMemcpyDeviceToHost));
    mx = 0; md = 0;
    for (size_t i = 0; i < (size_t)P * P; ++i) {
      double a = 0, b = 0;
      for (int q = 0; q < 4; ++q) a += s0[q * (size_t)P * P + i];
      for (int q = 0; q < 8; ++q) b += s1[q * (size_t)P * P + i];
      mx = fmax(mx, fabs(a)); md = fmax(md, fabs(a - b));
    }
    printf("check TN  x3 vs fp32-MFMA: max|diff| %.3e  max|C| %.3e  rel %.3e\n", md, mx, md / mx);
    mx = 0; md = 0;
    for (int i = 0; i < P; ++i) {
      double a = 0, b = 0;
      for (int q = 0; q < 4; ++q) a += b0[q * P + i];
      for (int q = 0; q < 8; ++q) b += b1[q * P + i];
      mx = fmax(mx, fabs(a)); md = fmax(md, fabs(a - b));
    }
    printf("check TN bias sums:        max|diff| %.3e  max %.3e\n", md, mx);
    CK(hipFree(slab2)); CK(hipFree(bslab2));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;    // algorithmic: 8 flop per complex MAC

  // kind: 0 = 4M NT, 1 = 4M TN, 2 = 3M NT, 3 = 3M TN
  struct Var { const char* name; int kind; int epi; int bk; int abl; };
  std::vector<Var> vars = {
      {"4M nt gabor_fwd bk16", 0, EPI_GABOR_FWD, 16, 0}, {"4M nt gabor_bwd bk16", 0, EPI_GABOR_BWD, 16, 0},
      {"4M tn wgrad", 1, 0, 0, 0},
      {"x3 nt store", 4, EPI_STORE, 0, 0},          {"x3 nt gabor_fwd", 4, EPI_GABOR_FWD, 0, 0},
      {"x3 nt gabor_bwd", 4, EPI_GABOR_BWD, 0, 0},  {"x3 tn wgrad", 5, 0, 0, 0},
      {"3M nt store", 2, EPI_STORE, 0, 0},          {"3M nt gabor_fwd", 2, EPI_GABOR_FWD, 0, 0},
      {"3M nt gabor_bwd", 2, EPI_GABOR_BWD, 0, 0},  {"3M tn wgrad", 3, 0, 0, 0},
      {"x3 tn tall wgrad", 5, 0, 1, 0},
      {"x3 tall store", 4, EPI_STORE, 1, 0},        {"x3 tall gabor_fwd", 4, EPI_GABOR_FWD, 1, 0},
      {"x3 tall gabor_bwd", 4, EPI_GABOR_BWD, 1, 0}, {"x3 tall mfma only", 4, EPI_STORE, 1, 7},
      {"x3 tall noglobal", 4, EPI_STORE, 1, 1},
      {"x3 store no A loads", 4, EPI_STORE, 0, 128}, {"x3 store no B loads", 4, EPI_STORE, 0, 256},
      {"x3 store nobarrier", 4, EPI_STORE, 0, 4},   {"x3 tall nobarrier", 4, EPI_STORE, 1, 4},
      {"x3 store noglobal", 4, EPI_STORE, 0, 1},    {"x3 store nosplit/lds", 4, EPI_STORE, 0, 2},
      {"x3 store nog+nolds", 4, EPI_STORE, 0, 3},   {"x3 store mfma only", 4, EPI_STORE, 0, 7},
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
      } else if (V.kind == 3) {
        CK(launch_gemm3m_tn(0, A, P, out, P, N, Kp, Kp, S3, slab, bslab));
      } else if (V.kind == 4) {
        gemmx3_tune_set("x3_tall", V.bk);
        CK(launch_gemmx3_nt(0, V.epi, A, P, Bx3, N, P, P, ep));
      } else {
        gemmx3_tune_set("x3_tn_tall", V.bk);
        CK(launch_gemmx3_tn(0, A, P, out, P, N, P, P, SX, slab, bslab));
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
    }
  }
  // in-kernel clock of the split-bf16 NT kernel (s_memtime / s_memrealtime around the main loop, median over
  // workgroups) in three states: whole kernel, without global loads, MFMA + LDS reads only
  {
    const int abl[3] = {64, 64 | 1, 64 | 7};
    const char* nm[3] = {"x3 store (full)", "x3 store noglobal", "x3 store mfma only"};
    const int nwg = (int)((N + 127) / 128 + 7) / 8 * 8 * (P / 128);
    std::vector<unsigned long long> stamps((size_t)nwg * 2);
    for (int v = 0; v < 3; ++v) {
      GemmEpiParams ep; ep.o0 = o0; ep.ld0 = P; ep.ld1 = P; ep.i0 = lin; ep.ablate = abl[v];
      gemmx3_tune_set("x3_tall", 0);
      CK(hipMemset(lin, 0, (size_t)nwg * 16));
      for (int r = 0; r < 20; ++r) CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, N, P, P, ep));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(stamps.data(), lin, stamps.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> ghz;
      for (int w = 0; w < nwg; ++w) if (stamps[2 * w + 1] > 0) ghz.push_back((double)stamps[2 * w] / (double)stamps[2 * w + 1] * 0.1);
      std::sort(ghz.begin(), ghz.end());
      if (!ghz.empty()) printf("clock %-20s median %.3f GHz  (p10 %.3f, p90 %.3f, %zu workgroups)\n", nm[v], ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], ghz.size());
    }
  }
  printf("N=%lld P=%d  (%.1f algorithmic GFLOP per launch; fp32 MFMA peak 157.3 TF; 3M executes 0.75x)\n",
         (long long)N, P, flop / 1e9);
  for (size_t v = 0; v < vars.size(); ++v) {
    const double tf = flop / (sum[v] / rounds * 1e-3) / 1e12;
    const double mf = ((vars[v].kind == 2 || vars[v].kind == 3) ? 0.75 : 1.0) * tf;   // 3M kinds execute 0.75x the flops
    printf("  %-22s mean %7.3f ms  min %7.3f ms  -> %6.1f alg TF (%5.1f%%)   MFMA busy %5.1f%%\n",
           vars[v].name, sum[v] / rounds, best[v], tf, 100.0 * tf / 157.3, 100.0 * mf / 157.3);
  }
  return 0;
}
