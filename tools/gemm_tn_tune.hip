// gemm_tn_tune.hip -- A/B of the split-bf16 weight-gradient (TN) kernels in one process: the 128 x 128 / 32x32x16
// kernel against the 256 x 256 / 16x16x32 one ("x3_tn16"), random operands, HIP events, interleaved rounds; plus
// agreement of their slab sums (fp32 round-off; checked against an fp64 host product on a small case).
//   bash tools/build_tn_tune.sh && ./build/gemm_tn_tune [N] [P] [rounds] [timing-only]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../wire_amd/csrc/wire_gemm.h"

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

// sum of the slabs (and bias slabs) of one launch, in double on the host
static void run_sum(int mode, const float* G, const float* Z, int64_t n, int P, std::vector<double>& W,
                    std::vector<double>& B, int* S_out) {
  gemmx3_tune_set("x3_tn16", mode);
  const int S = gemmx3_tn_splits(n, P, P, 256);
  float* slab; CK(hipMalloc(&slab, (size_t)S * P * P * 4));
  float* bslab; CK(hipMalloc(&bslab, (size_t)S * P * 4));
  CK(hipMemset(slab, 0xff, (size_t)S * P * P * 4));
  CK(hipMemset(bslab, 0xff, (size_t)S * P * 4));
  CK(launch_gemmx3_tn(0, G, P, Z, P, n, P, P, S, slab, bslab));
  std::vector<float> h((size_t)S * P * P), hb((size_t)S * P);
  CK(hipMemcpy(h.data(), slab, h.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), bslab, hb.size() * 4, hipMemcpyDeviceToHost));
  W.assign((size_t)P * P, 0.0); B.assign(P, 0.0);
  for (int s = 0; s < S; ++s) {
    for (size_t i = 0; i < (size_t)P * P; ++i) W[i] += h[(size_t)s * P * P + i];
    for (int i = 0; i < P; ++i) B[i] += hb[(size_t)s * P + i];
  }
  CK(hipFree(slab)); CK(hipFree(bslab));
  *S_out = S;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 262144;
  const int P = argc > 2 ? atoi(argv[2]) : 512;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  float* G = dalloc((size_t)N * P, 1, 1.0f);
  float* Z = dalloc((size_t)N * P, 5, 1.0f);
  CK(hipDeviceSynchronize());
  // ---- agreement: full and ragged row counts; the small one also against fp64
  const bool timing_only = argc > 4;               // 4th argument: skip the agreement checks
  for (int64_t n : {(int64_t)1000 + 13, (int64_t)40000 + 5, N}) {
    if (n > N || timing_only) continue;
    std::vector<double> W0, B0, W1, B1; int S0, S1;
    run_sum(0, G, Z, n, P, W0, B0, &S0);
    run_sum(1, G, Z, n, P, W1, B1, &S1);
    double md = 0, mx = 0, mb = 0;
    for (size_t i = 0; i < W0.size(); ++i) { md = fmax(md, fabs(W0[i] - W1[i])); mx = fmax(mx, fabs(W0[i])); }
    for (size_t i = 0; i < B0.size(); ++i) mb = fmax(mb, fabs(B0[i] - B1[i]));
    printf("rows %lld: tn16 (%d splits) vs tn (%d splits): max |dW| %.3e of max |W| %.3e (rel %.2e), max |db| %.3e\n",
           (long long)n, S1, S0, md, mx, md / mx, mb);
    if (n < 2000) {
      std::vector<float> hg((size_t)n * P), hz((size_t)n * P);
      CK(hipMemcpy(hg.data(), G, hg.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hz.data(), Z, hz.size() * 4, hipMemcpyDeviceToHost));
      double e0 = 0, e1 = 0, eb = 0;
      for (int m = 0; m < P; m += 3)
        for (int c = 0; c < P; c += 5) {
          double acc = 0;
          for (int64_t r = 0; r < n; ++r) acc += (double)hg[r * P + m] * hz[r * P + c];
          e0 = fmax(e0, fabs(acc - W0[(size_t)m * P + c]));
          e1 = fmax(e1, fabs(acc - W1[(size_t)m * P + c]));
        }
      for (int m = 0; m < P; ++m) {
        double acc = 0;
        for (int64_t r = 0; r < n; ++r) acc += hg[r * P + m];
        eb = fmax(eb, fabs(acc - B1[m]));
      }
      printf("   vs fp64 host product: tn %.3e  tn16 %.3e  (bias tn16 %.3e)\n", e0, e1, eb);
    }
  }
  // ---- timing
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;
  double tsum[2] = {0, 0}; int Ss[2] = {0, 0};
  float* slab; CK(hipMalloc(&slab, (size_t)256 * P * P * 4));
  float* bslab; CK(hipMalloc(&bslab, (size_t)256 * P * 4));
  for (int r = 0; r < rounds + 1; ++r)
    for (int mode = 0; mode < 2; ++mode) {
      gemmx3_tune_set("x3_tn16", mode);
      const int S = gemmx3_tn_splits(N, P, P, 256);
      Ss[mode] = S;
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) CK(launch_gemmx3_tn(0, G, P, Z, P, N, P, P, S, slab, bslab));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) tsum[mode] += ms / 4;
    }
  const char* names[2] = {"tn   128x128 32x32x16", "tn16 256x256 16x16x32"};
  for (int mode = 0; mode < 2; ++mode)
    printf("  %-28s mean %7.3f ms  (%d row splits)  -> %6.1f alg TF  frac %.3f\n", names[mode], tsum[mode] / rounds, Ss[mode],
           flop / (tsum[mode] / rounds * 1e-3) / 1e12, flop / (tsum[mode] / rounds * 1e-3) / 1e12 / 416.7);
  return 0;
}
