// bf16x3_numerics.hip -- is a 6-product 3xbf16 split GEMM on v_mfma_f32_32x32x16_bf16 as accurate as
// the exact-f32 v_mfma_f32_32x32x2_f32 chain?  C[M][N] = A[M][K] * B[N][K]^T, one wave per 32x32 tile,
// fp64 host reference.  Build: hipcc --offload-arch=gfx950 -O3 tools/bf16x3_numerics.hip -o /tmp/bf16x3_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned h) { return __uint_as_float(h << 16); }
// x = h + m + l (+ <= 2^-27 |x|)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = bf16_rne(x);
  const float r1 = x - bf16_f(h);
  m = bf16_rne(r1);
  const float r2 = r1 - bf16_f(m);
  l = bf16_rne(r2);
}
struct Frag3 { bf16x8 h, m, l; };
__device__ __forceinline__ Frag3 load_split(const float* p) {
  unsigned hh[8], mm[8], ll[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(p[j], hh[j], mm[j], ll[j]);
  uint4 H = {hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16)};
  uint4 Mv = {mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16), mm[4] | (mm[5] << 16), mm[6] | (mm[7] << 16)};
  uint4 L = {ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16), ll[4] | (ll[5] << 16), ll[6] | (ll[7] << 16)};
  Frag3 f;
  f.h = __builtin_bit_cast(bf16x8, H); f.m = __builtin_bit_cast(bf16x8, Mv); f.l = __builtin_bit_cast(bf16x8, L);
  return f;
}
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// variant 0: exact f32 MFMA; 1: one accumulator, small terms first inside each 16-chunk; 2: hh in one
// accumulator, the five cross terms in another; 3: passes over K per term, small first; 4: nine products;
// 5: one accumulator, big term first
template <int V>
__global__ __launch_bounds__(64) void gemm(float* C, const float* A, const float* B, int M, int N, int K) {
  const int tm = blockIdx.x * 32, tn = blockIdx.y * 32, lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {}, acc2 = {};
  if (V == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(tm + r) * K + k + h], B[(size_t)(tn + r) * K + k + h], acc, 0, 0, 0);
  } else if (V == 3) {
    for (int t = 0; t < 6; ++t)
      for (int k = 0; k < K; k += 16) {
        Frag3 a = load_split(A + (size_t)(tm + r) * K + k + 8 * h), b = load_split(B + (size_t)(tn + r) * K + k + 8 * h);
        if (t == 0) MF(a.h, b.l, acc);
        if (t == 1) MF(a.l, b.h, acc);
        if (t == 2) MF(a.m, b.m, acc);
        if (t == 3) MF(a.h, b.m, acc);
        if (t == 4) MF(a.m, b.h, acc);
        if (t == 5) MF(a.h, b.h, acc);
      }
  } else {
    for (int k = 0; k < K; k += 16) {
      Frag3 a = load_split(A + (size_t)(tm + r) * K + k + 8 * h), b = load_split(B + (size_t)(tn + r) * K + k + 8 * h);
      if (V == 1) { MF(a.h, b.l, acc); MF(a.l, b.h, acc); MF(a.m, b.m, acc); MF(a.h, b.m, acc); MF(a.m, b.h, acc); MF(a.h, b.h, acc); }
      if (V == 5) { MF(a.h, b.h, acc); MF(a.h, b.m, acc); MF(a.m, b.h, acc); MF(a.m, b.m, acc); MF(a.h, b.l, acc); MF(a.l, b.h, acc); }
      if (V == 2) { MF(a.h, b.l, acc2); MF(a.l, b.h, acc2); MF(a.m, b.m, acc2); MF(a.h, b.m, acc2); MF(a.m, b.h, acc2); MF(a.h, b.h, acc); }
      if (V == 4) { MF(a.l, b.l, acc2); MF(a.m, b.l, acc2); MF(a.l, b.m, acc2); MF(a.h, b.l, acc2); MF(a.l, b.h, acc2); MF(a.m, b.m, acc2); MF(a.h, b.m, acc2); MF(a.m, b.h, acc2); MF(a.h, b.h, acc); }
    }
    if (V == 2 || V == 4)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[(size_t)(tm + row) * N + tn + r] = acc[i];
  }
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

int main() {
  const int M = 256, N = 256, K = 512;
  const char* names[6] = {"f32 mfma 32x32x2", "3xbf16 6p small-first", "3xbf16 6p two-acc", "3xbf16 6p term-passes", "3xbf16 9p two-acc", "3xbf16 6p big-first"};
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: signed gaussian x wide log-uniform magnitudes (Gabor-like activations) against U(-1/16,1/16) weights
    // mode 1: all positive (exposes a truncation bias in the accumulate); mode 2: plain gaussian
    std::vector<float> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N);
    std::vector<double> R((size_t)M * N), S((size_t)M * N);
    srand(1234 + mode);
    for (auto& a : A) a = mode == 0 ? (float)(nrand() * std::exp(-12.0 * urand())) : mode == 1 ? (float)urand() : (float)nrand();
    for (auto& b : B) b = mode == 1 ? (float)(urand() / 16) : (float)((2 * urand() - 1) / 16);
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        double s = 0, sa = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)j * K + k]; s += p; sa += std::fabs(p); }
        R[(size_t)i * N + j] = s; S[(size_t)i * N + j] = sa;
      }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("mode %d\n", mode);
    for (int v = 0; v < 6; ++v) {
      dim3 g(M / 32, N / 32);
      if (v == 0) gemm<0><<<g, 64>>>(dC, dA, dB, M, N, K);
      if (v == 1) gemm<1><<<g, 64>>>(dC, dA, dB, M, N, K);
      if (v == 2) gemm<2><<<g, 64>>>(dC, dA, dB, M, N, K);
      if (v == 3) gemm<3><<<g, 64>>>(dC, dA, dB, M, N, K);
      if (v == 4) gemm<4><<<g, 64>>>(dC, dA, dB, M, N, K);
      if (v == 5) gemm<5><<<g, 64>>>(dC, dA, dB, M, N, K);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double maxrel = 0, sum2 = 0, bias = 0;
      for (size_t i = 0; i < C.size(); ++i) {
        const double e = ((double)C[i] - R[i]) / S[i];   // error in units of sum|a b|
        maxrel = std::fmax(maxrel, std::fabs(e)); sum2 += e * e; bias += e;
      }
      printf("  %-24s err/sum|ab|: max %.3e  rms %.3e  mean %+.3e\n", names[v], maxrel, std::sqrt(sum2 / C.size()), bias / C.size());
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
