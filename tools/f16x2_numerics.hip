// f16x2_numerics.hip -- is a 3-product 2 x fp16 split GEMM (x = h + l, products hh + hl + lh, operands pre-scaled by a
// power of two so that the tensor maximum sits at 2^14) on v_mfma_f32_32x32x16_f16 as accurate as the exact-f32
// v_mfma_f32_32x32x2_f32 chain and as the 6-product 3 x bf16 split?  Also: does the f16 MFMA honour fp16 subnormals?
// C[M][N] = A[M][K] * B[N][K]^T, one wave per 32x32 tile, fp64 host reference.
// Build: hipcc --offload-arch=gfx950 -O3 tools/f16x2_numerics.hip -o build/f16x2_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned h) { return __uint_as_float(h << 16); }
struct Frag3 { bf16x8 h, m, l; };
__device__ __forceinline__ Frag3 load_split3(const float* p) {
  unsigned hh[8], mm[8], ll[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hh[j] = bf16_rne(p[j]);
    const float r1 = p[j] - bf16_f(hh[j]);
    mm[j] = bf16_rne(r1);
    ll[j] = bf16_rne(r1 - bf16_f(mm[j]));
  }
  uint4 H = {hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16)};
  uint4 Mv = {mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16), mm[4] | (mm[5] << 16), mm[6] | (mm[7] << 16)};
  uint4 L = {ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16), ll[4] | (ll[5] << 16), ll[6] | (ll[7] << 16)};
  Frag3 f;
  f.h = __builtin_bit_cast(bf16x8, H); f.m = __builtin_bit_cast(bf16x8, Mv); f.l = __builtin_bit_cast(bf16x8, L);
  return f;
}
struct Frag2 { f16x8 h, l; };
// x * s = h + l (+ <= 2^-24 |x s|); s a power of two
__device__ __forceinline__ Frag2 load_split2(const float* p, float s) {
  Frag2 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)(p[j] * s);
    const float r = __builtin_fmaf(p[j], s, -(float)h);
    f.h[j] = h;
    f.l[j] = (_Float16)r;
  }
  return f;
}
#define MB(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MH(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// variant 0: exact f32 MFMA; 1: 3 x bf16, 6 products, small terms first; 2: 2 x f16, 3 products, small first, scaled;
// 3: the same with the two small products in their own accumulator; 4: 2 x f16 unscaled (sa = sb = 1); 5: 2 x f16 scaled, 4 products (+ ll)
template <int V>
__global__ __launch_bounds__(64) void gemm(float* C, const float* A, const float* B, int M, int N, int K, float sa, float sb) {
  const int tm = blockIdx.x * 32, tn = blockIdx.y * 32, lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {}, acc2 = {};
  if (V == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(tm + r) * K + k + h], B[(size_t)(tn + r) * K + k + h], acc, 0, 0, 0);
  } else if (V == 1) {
    for (int k = 0; k < K; k += 16) {
      Frag3 a = load_split3(A + (size_t)(tm + r) * K + k + 8 * h), b = load_split3(B + (size_t)(tn + r) * K + k + 8 * h);
      MB(a.h, b.l, acc); MB(a.l, b.h, acc); MB(a.m, b.m, acc); MB(a.h, b.m, acc); MB(a.m, b.h, acc); MB(a.h, b.h, acc);
    }
  } else {
    const float s1 = V == 4 ? 1.f : sa, s2 = V == 4 ? 1.f : sb;
    for (int k = 0; k < K; k += 16) {
      Frag2 a = load_split2(A + (size_t)(tm + r) * K + k + 8 * h, s1), b = load_split2(B + (size_t)(tn + r) * K + k + 8 * h, s2);
      if (V == 3) { MH(a.h, b.l, acc2); MH(a.l, b.h, acc2); MH(a.h, b.h, acc); }
      else if (V == 5) { MH(a.l, b.l, acc); MH(a.h, b.l, acc); MH(a.l, b.h, acc); MH(a.h, b.h, acc); }
      else { MH(a.h, b.l, acc); MH(a.l, b.h, acc); MH(a.h, b.h, acc); }
    }
    const float inv = 1.f / (s1 * s2);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (acc[i] + acc2[i]) * inv;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[(size_t)(tm + row) * N + tn + r] = acc[i];
  }
}

// subnormal probe: a = 2^-20 (an fp16 subnormal), b = 1: sum over 16 k = 2^-16 when subnormal inputs are honoured, 0 when flushed
__global__ __launch_bounds__(64) void denorm_probe(float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1.f; }
  f32x16 acc = {};
  MH(a, b, acc);
  if (threadIdx.x == 0) out[0] = acc[0];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc4 = {};
  acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4, 0, 0, 0);
  if (threadIdx.x == 0) out[1] = acc4[0];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }
static float pow2_scale(const std::vector<float>& v) {   // max |v| * s in [2^14, 2^15)
  float m = 0;
  for (float x : v) m = std::fmax(m, std::fabs(x));
  if (!(m > 0)) return 1.f;
  int e;
  std::frexp(m, &e);   // m = f 2^e, f in [0.5, 1)
  return std::ldexp(1.f, 15 - e);
}

int main() {
  {
    float* d; hipMalloc(&d, 8);
    denorm_probe<<<1, 64>>>(d);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("subnormal probe: 16 x 2^-20 x 1 -> 32x32x16_f16 %.6e, 16x16x32_f16 (32 k) %.6e   (2^-16 = %.6e)\n", h[0], h[1], std::ldexp(1.0, -16));
    hipFree(d);
  }
  const int M = 256, N = 256, K = 512;
  const char* names[6] = {"f32 mfma 32x32x2", "3xbf16 6p small-first", "2xf16 3p scaled", "2xf16 3p scaled two-acc", "2xf16 3p UNscaled", "2xf16 4p scaled"};
  for (int mode = 0; mode < 5; ++mode) {
    // mode 0: signed gaussian x wide log-uniform magnitudes (Gabor-like activations) against U(-1/16,1/16) weights
    // mode 1: all positive; mode 2: plain gaussian; mode 3: gradient-like: 1e-7 x gaussian x log-uniform over 12 e-folds
    // mode 4: mode 0 with activations of maximum 1e-3 (a layer whose outputs are all small)
    std::vector<float> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N);
    std::vector<double> R((size_t)M * N), S((size_t)M * N);
    srand(1234 + mode);
    for (auto& a : A)
      a = mode == 0 ? (float)(nrand() * std::exp(-12.0 * urand())) : mode == 1 ? (float)urand() : mode == 2 ? (float)nrand()
          : mode == 3 ? (float)(1e-7 * nrand() * std::exp(-12.0 * urand())) : (float)(3e-4 * nrand() * std::exp(-12.0 * urand()));
    for (auto& b : B) b = mode == 1 ? (float)(urand() / 16) : (float)((2 * urand() - 1) / 16);
    double rmax = 0;
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        double s = 0, sa = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)j * K + k]; s += p; sa += std::fabs(p); }
        R[(size_t)i * N + j] = s; S[(size_t)i * N + j] = sa; rmax = std::fmax(rmax, std::fabs(s));
      }
    const float sa = pow2_scale(A), sb = pow2_scale(B);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("mode %d  (scales 2^%d, 2^%d)\n", mode, (int)std::log2(sa), (int)std::log2(sb));
    for (int v = 0; v < 6; ++v) {
      dim3 g(M / 32, N / 32);
      if (v == 0) gemm<0><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      if (v == 1) gemm<1><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      if (v == 2) gemm<2><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      if (v == 3) gemm<3><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      if (v == 4) gemm<4><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      if (v == 5) gemm<5><<<g, 64>>>(dC, dA, dB, M, N, K, sa, sb);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double maxrel = 0, sum2 = 0, bias = 0, maxabs = 0;
      for (size_t i = 0; i < C.size(); ++i) {
        const double e = ((double)C[i] - R[i]) / S[i];   // error in units of sum|a b|
        maxrel = std::fmax(maxrel, std::fabs(e)); sum2 += e * e; bias += e;
        maxabs = std::fmax(maxabs, std::fabs((double)C[i] - R[i]));
      }
      printf("  %-24s err/sum|ab|: max %.3e  rms %.3e  mean %+.3e   max|err|/max|C| %.3e\n", names[v], maxrel, std::sqrt(sum2 / C.size()),
             bias / C.size(), maxabs / rmax);
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
