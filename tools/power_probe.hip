// power_probe.hip -- do matrix-core work and HBM traffic draw on one budget on this chip?
// Two kernels that share nothing but the chip: an MFMA-only loop (v_mfma_f32_16x16x32_f16 on finite pseudo-random operands,
// one 4-wave workgroup per CU, no memory traffic) and a plain float4 copy (HBM read + write).  Each alone, then both at once on
// two streams.  If the pair finishes in max(t_mfma, t_copy) the two are independent resources and a GEMM whose time is
// "MFMA time + bytes / bandwidth" is leaving overlap on the table; if it takes about the sum, they share a budget (power).
//   hipcc --offload-arch=gfx950 -O3 -o build/power_probe tools/power_probe.hip && ./build/power_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 rnd8(unsigned x, int zero) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float f = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f);
    v[i] = zero ? (_Float16)0.f : (_Float16)f;
    x += 0x9e3779b9u;
  }
  return v;
}

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned seed, int zero) {
  const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 977u + seed;
  f16x8 a[4], b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = rnd8(id + 17u * i, zero);
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = rnd8(id * 3u + 131u * j, zero);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 3; ++r)          // three products per block, as the 2 x fp16 split issues them
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[(i + r) & 3], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ src, float* __restrict__ sink, size_t n4) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const f32x4 v = src[i]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 3;      // GiB read (and written by the copy)
  const size_t n4 = gb * (1ull << 30) / 16;
  f32x4 *src, *dst; float* out;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&out, 4096 * 256 * 4));
  CK(hipMemset(src, 0x3c, n4 * 16)); CK(hipMemset(dst, 0, n4 * 16));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  auto wall = [&](auto&& fn) {
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    fn();
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  const double mfma_flop_per_wg = 4.0 * iters * 96 * (2.0 * 16 * 16 * 32);
  for (int zero = 0; zero < 2; ++zero)
    for (int wgs : {256, 512, 128}) {
      auto k_m = [&] { hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, s1, out, iters, 12345u, zero); };
      auto k_c = [&] { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, s2, src, dst, n4); };
      auto k_r = [&] { hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, s2, src, out, n4); };
      double tm = 1e30, tc = 1e30, tr = 1e30, tmc = 1e30, tmr = 1e30;
      for (int rep = 0; rep < 4; ++rep) {
        const double a = wall(k_m), b = wall(k_c), c = wall(k_r);
        const double d = wall([&] { k_m(); k_c(); }), e = wall([&] { k_m(); k_r(); });
        if (rep) { tm = a < tm ? a : tm; tc = b < tc ? b : tc; tr = c < tr ? c : tr; tmc = d < tmc ? d : tmc; tmr = e < tmr ? e : tmr; }
      }
      printf("%s operands, %3d MFMA workgroups (4 waves each): mfma alone %.3f ms (%.0f TFLOP/s f16) | copy %zu GiB alone %.3f ms (%.2f TB/s r+w) | "
             "read alone %.3f ms (%.2f TB/s)\n", zero ? "ZERO  " : "random", wgs, tm, wgs * mfma_flop_per_wg / tm / 1e9, gb, tc,
             2.0 * n4 * 16 / tc / 1e9, tr, n4 * 16.0 / tr / 1e9);
      printf("      together: mfma + copy %.3f ms (max %.3f, sum %.3f) | mfma + read %.3f ms (max %.3f, sum %.3f)\n", tmc,
             tm > tc ? tm : tc, tm + tc, tmr, tm > tr ? tm : tr, tm + tr);
    }
  return 0;
}
