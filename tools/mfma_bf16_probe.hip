// mfma_bf16_probe.hip -- what does an instruction cost beside a v_mfma_f32_32x32x16_bf16 stream on gfx950?
// W waves per SIMD: NITER x 4 x { 1 MFMA + K filler instructions of one kind }.  Reports SIMD cycles per
// MFMA slot per wave-set (wall x 2.4 GHz nominal / MFMAs per SIMD): 32 = matrix pipe saturated.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { F_FMA = 0, F_DSREAD = 1, F_DSWRITE = 2, F_GLOAD_HIT = 3, F_GLOAD_L2 = 4, F_GLOAD_B32 = 5, F_CVT = 6 };

template <int KIND, int K>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int niter, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(x + i); fb[i] = (__bf16)(y + i); }
  float v[8];
  f32x4 q[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = f32x4{x, y, x, y};
  const float* gp = in + threadIdx.x * 4;                              // same 4 KB for every workgroup: L1 hits
  const float* gp2 = in + ((size_t)blockIdx.x * 65536 + threadIdx.x * 4) % (12u << 20);   // floats: a 48 MB window, streams through L2
  const unsigned la = (threadIdx.x * 16) & 0x7fff;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a0, 0, 0, 0);
      if (m == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a1, 0, 0, 0);
      if (m == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a2, 0, 0, 0);
      if (m == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, a3, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const int j = (i + 2 * m) & 7;
        // asynchronous loads land in FIXED registers that are declared clobbered and never used otherwise: an
        // "=v" output would let the compiler recycle the register while the load is still in flight
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(y), "v"(x));
        if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(y));
        if (KIND == F_DSREAD) {
          if ((j & 1) == 0) asm volatile("ds_read_b128 v[88:91], %0" ::"v"(la) : "v88", "v89", "v90", "v91");
          else asm volatile("ds_read_b128 v[92:95], %0" ::"v"(la) : "v92", "v93", "v94", "v95");
        }
        if (KIND == F_DSWRITE) asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(q[j & 3]));
        if (KIND == F_GLOAD_HIT) {
          if ((j & 1) == 0) asm volatile("global_load_dwordx4 v[88:91], %0, off" ::"v"(gp) : "v88", "v89", "v90", "v91");
          else asm volatile("global_load_dwordx4 v[92:95], %0, off" ::"v"(gp) : "v92", "v93", "v94", "v95");
        }
        if (KIND == F_GLOAD_L2) {
          const float* a = gp2 + (size_t)((it * 4 + m) & 15) * 1024;
          if ((j & 1) == 0) asm volatile("global_load_dwordx4 v[88:91], %0, off" ::"v"(a) : "v88", "v89", "v90", "v91");
          else asm volatile("global_load_dwordx4 v[92:95], %0, off" ::"v"(a) : "v92", "v93", "v94", "v95");
        }
        if (KIND == F_GLOAD_B32) asm volatile("global_load_dword v88, %0, off" ::"v"(gp) : "v88");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_waves = 1;
template <int KIND, int K>
void run(float* out, const float* in, int niter, unsigned long long* clk, double& cyc, double& ghz) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * g_waves;
  probe<KIND, K><<<blocks, 256>>>(out, in, niter, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND, K><<<blocks, 256>>>(out, in, niter, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  ghz = (double)c[0] / (double)c[1] * 0.1;
  // in-kernel cycles per MFMA issued on a SIMD (all waves of the SIMD together)
  cyc = ms * 1e-3 * ghz * 1e9 / (4.0 * niter * g_waves);
}

template <int KIND>
void sweep(const char* name, float* out, const float* in, int niter, unsigned long long* clk) {
  double c[4], g[4];
  run<KIND, 0>(out, in, niter, clk, c[0], g[0]); run<KIND, 1>(out, in, niter, clk, c[1], g[1]);
  run<KIND, 2>(out, in, niter, clk, c[2], g[2]); run<KIND, 4>(out, in, niter, clk, c[3], g[3]);
  printf("  %-26s cycles/MFMA: K=0 %5.1f  K=1 %5.1f  K=2 %5.1f  K=4 %5.1f   (clock %.2f / %.2f / %.2f / %.2f GHz)\n", name, c[0],
         c[1], c[2], c[3], g[0], g[1], g[2], g[3]);
}

int main() {
  float *out, *in; unsigned long long* clk;
  hipMalloc(&out, (size_t)1024 * 256 * sizeof(float));
  hipMalloc(&in, (size_t)80 << 20);      // 80 MB >= 48 MB window + 16 x 4 KB strides
  hipMemset(in, 0, (size_t)80 << 20);
  hipMalloc(&clk, 1024 * 16);
  const int niter = 20000;
  for (g_waves = 1; g_waves <= 2; ++g_waves) {
    printf("%d wave(s) per SIMD\n", g_waves);
    sweep<F_FMA>("v_fma_f32", out, in, niter, clk);
    sweep<F_CVT>("v_cvt_pk_bf16_f32", out, in, niter, clk);
    sweep<F_DSREAD>("ds_read_b128", out, in, niter, clk);
    sweep<F_DSWRITE>("ds_write_b128", out, in, niter, clk);
    sweep<F_GLOAD_HIT>("global_load_dwordx4 (L1)", out, in, niter, clk);
    sweep<F_GLOAD_B32>("global_load_dword (L1)", out, in, niter, clk);
  }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
  return 0;
}
