// mfma_issue_probe.hip -- which instruction kinds take time away from v_mfma_f32_32x32x2_f32?
//  mode SAME : one wave per SIMD runs NITER x 4 x {1 MFMA + K fillers}
//  mode CROSS: two waves per SIMD; waves 0-3 run MFMAs only, waves 4-7 run ONLY fillers
//              (4*K*NITER of them); the kernel time shows whether the fillers of another wave
//              steal matrix-pipe time or just run alongside.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { F_FMA = 0, F_SALU = 1, F_DSREAD = 2, F_NOP = 3, F_MOV = 4 };

template <int KIND>
__device__ __forceinline__ void filler(float& v, float y, float x, int& sc, f32x4& q, unsigned la) {
  if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(y), "v"(x));
  if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(y));
  if (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
  if (KIND == F_NOP) asm volatile("s_nop 0");
  if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(la));
}

template <int KIND, int K, bool CROSS>
__global__ __launch_bounds__(512) void probe(float* out, int niter) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8];
  f32x4 q[4];
  int sc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = f32x4{x, y, x, y};
  const unsigned la = ((threadIdx.x & 255) * 16) & 0x3fff;
  const bool mfma_role = !CROSS || (threadIdx.x < 256);
  const bool fill_role = !CROSS || (threadIdx.x >= 256);
  if (CROSS && !mfma_role) {
    for (int it = 0; it < niter; ++it) {
#pragma unroll
      for (int i = 0; i < 4 * K; ++i) filler<KIND>(v[i & 7], y, x, sc, q[i & 3], la);
    }
  } else {
    for (int it = 0; it < niter; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        if (m == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        if (m == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        if (m == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        if (fill_role) {
#pragma unroll
          for (int i = 0; i < K; ++i) filler<KIND>(v[(i + 2 * m) & 7], y, x, sc, q[i & 3], la);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  float s = sc;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K, bool CROSS>
double run(float* out, int niter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = CROSS ? 512 : 256;
  probe<KIND, K, CROSS><<<256, threads>>>(out, niter);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND, K, CROSS><<<256, threads>>>(out, niter);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / (4.0 * niter);
}
template <int KIND>
void sweep(const char* name, float* out, int niter) {
  printf("%-14s same wave : K=0 %6.1f  K=4 %6.1f  K=8 %6.1f  K=16 %6.1f | other wave: K=4 %6.1f  K=8 %6.1f  K=16 %6.1f  (cycles per MFMA slot)\n",
         name, run<KIND, 0, false>(out, niter), run<KIND, 4, false>(out, niter), run<KIND, 8, false>(out, niter),
         run<KIND, 16, false>(out, niter), run<KIND, 4, true>(out, niter), run<KIND, 8, true>(out, niter),
         run<KIND, 16, true>(out, niter));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  const int niter = 10000;
  sweep<F_FMA>("v_fma_f32", out, niter);
  sweep<F_MOV>("v_mov_b32", out, niter);
  sweep<F_SALU>("s_add_u32", out, niter);
  sweep<F_NOP>("s_nop 0", out, niter);
  sweep<F_DSREAD>("ds_read_b128", out, niter);
  return 0;
}
