// mfma_valu_probe.hip -- what shares the SIMD with v_mfma_f32_32x32x2_f32 on gfx950?
// One wave per SIMD: NITER x 4 x { 1 MFMA + K filler instructions of one kind }.
// Reports cycles per MFMA slot (2.4 GHz nominal); the slope in K is the filler's cost.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_FMA = 0, F_EXP = 1, F_SIN = 2, F_PKADD = 3, F_DSREAD = 4, F_SALU = 5, F_GLOAD = 6, F_MOV = 7, F_GSTORE = 8, F_DSWRITE = 9, F_DSREAD32 = 10 };

template <int KIND, int K>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int niter) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  float v[8];
  f32x2 p[4];
  f32x4 q[4];
  int sc2 = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x + i;
#pragma unroll
  for (int i = 0; i < 4; ++i) { p[i] = f32x2{x, y}; q[i] = f32x4{x, y, x, y}; }
  const float* gp = in + threadIdx.x * 4;
  float* op = out + 65536 + (blockIdx.x * 256 + threadIdx.x) * 4;
  const unsigned la = (threadIdx.x * 16) & 0x3fff;
  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      if (m == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      if (m == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      if (m == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const int j = (i + 2 * m) & 7;
        if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(y), "v"(x));
        if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if (KIND == F_SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(v[j]));
        if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(y));
        if (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3]));
        if (KIND == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(la));
        if (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc2));
        if (KIND == F_DSWRITE) asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(q[j & 3]));
        if (KIND == F_DSREAD32) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(la));
        if (KIND == F_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[j & 3]) : "v"(gp));
        if (KIND == F_GSTORE) asm volatile("global_store_dword %0, %1, off" ::"v"(op), "v"(v[j]));
      }
      // no wait here: the fillers stay in flight (counters saturate only for large K)
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  float s = sc2;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + q[i][0] + q[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K>
double run(float* out, const float* in, int niter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND, K><<<256, 256>>>(out, in, niter);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND, K><<<256, 256>>>(out, in, niter);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / (4.0 * niter);
}

template <int KIND>
void sweep(const char* name, float* out, const float* in, int niter) {
  const double c0 = run<KIND, 0>(out, in, niter), c2 = run<KIND, 2>(out, in, niter), c4 = run<KIND, 4>(out, in, niter),
               c8 = run<KIND, 8>(out, in, niter);
  printf("%-22s cycles/MFMA slot: K=0 %6.1f  K=2 %6.1f  K=4 %6.1f  K=8 %6.1f   -> ~%5.2f cycles per filler\n", name, c0,
         c2, c4, c8, (c8 - c0) / 8.0);
}

int main() {
  float *out, *in;
  hipMalloc(&out, (65536 + 256 * 256 * 4) * sizeof(float));
  hipMalloc(&in, 65536 * sizeof(float));
  hipMemset(in, 0, 65536 * sizeof(float));
  const int niter = 10000;
  sweep<F_FMA>("v_fma_f32", out, in, niter);
  sweep<F_MOV>("v_mov_b32", out, in, niter);
  sweep<F_EXP>("v_exp_f32", out, in, niter);
  sweep<F_SIN>("v_sin_f32", out, in, niter);
  sweep<F_PKADD>("v_pk_add_f32", out, in, niter);
  sweep<F_DSREAD>("ds_read_b128", out, in, niter);
  sweep<F_DSREAD32>("ds_read_b32", out, in, niter);
  sweep<F_DSWRITE>("ds_write_b128", out, in, niter);
  sweep<F_SALU>("s_add_u32", out, in, niter);
  sweep<F_GLOAD>("global_load_dwordx4", out, in, niter);
  sweep<F_GSTORE>("global_store_dword", out, in, niter);
  return 0;
}
