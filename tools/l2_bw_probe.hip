// l2_bw_probe.hip -- what does the L2 of gfx950 deliver to the CUs?  Every workgroup streams the SAME small buffer (fits the
// 4 MB L2 of its XCD, not the CU's vector cache) over and over: by 16-byte loads into registers and by 16-byte LDS-DMA pieces
// (the two ways the GEMMs read their operands).  Also: a buffer that fits nothing (HBM) for comparison.
//   hipcc --offload-arch=gfx950 -O3 -o build/l2_bw_probe tools/l2_bw_probe.hip && ./build/l2_bw_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_regs(const f32x4* __restrict__ buf, size_t n4, int reps, float* sink, int spread) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  // workgroups start at different places of the buffer (spread) so that they do not all hit the same channel at once
  const size_t start = ((size_t)blockIdx.x * 4099u * spread) % n4;
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < n4; i += 256 * 4) {
      const size_t a = (start + i) % n4, b = (start + i + 256) % n4, c = (start + i + 512) % n4, d = (start + i + 768) % n4;
      const f32x4 v0 = buf[a], v1 = buf[b], v2 = buf[c], v3 = buf[d];
      s += v0 + v1 + v2 + v3;
    }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

__global__ __launch_bounds__(256) void read_dma(const f32x4* __restrict__ buf, size_t n4, int reps, float* sink, int spread) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[16384];
  const int wave = threadIdx.x >> 6;
  const size_t start = ((size_t)blockIdx.x * 4099u * spread) % n4;
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < n4; i += 256 * 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t a = (start + i + 256 * j) % n4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + a),
                                         (__attribute__((address_space(3))) void*)(lds + wave * 4096 + j * 1024), 16, 0, 0);
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[threadIdx.x] == 77 && lds[threadIdx.x + 256] == 78) sink[0] = 1.f;
}

int main() {
  float* sink; CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t kb : {512, 2048, 16384, 1048576}) {
    const size_t bytes = kb * 1024, n4 = bytes / 16;
    f32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0x3c, bytes));
    const int wgs = 1024;
    const int reps = (int)((size_t)256 * 1024 * 1024 / bytes) > 0 ? (int)((size_t)256 * 1024 * 1024 / bytes) : 1;   // 256 MB per workgroup... capped below
    const int r = kb >= 1048576 ? 1 : (reps > 64 ? 64 : reps);
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e30f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(read_regs, dim3(kb >= 1048576 ? 4096 : wgs), dim3(256), 0, 0, buf, kb >= 1048576 ? n4 / 4096 : n4, r, sink, 1);
        else hipLaunchKernelGGL(read_dma, dim3(kb >= 1048576 ? 4096 : wgs), dim3(256), 0, 0, buf, kb >= 1048576 ? n4 / 4096 : n4, r, sink, 1);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
      }
      const double total = kb >= 1048576 ? (double)bytes / 4096 * 4096 : (double)bytes * r * wgs;
      printf("%8zu KB buffer, %s: %.3f ms, %.2f TB/s delivered to the CUs%s\n", kb, mode ? "LDS-DMA  " : "registers", best,
             total / best / 1e9, kb >= 1048576 ? "  (every workgroup its own 256 KB slice: HBM)" : "");
    }
    CK(hipFree(buf));
  }
  return 0;
}
