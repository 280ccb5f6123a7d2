// hbm_write_probe.hip -- HBM bandwidth of plain streaming kernels on this chip: read only, write only, copy, and a 1 : 2
// read : write mix (the forward GEMM's own ratio: 0.54 GB of activations in, 1.07 GB of lin + out back).
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_write_probe tools/hbm_write_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0 read, 1 write, 2 copy (1 : 1), 3 read 1 : write 2
__global__ __launch_bounds__(256) void stream(const f32x4* __restrict__ src, f32x4* __restrict__ dst, f32x4* __restrict__ dst2,
                                             size_t n4, int mode, float* sink) {
  f32x4 s = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    if (mode != 1) s += src[i];
    if (mode >= 1) dst[i] = s;
    if (mode == 3) dst2[i] = s;
  }
  if (mode == 0 && s[0] == 123.456f) sink[0] = s[1];
}

int main() {
  const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
  f32x4 *a, *b, *c; float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0x3c, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"read 1 GiB", "write 1 GiB", "copy 1 GiB -> 1 GiB", "read 1 GiB, write 2 GiB"};
  const double moved[4] = {1.0, 1.0, 2.0, 3.0};
  for (int mode = 0; mode < 4; ++mode)
    for (int wgs : {2048, 8192}) {
      float best = 1e30f;
      for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stream, dim3(wgs), dim3(256), 0, 0, a, b, c, n4, mode, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
      }
      printf("%-26s %5d workgroups: %.3f ms  %.2f TB/s\n", names[mode], wgs, best, moved[mode] * bytes / best / 1e9);
    }
  return 0;
}
