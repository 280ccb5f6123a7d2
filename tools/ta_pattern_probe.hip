// ta_pattern_probe.hip -- what does the lane -> address mapping of a 16-byte-per-lane load cost on the vector-memory path?
// Every wave reads 1 KB per instruction out of rows that lie 2 KB apart (a [rows][512] fp32 matrix, 128 bytes = one 32-k stage
// per row), from a buffer that stays in the L2 (so the L2 / HBM traffic is the same for every pattern and only the request
// handling differs).  Patterns = the mappings the GEMM kernels of wire_gemmx2h.hip / wire_gemmh_epi.h use or used:
//   frag   lane = (row l & 15, k slot l >> 4): 16 rows x 16 B per 16-lane group  (AMODE 1 register loads)
//   half   lane = (k slot l >> 5, half (l >> 4) & 1, row l & 15): the same per 16-lane group (AMODE 0 LDS-DMA pieces)
//   line8  lane = (row l >> 3, chunk l & 7): 2 whole lines per 16-lane group  (AMODE 2)
//   epi    lane = (row l & 7, chunk 4 ((l >> 3) & 1) + (l >> 4)): 8 rows x 2 chunks per 16-lane group  (epilogue loads / stores)
//   tn     lane = (row (l >> 2) & 7, chunk 4 (l >> 5) + (l & 3)): 4 rows x 64 B per 16-lane group  (weight-gradient loader)
//   linear lane = chunk l of one 1 KB run
//   hipcc --offload-arch=gfx950 -O3 -o build/ta_pattern_probe tools/ta_pattern_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool STORE>
__global__ __launch_bounds__(256) void pat_kernel(char* __restrict__ buf, int rows, int reps, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int row, chunk, rows_per_instr;
  if (PAT == 0) { row = lane & 15; chunk = (lane >> 4) * 2; rows_per_instr = 16; }
  else if (PAT == 1) { row = lane & 15; chunk = (lane >> 5) * 2 + ((lane >> 4) & 1); rows_per_instr = 16; }
  else if (PAT == 2) { row = lane >> 3; chunk = lane & 7; rows_per_instr = 8; }
  else if (PAT == 3) { row = lane & 7; chunk = 4 * ((lane >> 3) & 1) + (lane >> 4); rows_per_instr = 8; }
  else if (PAT == 4) { row = (lane >> 2) & 7; chunk = 4 * (lane >> 5) + (lane & 3); rows_per_instr = 8; }
  else { row = 0; chunk = lane; rows_per_instr = 0; }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  // the workgroup walks its own window of 256 rows (512 KB) -- L2-resident across the repetitions
  const size_t base = ((size_t)blockIdx.x * 256 % rows) * 2048;
  for (int r = 0; r < reps; ++r)
    for (int stage = 0; stage < 16; ++stage) {
      // 64 rows per wave and stage, as in the GEMM
      const int ninstr = PAT == 5 ? 8 : 64 / rows_per_instr * (PAT <= 1 ? 2 : 1);
#pragma unroll 8
      for (int i = 0; i < ninstr; ++i) {
        size_t off;
        if (PAT == 5) off = base + ((size_t)(wave * 64) * 2048) + (size_t)stage * 8192 + i * 1024 + lane * 16;   // 8 KB linear per stage
        else if (PAT <= 1) off = base + (size_t)(wave * 64 + (i >> 1) * 16 + row) * 2048 + stage * 128 + (PAT == 0 ? chunk * 16 + (i & 1) * 16 : (chunk + (i & 1) * 4) * 16);
        else off = base + (size_t)(wave * 64 + i * rows_per_instr + row) * 2048 + stage * 128 + chunk * 16;
        if (STORE) *reinterpret_cast<f32x4*>(buf + off) = s;
        else s += *reinterpret_cast<const f32x4*>(buf + off);
      }
    }
  if (!STORE && s[0] + s[1] + s[2] + s[3] == 123.456f) sink[0] = s[0];
}

template <int PAT, bool STORE>
static void run(const char* name, char* buf, int rows, float* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wgs = 1024, reps = 8;
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((pat_kernel<PAT, STORE>), dim3(wgs), dim3(256), 0, 0, buf, rows, reps, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double bytes = (double)wgs * reps * 16 * 4 * 8192;   // per workgroup and repetition: 16 stages x 4 waves x 8 KB
  printf("%-7s %s: %.3f ms  %.2f TB/s\n", name, STORE ? "stores" : "loads ", best, bytes / best / 1e9);
}

int main() {
  const int rows = 2048;                       // 4 MB: 8 windows of 256 rows, re-read by 128 workgroups each
  char* buf; CK(hipMalloc(&buf, (size_t)rows * 2048)); CK(hipMemset(buf, 0, (size_t)rows * 2048));
  float* sink; CK(hipMalloc(&sink, 64));
  run<5, false>("linear", buf, rows, sink); run<2, false>("line8", buf, rows, sink); run<0, false>("frag", buf, rows, sink);
  run<1, false>("half", buf, rows, sink); run<3, false>("epi", buf, rows, sink); run<4, false>("tn", buf, rows, sink);
  run<5, true>("linear", buf, rows, sink); run<2, true>("line8", buf, rows, sink); run<3, true>("epi", buf, rows, sink);
  run<4, true>("tn", buf, rows, sink);
  return 0;
}
