// gemm_x2_tune.hip -- the 2 x fp16 split NT GEMM (wire_gemmx2h.hip, 3 MFMAs per fp32 product) against the 3 x bf16 split
// one (wire_gemmx3h.hip, 6 per product) in one process: agreement of the results (and of both with an fp64 host
// reference on sampled rows), the maximum slots the epilogue fills, and interleaved timing rounds on random operands.
//   bash tools/build_x2_tune.sh && ./build/gemm_x2_tune [N] [P] [rounds]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../wire_amd/csrc/wire_gemm.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale, int gaborlike) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  float v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
  if (gaborlike) {      // wide log-uniform magnitudes, like Gabor activations / gradients
    unsigned y = x * 0x9e3779b9u; y ^= y >> 15;
    v *= __expf(-12.f * (y >> 8) * (1.0f / 16777216.0f));
  }
  p[i] = v;
}
// plain streaming read of n4 float4 (the HBM traffic of an operand, issued by waves that do nothing else)
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ src, float* __restrict__ sink, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) sink[0] = s;
}
static float* dalloc(size_t n, unsigned seed, float scale, int gaborlike = 0) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale, gaborlike);
  return p;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 262144;
  const int P = argc > 2 ? atoi(argv[2]) : 512;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  const float ascale = argc > 4 ? atof(argv[4]) : 1.0f;     // magnitude of A (1e-7: gradient-like)
  float* A = dalloc((size_t)N * P, 1, ascale, 1);
  float* Bt = dalloc((size_t)P * P, 2, 0.06f);
  float* bias = dalloc(P, 3, 0.06f);
  float* lin = dalloc((size_t)N * P, 4, 0.05f);
  float* out = dalloc((size_t)N * P, 5, 1.0f);
  float* o0 = dalloc((size_t)N * P, 6, 0.f);
  float* o1 = dalloc((size_t)N * P, 7, 0.f);
  float* r0 = dalloc((size_t)N * P, 8, 0.f);
  float* Bx3; CK(hipMalloc(&Bx3, (size_t)gemmx3_b_image_floats(P, P) * 4));
  CK(launch_x3_split_b(0, Bt, P, P, P, Bx3));
  float* Bx2; CK(hipMalloc(&Bx2, (size_t)gemmx2_b_image_floats(P, P) * 4));
  unsigned* slots; CK(hipMalloc(&slots, 4 * 64 * 4));      // A, B, out, spare
  CK(hipMemset(slots, 0, 4 * 64 * 4));
  CK(launch_amax(0, A, (int64_t)N * P, slots));
  CK(launch_amax(0, Bt, (int64_t)P * P, slots + 64));
  X2SplitBatch sb{};
  sb.src[0] = Bt; sb.dst[0] = Bx2; sb.slots[0] = slots + 64;
  CK(launch_x2_split_b_batch(0, sb, 1, P, P, P));
  CK(hipDeviceSynchronize());
  {
    unsigned hs[128]; CK(hipMemcpy(hs, slots, 512, hipMemcpyDeviceToHost));
    unsigned ma = 0, mb = 0;
    for (int i = 0; i < 64; ++i) { ma = hs[i] > ma ? hs[i] : ma; mb = hs[64 + i] > mb ? hs[64 + i] : mb; }
    float fa, fb; memcpy(&fa, &ma, 4); memcpy(&fb, &mb, 4);
    printf("max |A| %.6e  max |B| %.6e (from the slots)\n", fa, fb);
  }
  std::vector<float> hA((size_t)4096 * P), hB((size_t)P * P);
  CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hB.data(), Bt, hB.size() * 4, hipMemcpyDeviceToHost));
  for (int64_t Nc : {N, (int64_t)4096 + 37}) {
    if (Nc > N) continue;
    GemmEpiParams ep; ep.ld0 = P; ep.ld1 = P;
    gemmx3g_tune_set("x3_glds", 0);
    gemmx3h_tune_set("x3_h16", 3);
    ep.o0 = r0; CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, Nc, P, P, ep));
    std::vector<float> ref((size_t)Nc * P), got((size_t)Nc * P);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(o0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = o0; ep.amax_a = slots; ep.amax_b = slots + 64;
    CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    CK(hipMemcpy(got.data(), o0, got.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) { md = fmax(md, fabs((double)ref[i] - got[i])); mx = fmax(mx, fabs((double)ref[i])); }
    printf("check rows %lld  2xfp16 vs 3xbf16: max |diff| %.3e of max |C| %.3e (rel %.2e)\n", (long long)Nc, md, mx, md / mx);
    // fp64 reference on 64 sampled rows
    double e3 = 0, e2 = 0, m64 = 0;
    for (int rr = 0; rr < 64; ++rr) {
      const int row = (rr * 61) % 4096;
      for (int j = 0; j < P; ++j) {
        double s = 0;
        for (int k = 0; k < P; ++k) s += (double)hA[(size_t)row * P + k] * hB[(size_t)j * P + k];
        e3 = fmax(e3, fabs(s - ref[(size_t)row * P + j])); e2 = fmax(e2, fabs(s - got[(size_t)row * P + j]));
        m64 = fmax(m64, fabs(s));
      }
    }
    printf("   vs fp64 on 64 rows: 3xbf16 %.3e  2xfp16 %.3e  (relative to max |C| of those rows)\n", e3 / m64, e2 / m64);
    // A straight into registers: the same arithmetic in the same order -> the same bits
    gemmx2h_tune_set("x2_amode", 1);
    CK(hipMemset(r0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = r0; CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    gemmx2h_tune_set("x2_amode", 0);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("   A via registers vs A via LDS: %zu of %zu words differ\n", bad, ref.size());
    gemmx2h_tune_set("x2_amode", 2);
    CK(hipMemset(r0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = r0; CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    gemmx2h_tune_set("x2_amode", 0);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("   A via LDS in whole lines vs A via LDS: %zu of %zu words differ\n", bad, ref.size());
    // 128-row tiles (3 workgroups per CU): the same arithmetic per element -> the same bits
    gemmx2h_tune_set("x2_rows128", 1);
    CK(hipMemset(r0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = r0; CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    gemmx2h_tune_set("x2_rows128", 0);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("   128-row tiles vs 256-row tiles: %zu of %zu words differ\n", bad, ref.size());
    gemmx2h_tune_set("x2_rows128", 3);
    CK(hipMemset(r0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = r0; CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    gemmx2h_tune_set("x2_rows128", 0);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("   weight ring of 3 vs default: %zu of %zu words differ\n", bad, ref.size());
    gemmx2h_tune_set("x2_rows128", 2);
    CK(hipMemset(r0, 0xff, (size_t)Nc * P * 4));
    ep.o0 = r0; CK(launch_gemmx2h_nt(0, EPI_STORE, A, P, Bx2, Nc, P, P, ep));
    gemmx2h_tune_set("x2_rows128", 0);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
    printf("   128-row tiles + 2-stage prefetch vs 256-row tiles: %zu of %zu words differ\n", bad, ref.size());
  }
  // gabor forward epilogue + maximum slots of its output
  {
    GemmEpiParams ep;
    ep.bias = bias; ep.o0 = o0; ep.o1 = o1; ep.ld0 = P; ep.ld1 = P; ep.omega = 20.f; ep.scale = 30.f; ep.kvalid = P / 2;
    gemmx3h_tune_set("x3_h16", 3);
    CK(launch_gemmx3_nt(0, EPI_GABOR_FWD, A, P, Bx3, N, P, P, ep));
    std::vector<float> ref((size_t)N * P), got((size_t)N * P);
    CK(hipMemcpy(ref.data(), o1, ref.size() * 4, hipMemcpyDeviceToHost));
    ep.amax_a = slots; ep.amax_b = slots + 64; ep.amax_out = slots + 128;
    CK(launch_gemmx2h_nt(0, EPI_GABOR_FWD, A, P, Bx2, N, P, P, ep));
    CK(hipMemcpy(got.data(), o1, got.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) { md = fmax(md, fabs((double)ref[i] - got[i])); mx = fmax(mx, fabs((double)got[i])); }
    unsigned hs[64]; CK(hipMemcpy(hs, slots + 128, 256, hipMemcpyDeviceToHost));
    unsigned mo = 0; for (int i = 0; i < 64; ++i) mo = hs[i] > mo ? hs[i] : mo;
    float fo; memcpy(&fo, &mo, 4);
    printf("gabor_fwd out: 2xfp16 vs 3xbf16 max |diff| %.3e of max %.3e; slots say max |out| = %.9e (host: %.9e)\n", md, mx, fo, mx);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;
  struct Var { const char* name; int fam; int epi; int recompute; int amode; };
  std::vector<Var> vars = {
      {"x3h store", 3, EPI_STORE, 0, 0}, {"x2h store (A via LDS)", 2, EPI_STORE, 0, 0}, {"x2h store (A via regs)", 2, EPI_STORE, 0, 1},
      {"x3h gabor_fwd", 3, EPI_GABOR_FWD, 0, 0}, {"x2h gabor_fwd (LDS)", 2, EPI_GABOR_FWD, 0, 0},
      {"x2h gabor_fwd (regs)", 2, EPI_GABOR_FWD, 0, 1},
      {"x3h gabor_bwd recompute", 3, EPI_GABOR_BWD, 1, 0}, {"x2h gabor_bwd recompute (LDS)", 2, EPI_GABOR_BWD, 1, 0},
      {"x2h gabor_bwd recompute (regs)", 2, EPI_GABOR_BWD, 1, 1},
      {"x2h gabor_fwd no out (LDS)", 2, EPI_GABOR_FWD, 2, 0}, {"x2h gabor_fwd no out (regs)", 2, EPI_GABOR_FWD, 2, 1},
      {"x2h store (128-row tiles)", 2, EPI_STORE, 0, 101}, {"x2h gabor_fwd (128-row tiles)", 2, EPI_GABOR_FWD, 0, 101},
      {"x2h gabor_bwd rec (128-row tiles)", 2, EPI_GABOR_BWD, 1, 101}, {"x2h fwd no out (128-row tiles)", 2, EPI_GABOR_FWD, 2, 101},
      {"x2h store (128-row, prefetch 2)", 2, EPI_STORE, 0, 201}, {"x2h gabor_fwd (128-row, prefetch 2)", 2, EPI_GABOR_FWD, 0, 201},
      {"x2h gabor_bwd rec (128-row, pf 2)", 2, EPI_GABOR_BWD, 1, 201}, {"x2h fwd no out (128-row, pf 2)", 2, EPI_GABOR_FWD, 2, 201},
      // probe (results wrong): A addressed as if stored k-slab-major inside 256-row tiles (a stage = one contiguous 32 KB)
      {"x2h store, tiled-A addressing [probe]", 2, EPI_STORE, 0, 1001}, {"x2h fwd no out, tiled-A [probe]", 2, EPI_GABOR_FWD, 2, 1001},
      // probe (results wrong): the loaded 32 bytes used as the fp16 fragments themselves (a producer-side split)
      {"x2h store, A pre-split [probe]", 2, EPI_STORE, 0, 2001}, {"x2h gabor_fwd, A pre-split [probe]", 2, EPI_GABOR_FWD, 0, 2001},
      {"x2h fwd no out, A pre-split [probe]", 2, EPI_GABOR_FWD, 2, 2001}, {"x2h gabor_bwd rec, A pre-split [probe]", 2, EPI_GABOR_BWD, 1, 2001},
      // A through LDS in whole cache lines (8 rows x 128 B per DMA instruction, source-side swizzle): bit-identical
      {"x2h store (LDS, whole lines)", 2, EPI_STORE, 0, 2}, {"x2h gabor_fwd (LDS, whole lines)", 2, EPI_GABOR_FWD, 0, 2},
      {"x2h gabor_bwd rec (LDS, whole lines)", 2, EPI_GABOR_BWD, 1, 2}, {"x2h fwd no out (LDS, whole lines)", 2, EPI_GABOR_FWD, 2, 2},
      // the same bytes (lin + out written) with cheaper epilogue arithmetic: max(lin, 0) / one sine per element
      {"x2h relu_fwd (LDS, whole lines)", 2, EPI_RELU_FWD, 0, 2}, {"x2h siren_fwd (LDS, whole lines)", 2, EPI_SIREN_FWD, 0, 2},
      // weight pieces two stages ahead in a ring of three buffers (bit-identical)
      {"x2h store (weight ring 3)", 2, EPI_STORE, 0, 301}, {"x2h gabor_fwd (weight ring 3)", 2, EPI_GABOR_FWD, 0, 301},
      {"x2h gabor_bwd rec (weight ring 3)", 2, EPI_GABOR_BWD, 1, 301}, {"x2h fwd no out (weight ring 3)", 2, EPI_GABOR_FWD, 2, 301},
      // probes (results wrong): the stage-end synchronisation taken out
      {"x2h store, no load wait [probe]", 2, EPI_STORE, 0, 3001}, {"x2h store, no barrier [probe]", 2, EPI_STORE, 0, 4001},
      {"x2h store, no wait no barrier [probe]", 2, EPI_STORE, 0, 5001}, {"x2h gabor_fwd, no wait no barrier [probe]", 2, EPI_GABOR_FWD, 0, 5001},
      {"x2h gabor_bwd rec, no wait no barrier [probe]", 2, EPI_GABOR_BWD, 1, 5001},
      {"x2h store, loads not waited for [probe]", 2, EPI_STORE, 0, 6001}, {"x2h gabor_fwd, loads not waited for [probe]", 2, EPI_GABOR_FWD, 0, 6001},
      {"x2h fwd no out, loads not waited for [probe]", 2, EPI_GABOR_FWD, 2, 6001},
      {"x2h store, loads (to a register) not waited for [probe]", 2, EPI_STORE, 0, 7001},
      {"x2h store (LDS), whole-line DMA pieces [probe]", 2, EPI_STORE, 0, 9000}, {"x2h gabor_fwd (LDS), whole-line DMA [probe]", 2, EPI_GABOR_FWD, 0, 9000},
      {"x2h fwd no out (LDS), whole-line DMA [probe]", 2, EPI_GABOR_FWD, 2, 9000},
      {"x2h store, A = tile 0 for all (L2 hits, real pattern) [probe]", 2, EPI_STORE, 0, 8001},
      {"x2h fwd no out, A = tile 0 for all [probe]", 2, EPI_GABOR_FWD, 2, 8001},
      // ablation (results wrong): every row of A is row 0 -> all A loads hit the caches
      {"x2h store (LDS), A cached [abl]", 2, EPI_STORE, 0, 10}, {"x2h store (regs), A cached [abl]", 2, EPI_STORE, 0, 11},
  };
  if (const char* only = getenv("X2_ONLY")) {            // keep the variants whose name contains this
    std::vector<Var> keep;
    for (const Var& v : vars) if (strstr(v.name, only)) keep.push_back(v);
    vars.swap(keep);
  }
  std::vector<double> best(vars.size(), 1e30), sum(vars.size(), 0);
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t v = 0; v < vars.size(); ++v) {
      const Var& V = vars[v];
      GemmEpiParams ep;
      ep.bias = bias; ep.o0 = o0; ep.o1 = V.recompute == 2 ? nullptr : o1; ep.i0 = lin; ep.i1 = out; ep.ld0 = P; ep.ld1 = P;
      ep.omega = 20.f; ep.scale = 30.f; ep.kvalid = P / 2;
      ep.recompute_out = V.recompute == 1;
      ep.amax_a = slots; ep.amax_b = slots + 64; ep.amax_out = slots + 192;
      if (V.amode >= 7000) ep.kvalid = -783;
      ep.stagger = V.amode >= 9000 ? 785 : V.amode >= 8000 ? 784 : V.amode >= 6000 ? 782 : V.amode >= 5000 ? 781 : V.amode >= 4000 ? 780 : V.amode >= 3000 ? 779 : V.amode >= 2000 ? 778 : (V.amode >= 1000 ? 777 : 0);
      gemmx2h_tune_set("x2_amode", V.amode % 10);
      gemmx2h_tune_set("x2_rows128", V.amode >= 1000 ? 0 : (V.amode >= 300 ? 3 : (V.amode >= 200 ? 2 : (V.amode >= 100 ? 1 : 0))));
      const int lda = (V.amode >= 10 && V.amode < 100) ? 0 : P;
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) {
        if (V.fam == 3) CK(launch_gemmx3_nt(0, V.epi, A, P, Bx3, N, P, P, ep));
        else CK(launch_gemmx2h_nt(0, V.epi, A, lda, Bx2, N, P, P, ep));
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 4;
      if (r > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
    }
  }
  // ---- who pays for the HBM reads?  The store GEMM with every A row served from cache (no HBM read of A) on one stream,
  // a plain streaming read of A's 0.54 GB by OTHER waves on a second stream, each alone and both at once: if the pair takes
  // what the cached GEMM takes alone, HBM traffic is cheap when the MFMA waves do not issue it themselves; if it takes what
  // the real GEMM takes (A via HBM), the bytes cost the same whoever moves them
  {
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    GemmEpiParams ep; ep.ld0 = P; ep.ld1 = P; ep.amax_a = slots; ep.amax_b = slots + 64; ep.o0 = o0;
    gemmx2h_tune_set("x2_amode", 1); gemmx2h_tune_set("x2_rows128", 0);
    const size_t n4 = (size_t)N * P / 4;
    auto wall = [&](int what) {
      CK(hipDeviceSynchronize());
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      const auto t0 = std::chrono::steady_clock::now();
      for (int q = 0; q < 4; ++q) {
        if (what & 1) CK(launch_gemmx2h_nt(s1, EPI_STORE, A, (what & 4) ? P : 0, Bx2, N, P, P, ep));
        if (what & 2) hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, s2, (const float4*)lin, r0, n4);
      }
      CK(hipDeviceSynchronize());
      (void)a; (void)b;
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 4;
    };
    double best[8]; for (double& x : best) x = 1e30;
    for (int r = 0; r < rounds + 1; ++r)
      for (int w : {1, 2, 3, 5, 7}) { const double t = wall(w); if (r > 0 && t < best[w]) best[w] = t; }
    printf("  [who pays] store GEMM, A cached: %.3f ms | stream read of 0.54 GB alone: %.3f ms | both at once: %.3f ms | "
           "store GEMM, A from HBM: %.3f ms | that + the stream read: %.3f ms\n", best[1], best[2], best[3], best[5], best[7]);
  }
  // ---- does the 256 MiB Infinity Cache pay for a chunk-major schedule?  Two chained store GEMMs (A -> o0 -> o1), layer by
  // layer over all rows against chunk by chunk (each chunk's o0 slice is consumed right after it was produced)
  {
    GemmEpiParams ep; ep.ld0 = P; ep.ld1 = P; ep.amax_a = slots; ep.amax_b = slots + 64;
    gemmx2h_tune_set("x2_amode", 1); gemmx2h_tune_set("x2_rows128", 0);
    for (int nchunk : {1, 4, 8, 16}) {
      const int64_t rows = N / nchunk;
      double tsum = 0;
      for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int c = 0; c < nchunk; ++c) {
          ep.o0 = o0 + (size_t)c * rows * P; CK(launch_gemmx2h_nt(0, EPI_STORE, A + (size_t)c * rows * P, P, Bx2, rows, P, P, ep));
          ep.o0 = o1 + (size_t)c * rows * P; CK(launch_gemmx2h_nt(0, EPI_STORE, o0 + (size_t)c * rows * P, P, Bx2, rows, P, P, ep));
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) tsum += ms;
      }
      printf("  two chained store GEMMs, %2d row chunk(s) of %lld rows: %7.3f ms for both layers\n", nchunk, (long long)rows, tsum / rounds);
    }
  }
  // ---- weight gradient (TN): slab[s] = G^T Z over row splits, 3 x bf16 (gemmx3_tn16) against 2 x fp16 (gemmx2_tn16)
  if (gemmx2_tn_applies(P, P)) {
    const int S = gemmx3_tn_splits(N, P, P, 256), S2 = gemmx2_tn_splits(N, P, P, 256);
    float *slab3, *slab2, *bs3, *bs2;
    CK(hipMalloc(&slab3, (size_t)S * P * P * 4)); CK(hipMalloc(&slab2, (size_t)S2 * P * P * 4));
    CK(hipMalloc(&bs3, (size_t)S * P * 4)); CK(hipMalloc(&bs2, (size_t)S2 * P * 4));
    CK(hipMemset(slots + 128, 0, 256));
    CK(launch_amax(0, out, (int64_t)N * P, slots + 128));
    for (int64_t Nr : {N, (int64_t)4096 + 37}) {
      if (Nr > N) continue;
      const int Sr = gemmx3_tn_splits(Nr, P, P, 256), Sr2 = gemmx2_tn_splits(Nr, P, P, 256);
      CK(launch_gemmx3_tn(0, A, P, out, P, Nr, P, P, Sr, slab3, bs3));
      CK(launch_gemmx2_tn(0, A, P, out, P, Nr, P, P, Sr2, slab2, bs2, slots, slots + 128));
      std::vector<float> a((size_t)Sr * P * P), b2((size_t)Sr2 * P * P), ba((size_t)Sr * P), bb((size_t)Sr2 * P);
      CK(hipMemcpy(a.data(), slab3, a.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b2.data(), slab2, b2.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ba.data(), bs3, ba.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(bb.data(), bs2, bb.size() * 4, hipMemcpyDeviceToHost));
      // sum over the splits (what the reduce kernel does) in double
      std::vector<double> s3((size_t)P * P, 0.0), s2((size_t)P * P, 0.0);
      for (int sp = 0; sp < Sr; ++sp)
        for (size_t i = 0; i < (size_t)P * P; ++i) s3[i] += a[(size_t)sp * P * P + i];
      for (int sp = 0; sp < Sr2; ++sp)
        for (size_t i = 0; i < (size_t)P * P; ++i) s2[i] += b2[(size_t)sp * P * P + i];
      std::vector<double> b3s(P, 0.0), b2s(P, 0.0);
      for (int sp = 0; sp < Sr; ++sp) for (int i = 0; i < P; ++i) b3s[i] += ba[(size_t)sp * P + i];
      for (int sp = 0; sp < Sr2; ++sp) for (int i = 0; i < P; ++i) b2s[i] += bb[(size_t)sp * P + i];
      double md = 0, mx = 0, mb = 0;
      for (size_t i = 0; i < s3.size(); ++i) { md = fmax(md, fabs(s3[i] - s2[i])); mx = fmax(mx, fabs(s3[i])); }
      for (int i = 0; i < P; ++i) mb = fmax(mb, fabs(b3s[i] - b2s[i]));
      printf("TN rows %lld (%d splits): summed slabs 2xfp16 vs 3xbf16 max |diff| %.3e of max %.3e (rel %.2e); bias slabs max |diff| %.3e\n",
             (long long)Nr, Sr, md, mx, md / mx, mb);
      if (Nr == 4096 + 37) {   // fp64 reference of a few entries
        std::vector<float> hG((size_t)Nr * P), hZ((size_t)Nr * P);
        CK(hipMemcpy(hG.data(), A, hG.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hZ.data(), out, hZ.size() * 4, hipMemcpyDeviceToHost));
        double e3 = 0, e2 = 0, m64 = 0;
        for (int t = 0; t < 256; ++t) {
          const int m = (t * 37) % P, c = (t * 101 + 7) % P;
          double r = 0;
          for (int64_t i = 0; i < Nr; ++i) r += (double)hG[(size_t)i * P + m] * hZ[(size_t)i * P + c];
          e3 = fmax(e3, fabs(r - s3[(size_t)m * P + c])); e2 = fmax(e2, fabs(r - s2[(size_t)m * P + c])); m64 = fmax(m64, fabs(r));
        }
        printf("   vs fp64 on 256 entries: 3xbf16 %.3e  2xfp16 %.3e (relative to their max)\n", e3 / m64, e2 / m64);
      }
    }
    double t3 = 0, t2 = 0;
    for (int r = 0; r < rounds + 1; ++r) {
      float ms;
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) CK(launch_gemmx3_tn(0, A, P, out, P, N, P, P, S, slab3, bs3));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) t3 += ms / 4;
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) CK(launch_gemmx2_tn(0, A, P, out, P, N, P, P, S2, slab2, bs2, slots, slots + 128));
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) t2 += ms / 4;
    }
    printf("  %-36s mean %7.3f ms  (%d row splits) -> %6.1f alg TF\n", "x3 tn16 wgrad", t3 / rounds, S, flop / (t3 / rounds * 1e-3) / 1e12);
    printf("  %-36s mean %7.3f ms  (%d row splits) -> %6.1f alg TF\n", "x2 tn16 wgrad", t2 / rounds, S2, flop / (t2 / rounds * 1e-3) / 1e12);
  }
  printf("N=%lld P=%d  (%.1f algorithmic GFLOP per launch)\n", (long long)N, P, flop / 1e9);
  for (size_t v = 0; v < vars.size(); ++v) {
    const double tf = flop / (sum[v] / rounds * 1e-3) / 1e12;
    printf("  %-36s mean %7.3f ms  min %7.3f ms  -> %6.1f alg TF\n", vars[v].name, sum[v] / rounds, best[v], tf);
  }
  return 0;
}
