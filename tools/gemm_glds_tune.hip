// gemm_glds_tune.hip -- A/B of the split-bf16 NT GEMM editions in one process (interleaved rounds, random
// operands, HIP events): register-staged loader (wire_gemmx3.hip, "x3") against the LDS-DMA editions
// (wire_gemmx3g.hip: mode 1 = 256 x 128 tile / 4 waves / 2 buffers, mode 2 = 256 x 256 / 8 waves / 3 buffers).
// The editions run the same MFMA sequence per accumulator, so their outputs must agree bit for bit.
//   bash tools/build_glds_tune.sh && ./build/gemm_glds_tune [N] [P] [rounds]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../wire_amd/csrc/wire_gemm.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)(i * 2654435761u) ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
static float* dalloc(size_t n, unsigned seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale);
  return p;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 262144;
  const int P = argc > 2 ? atoi(argv[2]) : 512;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  float* A = dalloc((size_t)N * P, 1, 1.0f);
  float* Bt = dalloc((size_t)P * P, 2, 0.06f);
  float* bias = dalloc(P, 3, 0.06f);
  float* lin = dalloc((size_t)N * P, 4, 0.5f);
  float* out = dalloc((size_t)N * P, 5, 1.0f);
  float* o0 = dalloc((size_t)N * P, 6, 0.f);
  float* o1 = dalloc((size_t)N * P, 7, 0.f);
  float* r0 = dalloc((size_t)N * P, 8, 0.f);
  float* Bx3; CK(hipMalloc(&Bx3, (size_t)gemmx3_b_image_floats(P, P) * 4));
  CK(launch_x3_split_b(0, Bt, P, P, P, Bx3));
  CK(hipDeviceSynchronize());
  // ---- bit-exact agreement of the editions (store epilogue, full and ragged row counts)
  for (int64_t Nc : {N, (int64_t)4096 + 37}) {
    if (Nc > N) continue;
    GemmEpiParams ep; ep.ld0 = P; ep.ld1 = P;
    gemmx3g_tune_set("x3_glds", 0);
    ep.o0 = r0; CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, Nc, P, P, ep));
    std::vector<float> ref((size_t)Nc * P), got((size_t)Nc * P);
    CK(hipMemcpy(ref.data(), r0, ref.size() * 4, hipMemcpyDeviceToHost));
    gemmx3h_tune_set("x3_h16", 0);
    {
      // the 16 x 16 x 32 edition sums the same six products in another grouping: fp32 round-off, not bit equality
      gemmx3h_tune_set("x3_h16", 3);
      CK(hipMemset(o0, 0xff, (size_t)Nc * P * 4));
      ep.o0 = o0; CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, Nc, P, P, ep));
      CK(hipMemcpy(got.data(), o0, got.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (size_t i = 0; i < ref.size(); ++i) { md = fmax(md, fabs((double)ref[i] - got[i])); mx = fmax(mx, fabs((double)ref[i])); }
      printf("check rows %lld  16x16x32 edition vs register-staged: max |diff| %.3e of max |C| %.3e (rel %.2e)\n",
             (long long)Nc, md, mx, md / mx);
      gemmx3h_tune_set("x3_h16", 0);
    }
    for (int mode = 1; mode <= 2; ++mode) {
      gemmx3g_tune_set("x3_glds", mode);
      CK(hipMemset(o0, 0xff, (size_t)Nc * P * 4));
      ep.o0 = o0; CK(launch_gemmx3_nt(0, EPI_STORE, A, P, Bx3, Nc, P, P, ep));
      CK(hipMemcpy(got.data(), o0, got.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; double md = 0;
      for (size_t i = 0; i < ref.size(); ++i) {
        if (memcmp(&ref[i], &got[i], 4)) { ++bad; md = fmax(md, fabs((double)ref[i] - got[i])); }
      }
      printf("check rows %lld  glds mode %d vs register-staged: %zu of %zu words differ (max |diff| %.3e)\n",
             (long long)Nc, mode, bad, ref.size(), md);
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * N * P * P;
  struct Var { const char* name; int mode; int epi; int stagger; };
  std::vector<Var> vars = {
      {"x3h 16x16x32 store", 10, EPI_STORE, 0}, {"x3h 16x16x32 gabor_fwd", 10, EPI_GABOR_FWD, 0},
      {"x3h 16x16x32 gabor_bwd", 10, EPI_GABOR_BWD, 0},
      {"x3h gabor_bwd recompute out", 10, EPI_GABOR_BWD, -100},
      {"x3h fwd stagger 15us", 10, EPI_GABOR_FWD, 1500}, {"x3h fwd stagger 30us", 10, EPI_GABOR_FWD, 3000},
      {"x3h bwd stagger 15us", 10, EPI_GABOR_BWD, 1500}, {"x3h bwd stagger 30us", 10, EPI_GABOR_BWD, 3000},
      {"x3h store stagger 30us", 10, EPI_STORE, 3000},
      {"x3  regs  store", 0, EPI_STORE, 0}, {"x3g m1    store", 1, EPI_STORE, 0}, {"x3g m2    store", 2, EPI_STORE, 0},
      {"x3g m1 store stag 20us", 1, EPI_STORE, 2000}, {"x3g m1 store stag 40us", 1, EPI_STORE, 4000},
      {"x3g m1 store stag 60us", 1, EPI_STORE, 6000},
      {"x3  regs  gabor_fwd", 0, EPI_GABOR_FWD, 0}, {"x3g m1    gabor_fwd", 1, EPI_GABOR_FWD, 0},
      {"x3g m1 fwd stag 20us", 1, EPI_GABOR_FWD, 2000}, {"x3g m1 fwd stag 40us", 1, EPI_GABOR_FWD, 4000},
      {"x3g m1 fwd stag 60us", 1, EPI_GABOR_FWD, 6000},
      {"x3  regs  gabor_bwd", 0, EPI_GABOR_BWD, 0}, {"x3g m1    gabor_bwd", 1, EPI_GABOR_BWD, 0},
      {"x3g m1 bwd stag 40us", 1, EPI_GABOR_BWD, 4000},
      // ablations (results wrong; stagger field = 0, flags ride in stagger_lo): 1 no DMA, 2 no epilogue, 4 no fragment reads
      {"x3g m1 store noDMA", 1, EPI_STORE, -1}, {"x3g m1 store noEPI", 1, EPI_STORE, -2},
      {"x3g m1 store noDMA+noEPI", 1, EPI_STORE, -3}, {"x3g m1 store noREADS", 1, EPI_STORE, -4},
      {"x3g m1 store MFMA+VALU only", 1, EPI_STORE, -7},
      {"x3g m2 store noEPI", 2, EPI_STORE, -2}, {"x3g m2 store MFMA+VALU only", 2, EPI_STORE, -7},
      {"x3g m3 (1 wave/SIMD) store", 3, EPI_STORE, 0}, {"x3g m3 store noEPI", 3, EPI_STORE, -2},
      {"x3g m3 store MFMA+VALU only", 3, EPI_STORE, -7},
  };
  std::vector<double> best(vars.size(), 1e30), sum(vars.size(), 0);
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t v = 0; v < vars.size(); ++v) {
      const Var& V = vars[v];
      GemmEpiParams ep;
      ep.bias = bias; ep.o0 = o0; ep.o1 = o1; ep.i0 = lin; ep.i1 = out; ep.ld0 = P; ep.ld1 = P;
      ep.omega = 20.f; ep.scale = 30.f; ep.kvalid = P / 2;
      if (V.stagger == -100) ep.recompute_out = 1;     // data gradient that recomputes out from lin
      gemmx3g_tune_set("x3_glds", V.mode == 10 ? 0 : V.mode);
      gemmx3h_tune_set("x3_h16", V.mode == 10 ? 3 : 0);
      gemmx3h_tune_set("x3h_stagger", V.mode == 10 && V.stagger > 0 ? V.stagger : 0);
      gemmx3g_tune_set("x3_stagger", V.stagger > 0 ? V.stagger : 0);
      gemmx3g_tune_set("x3_stagger_lo", V.stagger < 0 ? -V.stagger : 256);
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) CK(launch_gemmx3_nt(0, V.epi, A, P, Bx3, N, P, P, ep));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 4;
      if (r > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
    }
  }
  // ---- weight-gradient (TN) kernel: slab[s] = G^T Z over row splits
  {
    const int S = gemmx3_tn_splits(N, P, P, 192);
    float* slab; CK(hipMalloc(&slab, (size_t)S * P * P * 4));
    float* bslab; CK(hipMalloc(&bslab, (size_t)S * P * 4));
    double tsum = 0;
    for (int r = 0; r < rounds + 1; ++r) {
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 4; ++q) CK(launch_gemmx3_tn(0, A, P, out, P, N, P, P, S, slab, bslab));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) tsum += ms / 4;
    }
    printf("  %-22s mean %7.3f ms  (%d row splits)  -> %6.1f alg TF  frac %.3f\n", "x3 tn wgrad", tsum / rounds, S,
           flop / (tsum / rounds * 1e-3) / 1e12, flop / (tsum / rounds * 1e-3) / 1e12 / 416.7);
  }
  printf("N=%lld P=%d  (%.1f algorithmic GFLOP per launch; split roofline 416.7 TF)\n", (long long)N, P, flop / 1e9);
  for (size_t v = 0; v < vars.size(); ++v) {
    const double tf = flop / (sum[v] / rounds * 1e-3) / 1e12;
    printf("  %-22s mean %7.3f ms  min %7.3f ms  -> %6.1f alg TF  frac %.3f\n", vars[v].name, sum[v] / rounds, best[v],
           tf, tf / 416.7);
  }
  return 0;
}
