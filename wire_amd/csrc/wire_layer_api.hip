// wire_layer_api.hip -- per-layer entry points of the C ABI:
// ComplexGaborLayer.forward / its backward and the final complex Linear + .real
// on native (interleaved complex64) tensors.  These serve `model.net[i](x)`
// (reference modules/utils.py:246-252) and standalone layers; the training hot
// path is wire_mlp_fwd / wire_mlp_bwd, which never leaves the blocked layout.
// The backward recomputes the layer's forward from x (one extra GEMM) instead
// of asking the caller to keep lin/act in a private layout.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/wire_hip.h"
#include "wire_gemm.h"
#include "wire_point.h"

extern int wire_fail_(int code, const char* msg);   // wire_api.hip
// GEMM family the tuning flags select (wire_api.hip): 2 = split-bf16 (wire_gemmx3.hip, every kind),
// 1 = 3-multiplication complex fp32 MFMA (wire_gemm3m.hip, ComplexGaborLayer only), 0 = 4M fp32 MFMA.
// The per-layer entry points run the SAME kernels as wire_mlp_fwd / wire_mlp_bwd, so the per-layer parity
// tests (protocol step (i)) check the code the bench times.
extern int wire_family_(int kind);
extern int wire_split_f16_();       // wire_api.hip: the "split_f16" knob
enum { FAM_4M = 0, FAM_3M = 1, FAM_X3 = 2 };
#ifndef WIRE_AMAX_SLOTS
#define WIRE_AMAX_SLOTS 64
#endif

namespace {
inline int rup(int v, int m) { return (v + m - 1) / m * m; }
inline int64_t rup64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct LayerWs {
  int Pin, Pout, S;
  int64_t xb, lin, out, gact, glin, gxb, gu, btf, btd, bias, slab, bslab, fpw, fpb, crp, wf, bfr, btf_x3, btd_x3, x2, total;
};
// 2 x fp16 family in the per-layer entry points (split_f16, batches the 16 x 16 x 32 kernels take): a region of the
// workspace holds 4 maximum-slot sets (A operand, weights, G, Z) and the two split weight images.  The operand maxima of
// the fused path come from the producers' epilogues; here the tensors arrive from the caller, so one reduction pass
// (launch_amax) per operand precedes the GEMM.
inline int64_t x2_region_floats(int Pout_g, int Pin) {
  return 4 * WIRE_AMAX_SLOTS + gemmx2_b_image_floats(Pout_g, Pin) + gemmx2_b_image_floats(Pin, Pout_g);
}
// weight-gradient row splits of a family (the workspace is sized for the largest: its size must not depend on
// the tuning flags)
inline int layer_splits(int fam, int64_t n, int Pout, int Pin) {
  if (fam == FAM_X3) return gemmx3_tn_splits(n, Pout, Pin, 256);
  if (fam == FAM_3M) return gemm3m_tn_splits(n, Pout / 2, Pin / 2, 64);
  return gemm_tn_splits(n, Pout, Pin, 64);
}
LayerWs layer_ws(int64_t n, int in, int out) {
  LayerWs w{};
  w.Pin = rup(2 * in, 64);
  w.Pout = rup(2 * out, 64);
  const int Pmax = w.Pin > w.Pout ? w.Pin : w.Pout;
  w.S = layer_splits(FAM_4M, n, w.Pout, w.Pin);
  // weight-gradient slabs: room for every family, for the complex sizing and for the real layers' own
  // (narrower rows split finer, so neither sizing dominates the other)
  int64_t slab_f = 0, bslab_f = 0;
  auto fit = [&](int Pm, int Pn) {
    for (int fam = FAM_4M; fam <= FAM_X3; ++fam) {
      int64_t S = fam == FAM_X3 ? gemmx3_tn_splits_max(n, Pm, Pn, 256) : layer_splits(fam, n, Pm, Pn);
      if (fam == FAM_X3 && gemmx2_tn_splits(n, Pm, Pn, 256) > S) S = gemmx2_tn_splits(n, Pm, Pn, 256);
      const int64_t sl = fam == FAM_3M ? S * 3 * (Pm / 2) * (Pn / 2) : S * Pm * Pn;
      if (sl > slab_f) slab_f = sl;
      if (S * Pm > bslab_f) bslab_f = S * Pm;
    }
  };
  fit(w.Pout, w.Pin);
  fit(rup(out, 64), rup(in, 64));
  int64_t off = 0;
  auto take = [&](int64_t cnt) { int64_t o = off; off += rup64(cnt, 64); return o; };
  w.xb = take(n * w.Pin);
  w.lin = take(n * w.Pout);
  w.out = take(n * w.Pout);
  w.gact = take(n * w.Pout);
  w.glin = take(n * w.Pout);
  w.gxb = take(n * w.Pin);
  w.gu = take(n * (w.Pout / 2));
  w.btf = take((int64_t)w.Pout * w.Pin);
  w.btd = take((int64_t)w.Pout * w.Pin);
  w.bias = take(w.Pout);
  w.slab = take(slab_f);
  w.bslab = take(bslab_f);
  w.fpw = take((int64_t)(final_bwd_blocks(n) + 32) * 8 * Pmax);
  w.fpb = take((int64_t)(final_bwd_blocks(n) + 32) * 8);
  w.crp = take((int64_t)(colreduce_blocks(n) + 32) * (w.Pout / 2) * 5);
  w.wf = take((int64_t)8 * Pmax);
  w.bfr = take(64);
  w.btf_x3 = take(gemmx3_b_image_floats(w.Pout, w.Pin));
  w.btd_x3 = take(gemmx3_b_image_floats(w.Pin, w.Pout));
  w.x2 = take(x2_region_floats(w.Pout, w.Pin));
  w.total = off;
  return w;
}

// One layer's GEMMs on the selected family.  All three take blocked rows (wire_dev.h) and leave the same
// outputs; they differ in the weight image (pack_*) and in the slab format of the weight gradient.
struct LayerGemm {
  int fam;
  float* x2ws;       // x2_region_floats(...) of the workspace, or null: never the 2 x fp16 kernels
  bool use_x2(int epi, int64_t n) const {
    return fam == FAM_X3 && x2ws && wire_split_f16_() && gemmx3_nt_is_h16(epi, n);
  }
  unsigned* slots(int which) const { return reinterpret_cast<unsigned*>(x2ws) + which * WIRE_AMAX_SLOTS; }
  // weights: W_ + btf / btd hold the fp32 images of the family, btf_x3 / btd_x3 the split images
  hipError_t pack(hipStream_t s, int kind, const float* W, const float* b, const float* V, const float* c, int out,
                  int in, int Pout_g, int Pin, float* btf, float* btd, float* bias, float* btf_x3,
                  float* btd_x3) const {
    if (fam == FAM_3M) return launch_pack3m(s, W, b, out, in, Pout_g / 2, Pin / 2, btf, btd, bias);
    hipError_t e = launch_pack_hidden(s, kind, W, b, V, c, out, in, kind == NK_WIRE2D ? Pout_g / 2 : Pout_g, Pin,
                                      btf, btd, bias);
    if (e != hipSuccess || fam != FAM_X3) return e;
    e = launch_x3_split_b(s, btf, Pin, Pout_g, Pin, btf_x3);
    if (e != hipSuccess) return e;
    e = launch_x3_split_b(s, btd, Pout_g, Pin, Pout_g, btd_x3);
    if (e != hipSuccess || !x2ws || !wire_split_f16_()) return e;
    // 2 x fp16 images: max |weight| (the transposed image holds the same values), then both scaled splits
    e = hipMemsetAsync(x2ws, 0, 4 * WIRE_AMAX_SLOTS * sizeof(float), s);
    if (e != hipSuccess) return e;
    e = launch_amax(s, btf, (int64_t)Pout_g * Pin, slots(1));
    if (e != hipSuccess) return e;
    float* img_f = x2ws + 4 * WIRE_AMAX_SLOTS;
    float* img_d = img_f + gemmx2_b_image_floats(Pout_g, Pin);
    X2SplitBatch xf{}, xd{};
    xf.src[0] = btf; xf.dst[0] = img_f; xf.slots[0] = slots(1);
    xd.src[0] = btd; xd.dst[0] = img_d; xd.slots[0] = slots(1);
    e = launch_x2_split_b_batch(s, xf, 1, Pin, Pout_g, Pin);
    if (e != hipSuccess) return e;
    return launch_x2_split_b_batch(s, xd, 1, Pout_g, Pin, Pout_g);
  }
  // C[n][Nc] = A[n][Kd] * image^T + epilogue; dgrad = true uses the transposed-conjugate image
  hipError_t nt(hipStream_t s, int epi, const float* A, int64_t n, int Nc, int Kd, bool dgrad, const float* btf,
                const float* btd, const float* btf_x3, const float* btd_x3, const GemmEpiParams& ep) const {
    if (use_x2(epi, n)) {
      // forward: image [Nc = Pout_g][Kd = Pin]; data gradient: the transposed image [Nc = Pin][Kd = Pout_g]
      const int Pout_g = dgrad ? Kd : Nc, Pin = dgrad ? Nc : Kd;
      float* img_f = x2ws + 4 * WIRE_AMAX_SLOTS;
      float* img_d = img_f + gemmx2_b_image_floats(Pout_g, Pin);
      unsigned* sa = slots(dgrad ? 2 : 0);
      hipError_t e = hipMemsetAsync(sa, 0, WIRE_AMAX_SLOTS * sizeof(unsigned), s);
      if (e != hipSuccess) return e;
      e = launch_amax(s, A, n * Kd, sa);
      if (e != hipSuccess) return e;
      GemmEpiParams e2 = ep;
      e2.amax_a = sa; e2.amax_b = slots(1);
      return launch_gemmx2h_nt(s, epi, A, Kd, dgrad ? img_d : img_f, n, Nc, Kd, e2);
    }
    if (fam == FAM_X3) return launch_gemmx3_nt(s, epi, A, Kd, dgrad ? btd_x3 : btf_x3, n, Nc, Kd, ep);
    if (fam == FAM_3M) return launch_gemm3m_nt(s, epi, A, Kd, dgrad ? btd : btf, Kd, n, Nc / 2, Kd / 2, ep);
    return launch_gemm_nt(s, epi, A, Kd, dgrad ? btd : btf, Kd, n, Nc, Kd, ep);
  }
  hipError_t tn_reduce(hipStream_t s, int kind, const float* G, const float* Z, int64_t n, int Pm, int Pn, int out,
                       int in, float* slab, float* bslab, float* gW, float* gb, float* gV, float* gc) const {
    hipError_t e;
    if (use_x2(EPI_STORE, n) && gemmx2_tn_applies(Pm, Pn)) {
      const int S2 = gemmx2_tn_splits(n, Pm, Pn, 256);
      e = hipMemsetAsync(slots(2), 0, 2 * WIRE_AMAX_SLOTS * sizeof(unsigned), s);       // G and Z
      if (e != hipSuccess) return e;
      e = launch_amax(s, G, n * Pm, slots(2));
      if (e != hipSuccess) return e;
      e = launch_amax(s, Z, n * Pn, slots(3));
      if (e != hipSuccess) return e;
      e = launch_gemmx2_tn(s, G, Pm, Z, Pn, n, Pm, Pn, S2, slab, bslab, slots(2), slots(3));
      if (e != hipSuccess) return e;
      return launch_wgrad_reduce(s, kind, slab, bslab, S2, out, in, Pm, Pn, gW, gb, gV, gc);
    }
    const int S = layer_splits(fam, n, Pm, Pn);
    if (fam == FAM_3M) {
      e = launch_gemm3m_tn(s, G, Pm, Z, Pn, n, Pm / 2, Pn / 2, S, slab, bslab);
      if (e != hipSuccess) return e;
      return launch_wgrad3m_reduce(s, slab, bslab, S, out, in, Pm / 2, Pn / 2, gW, gb);
    }
    e = fam == FAM_X3 ? launch_gemmx3_tn(s, G, Pm, Z, Pn, n, Pm, Pn, S, slab, bslab)
                      : launch_gemm_tn(s, G, Pm, Z, Pn, n, Pm, Pn, S, slab, bslab);
    if (e != hipSuccess) return e;
    return launch_wgrad_reduce(s, kind, slab, bslab, S, out, in, Pm, Pn, gW, gb, gV, gc);
  }
};
}  // namespace

#define LCHK(expr)                                                   \
  do {                                                               \
    hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) return wire_fail_(WIRE_ERR_HIP, hipGetErrorString(e_)); \
  } while (0)

extern "C" int64_t wire_layer_ws_bytes(int64_t n, int in_features, int out_features) {
  if (n < 0 || in_features < 1 || out_features < 1) return wire_fail_(WIRE_ERR_ARG, "bad layer shape");
  return layer_ws(n, in_features, out_features).total * 4 + 256;
}

// forward into the workspace (lin, out blocked).  Shared by fwd and bwd.
static int layer_forward_ws(hipStream_t s, const LayerWs& w, float* W_, const void* x, const void* Wt,
                            const void* b, float omega0, float scale0, int64_t n, int in, int out,
                            int is_first, bool want_lin = false) {
  if (is_first) {
    if (in > 4) return wire_fail_(WIRE_ERR_ARG, "is_first layers support in_features <= 4");
    // u = x W0^T + b0 is real: [n][Pout / 2] rows in W_ + w.lin when asked for
    LCHK(launch_first_fwd(s, NK_WIRE, (const float*)x, n, in, (const float*)Wt, (const float*)b,
                          nullptr, nullptr, out, w.Pout, omega0, scale0, want_lin ? W_ + w.lin : nullptr,
                          W_ + w.out));
    return WIRE_OK;
  }
  const LayerGemm g{wire_family_(WIRE_KIND_WIRE), W_ + w.x2};
  LCHK(launch_c64_to_blocked(s, (const float*)x, n, in, w.Pin, W_ + w.xb));
  LCHK(g.pack(s, NK_WIRE, (const float*)Wt, (const float*)b, nullptr, nullptr, out, in, w.Pout, w.Pin, W_ + w.btf,
              W_ + w.btd, W_ + w.bias, W_ + w.btf_x3, W_ + w.btd_x3));
  GemmEpiParams ep;
  ep.bias = W_ + w.bias; ep.o0 = W_ + w.lin; ep.o1 = W_ + w.out; ep.ld0 = w.Pout; ep.ld1 = w.Pout;
  ep.omega = omega0; ep.scale = scale0; ep.kvalid = out;
  LCHK(g.nt(s, EPI_GABOR_FWD, W_ + w.xb, n, w.Pout, w.Pin, false, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3,
            W_ + w.btd_x3, ep));
  return WIRE_OK;
}

extern "C" int wire_gabor_fwd(void* stream, const void* x, const void* W, const void* b,
                              float omega0, float scale0, int64_t n, int in_features,
                              int out_features, int is_first, void* lin_out, void* act_out,
                              void* ws, int64_t ws_bytes) {
  if (n < 0 || in_features < 1 || out_features < 1 || !x || !W || !b || !act_out || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor_fwd");
  if (n == 0) return WIRE_OK;
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer_forward_ws(s, w, W_, x, W, b, omega0, scale0, n, in_features, out_features, is_first,
                                lin_out != nullptr))
    return rc;
  LCHK(launch_blocked_to_c64(s, W_ + w.out, n, out_features, w.Pout, (float*)act_out));
  if (lin_out) {
    // the pre-activation `lin` of modules/wire.py:89: complex64 [n][out], or float32 [n][out] for is_first
    if (is_first) LCHK(launch_unpad_rows(s, W_ + w.lin, n, out_features, w.Pout / 2, (float*)lin_out));
    else LCHK(launch_blocked_to_c64(s, W_ + w.lin, n, out_features, w.Pout, (float*)lin_out));
  }
  return WIRE_OK;
}

extern "C" int wire_gabor_bwd(void* stream, const void* g_act, const void* x, const void* W,
                              const void* b, float omega0, float scale0, int64_t n,
                              int in_features, int out_features, int is_first, void* g_x,
                              void* g_W, void* g_b, void* ws, int64_t ws_bytes) {
  if (n <= 0 || in_features < 1 || out_features < 1 || !g_act || !x || !W || !b || !g_W || !g_b || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor_bwd");
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer_forward_ws(s, w, W_, x, W, b, omega0, scale0, n, in_features, out_features, is_first))
    return rc;
  LCHK(launch_c64_to_blocked(s, (const float*)g_act, n, out_features, w.Pout, W_ + w.gact));
  if (is_first) {
    const int ldu = w.Pout / 2;
    LCHK(launch_gabor_bwd_first_point(s, W_ + w.gact, W_ + w.out, (const float*)x, in_features,
                                      (const float*)W, (const float*)b, n, out_features, w.Pout,
                                      omega0, scale0, W_ + w.gu, ldu));
    LCHK(launch_colreduce(s, W_ + w.gu, ldu, out_features, (const float*)x, in_features, n,
                          W_ + w.crp, (float*)g_W, (float*)g_b));
    return WIRE_OK;
  }
  LCHK(launch_gabor_bwd_point(s, W_ + w.gact, W_ + w.lin, W_ + w.out, n, w.Pout, omega0, scale0,
                              W_ + w.glin));
  const LayerGemm g{wire_family_(WIRE_KIND_WIRE), W_ + w.x2};
  if (g_x) {
    GemmEpiParams ep; ep.o0 = W_ + w.gxb; ep.ld0 = w.Pin; ep.ld1 = w.Pin;
    LCHK(g.nt(s, EPI_STORE, W_ + w.glin, n, w.Pin, w.Pout, true, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3,
              W_ + w.btd_x3, ep));
    LCHK(launch_blocked_to_c64(s, W_ + w.gxb, n, in_features, w.Pin, (float*)g_x));
  }
  LCHK(g.tn_reduce(s, NK_WIRE, W_ + w.glin, W_ + w.xb, n, w.Pout, w.Pin, out_features, in_features, W_ + w.slab,
                   W_ + w.bslab, (float*)g_W, (float*)g_b, nullptr, nullptr));
  return WIRE_OK;
}

// trainable omega_0 / scale_0 (ComplexGaborLayer(trainable=True), modules/wire.py:80-81): out2 = {dL/d omega_0,
// dL/d scale_0} (device).  Recomputes the layer's forward from x, like wire_gabor_bwd.
extern "C" int wire_gabor_hparam_grad(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                                      float omega0, float scale0, int64_t n, int in_features, int out_features,
                                      int is_first, float* out2, void* ws, int64_t ws_bytes) {
  if (n <= 0 || in_features < 1 || out_features < 1 || !g_act || !x || !W || !b || !out2 || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor_hparam_grad");
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer_forward_ws(s, w, W_, x, W, b, omega0, scale0, n, in_features, out_features, is_first, true))
    return rc;
  LCHK(launch_c64_to_blocked(s, (const float*)g_act, n, out_features, w.Pout, W_ + w.gact));
  // partial sums live in the (unused here) g_lin region: 2 * ceil(n / 32) floats <= n * Pout
  LCHK(launch_gabor_hparam_grad(s, W_ + w.gact, W_ + w.lin, W_ + w.out, n, out_features, w.Pout, is_first, scale0,
                                W_ + w.glin, out2));
  return WIRE_OK;
}

extern "C" int wire_final_fwd(void* stream, const void* z, const void* Wf, const void* bf, int64_t n,
                              int in_features, int out_features, float* y, void* ws,
                              int64_t ws_bytes) {
  if (n < 0 || in_features < 1 || out_features < 1 || out_features > 8 || !z || !Wf || !bf || !y || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_final_fwd");
  if (n == 0) return WIRE_OK;
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  LCHK(launch_c64_to_blocked(s, (const float*)z, n, in_features, w.Pin, W_ + w.xb));
  LCHK(launch_pack_final(s, NK_WIRE, (const float*)Wf, (const float*)bf, in_features, w.Pin,
                         out_features, W_ + w.wf, W_ + w.bfr));
  LCHK(launch_final_fwd(s, W_ + w.xb, n, w.Pin, out_features, W_ + w.wf, W_ + w.bfr, y));
  return WIRE_OK;
}

extern "C" int wire_final_bwd(void* stream, const float* g_y, const void* z, const void* Wf,
                              int64_t n, int in_features, int out_features, void* g_z, void* g_Wf,
                              void* g_bf, void* ws, int64_t ws_bytes) {
  if (n <= 0 || in_features < 1 || out_features < 1 || out_features > 8 || !g_y || !z || !Wf ||
      !g_Wf || !g_bf || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_final_bwd");
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  LCHK(launch_c64_to_blocked(s, (const float*)z, n, in_features, w.Pin, W_ + w.xb));
  // bias pointer is irrelevant for the gradient; reuse Wf so the kernel reads valid memory
  LCHK(launch_pack_final(s, NK_WIRE, (const float*)Wf, (const float*)Wf, in_features, w.Pin,
                         out_features, W_ + w.wf, W_ + w.bfr));
  LCHK(launch_final_bwd(s, NK_WIRE, 1, g_y, n, out_features, W_ + w.wf, nullptr, W_ + w.xb,
                        in_features, w.Pin, 0.f, 0.f, W_ + w.gxb, W_ + w.fpw, W_ + w.fpb));
  LCHK(launch_final_reduce(s, NK_WIRE, W_ + w.fpw, W_ + w.fpb, final_bwd_blocks(n), out_features,
                           in_features, w.Pin, (float*)g_Wf, (float*)g_bf));
  if (g_z) LCHK(launch_blocked_to_c64(s, W_ + w.gxb, n, in_features, w.Pin, (float*)g_z));
  return WIRE_OK;
}

// ---------------------------------------------------------------------------
// real-valued layers: SineLayer / GaussLayer / ReLULayer .forward
// (modules/siren.py:48-49, gauss.py:27-28, relu.py:28-29) on native [n][in] f32 tensors.
// ---------------------------------------------------------------------------
static int real_epi_fwd(int kind) {
  return kind == WIRE_KIND_SIREN ? EPI_SIREN_FWD : kind == WIRE_KIND_GAUSS ? EPI_GAUSS_FWD : EPI_RELU_FWD;
}
static int real_forward_ws(hipStream_t s, const LayerWs& w, float* W_, int kind, const void* x, const void* Wt,
                           const void* b, float omega0, float scale0, int64_t n, int in, int out, int Pin,
                           int Pout) {
  const LayerGemm g{wire_family_(kind), W_ + w.x2};
  LCHK(launch_pad_rows(s, (const float*)x, n, in, Pin, W_ + w.xb));
  LCHK(g.pack(s, kind, (const float*)Wt, (const float*)b, nullptr, nullptr, out, in, Pout, Pin, W_ + w.btf,
              W_ + w.btd, W_ + w.bias, W_ + w.btf_x3, W_ + w.btd_x3));
  GemmEpiParams ep;
  ep.bias = W_ + w.bias; ep.o0 = W_ + w.lin; ep.o1 = W_ + w.out; ep.ld0 = Pout; ep.ld1 = Pout;
  ep.omega = omega0; ep.scale = scale0; ep.kvalid = out;
  LCHK(g.nt(s, real_epi_fwd(kind), W_ + w.xb, n, Pout, Pin, false, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3,
            W_ + w.btd_x3, ep));
  return WIRE_OK;
}

extern "C" int wire_real_layer_fwd(void* stream, int kind, const float* x, const float* W, const float* b,
                                   float omega0, float scale0, int64_t n, int in_features, int out_features,
                                   float* act_out, void* ws, int64_t ws_bytes) {
  if (kind < WIRE_KIND_SIREN || kind > WIRE_KIND_RELU || n < 0 || in_features < 1 || out_features < 1 || !x ||
      !W || !b || !act_out || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_real_layer_fwd");
  if (n == 0) return WIRE_OK;
  const LayerWs w = layer_ws(n, in_features, out_features);   // complex sizing is an upper bound
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  const int Pin = rup(in_features, 64), Pout = rup(out_features, 64);
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = real_forward_ws(s, w, W_, kind, x, W, b, omega0, scale0, n, in_features, out_features, Pin, Pout))
    return rc;
  LCHK(launch_unpad_rows(s, W_ + w.out, n, out_features, Pout, act_out));
  return WIRE_OK;
}

extern "C" int wire_real_layer_bwd(void* stream, int kind, const float* g_act, const float* x, const float* W,
                                   const float* b, float omega0, float scale0, int64_t n, int in_features,
                                   int out_features, float* g_x, float* g_W, float* g_b, void* ws,
                                   int64_t ws_bytes) {
  if (kind < WIRE_KIND_SIREN || kind > WIRE_KIND_RELU || n <= 0 || in_features < 1 || out_features < 1 ||
      !g_act || !x || !W || !b || !g_W || !g_b || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_real_layer_bwd");
  const LayerWs w = layer_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  const int Pin = rup(in_features, 64), Pout = rup(out_features, 64);
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = real_forward_ws(s, w, W_, kind, x, W, b, omega0, scale0, n, in_features, out_features, Pin, Pout))
    return rc;
  LCHK(launch_pad_rows(s, g_act, n, out_features, Pout, W_ + w.gact));
  LCHK(launch_real_act_bwd_point(s, kind, W_ + w.gact, W_ + w.lin, W_ + w.out, n, Pout, omega0, scale0,
                                 W_ + w.glin));
  const LayerGemm g{wire_family_(kind), W_ + w.x2};
  if (g_x) {
    GemmEpiParams ep; ep.o0 = W_ + w.gxb; ep.ld0 = Pin; ep.ld1 = Pin;
    LCHK(g.nt(s, EPI_STORE, W_ + w.glin, n, Pin, Pout, true, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3, W_ + w.btd_x3,
              ep));
    LCHK(launch_unpad_rows(s, W_ + w.gxb, n, in_features, Pin, g_x));
  }
  LCHK(g.tn_reduce(s, kind, W_ + w.glin, W_ + w.xb, n, Pout, Pin, out_features, in_features, W_ + w.slab,
                   W_ + w.bslab, g_W, g_b, nullptr, nullptr));
  return WIRE_OK;
}

// ---------------------------------------------------------------------------
// ComplexGaborLayer2D (modules/wire2d.py:21-67) on native tensors: both Linears of the layer are ONE GEMM
// ([n, Pin] x [Pin -> 2 Pout], 128-column groups (lin_re | lin_im | sy_re | sy_im)).
// ---------------------------------------------------------------------------
namespace {
struct Layer2dWs {
  int Pin, Pout, S, ldu;
  int64_t xb, linsy, out, gact, glinsy, gxb, gup, btf, btd, bias, slab, bslab, crp, btf_x3, btd_x3, x2, total;
};
Layer2dWs layer2d_ws(int64_t n, int in, int out) {
  Layer2dWs w{};
  w.Pin = rup(2 * in, 64);
  w.Pout = rup(2 * out, 64);
  w.ldu = w.Pout / 2;
  w.S = gemm_tn_splits(n, 2 * w.Pout, w.Pin, 64);
  const int64_t s_max = std::max<int64_t>(w.S, gemmx3_tn_splits_max(n, 2 * w.Pout, w.Pin, 256));
  int64_t off = 0;
  auto take = [&](int64_t cnt) { int64_t o = off; off += rup64(cnt, 64); return o; };
  w.xb = take(n * w.Pin);
  w.linsy = take(n * 2 * w.Pout);
  w.out = take(n * w.Pout);
  w.gact = take(n * w.Pout);
  w.glinsy = take(n * 2 * w.Pout);
  w.gxb = take(n * w.Pin);
  w.gup = take(n * 2 * w.ldu);
  w.btf = take((int64_t)2 * w.Pout * w.Pin);
  w.btd = take((int64_t)2 * w.Pout * w.Pin);
  w.bias = take(2 * w.Pout);
  w.slab = take(s_max * 2 * w.Pout * w.Pin);
  w.bslab = take(s_max * 2 * w.Pout);
  w.crp = take((int64_t)(colreduce_blocks(n) + 32) * w.ldu * 5);
  w.btf_x3 = take(gemmx3_b_image_floats(2 * w.Pout, w.Pin));
  w.btd_x3 = take(gemmx3_b_image_floats(w.Pin, 2 * w.Pout));
  w.x2 = take(x2_region_floats(2 * w.Pout, w.Pin));
  w.total = off;
  return w;
}

// forward into the workspace (linsy, out blocked); shared by fwd and bwd
int layer2d_forward_ws(hipStream_t s, const Layer2dWs& w, float* W_, const void* x, const void* Wt, const void* b,
                       const void* Vt, const void* c, float omega0, float scale0, int64_t n, int in, int out,
                       int is_first, bool want_lin = false) {
  if (is_first) {
    if (in > 4) return wire_fail_(WIRE_ERR_ARG, "is_first layers support in_features <= 4");
    LCHK(launch_first_fwd(s, NK_WIRE2D, (const float*)x, n, in, (const float*)Wt, (const float*)b,
                          (const float*)Vt, (const float*)c, out, w.Pout, omega0, scale0,
                          want_lin ? W_ + w.linsy : nullptr, W_ + w.out));
    return WIRE_OK;
  }
  const LayerGemm g{wire_family_(WIRE_KIND_WIRE2D), W_ + w.x2};
  LCHK(launch_c64_to_blocked(s, (const float*)x, n, in, w.Pin, W_ + w.xb));
  LCHK(g.pack(s, NK_WIRE2D, (const float*)Wt, (const float*)b, (const float*)Vt, (const float*)c, out, in,
              2 * w.Pout, w.Pin, W_ + w.btf, W_ + w.btd, W_ + w.bias, W_ + w.btf_x3, W_ + w.btd_x3));
  GemmEpiParams ep;
  ep.bias = W_ + w.bias; ep.o0 = W_ + w.linsy; ep.o1 = W_ + w.out; ep.ld0 = 2 * w.Pout; ep.ld1 = w.Pout;
  ep.omega = omega0; ep.scale = scale0; ep.kvalid = out;
  LCHK(g.nt(s, EPI_GABOR2D_FWD, W_ + w.xb, n, 2 * w.Pout, w.Pin, false, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3,
            W_ + w.btd_x3, ep));
  return WIRE_OK;
}
}  // namespace

extern "C" int64_t wire_layer2d_ws_bytes(int64_t n, int in_features, int out_features) {
  if (n < 0 || in_features < 1 || out_features < 1) return wire_fail_(WIRE_ERR_ARG, "bad layer shape");
  return layer2d_ws(n, in_features, out_features).total * 4 + 256;
}

extern "C" int wire_gabor2d_fwd(void* stream, const void* x, const void* W, const void* b, const void* V,
                                const void* c, float omega0, float scale0, int64_t n, int in_features,
                                int out_features, int is_first, void* act_out, void* ws, int64_t ws_bytes) {
  if (n < 0 || in_features < 1 || out_features < 1 || !x || !W || !b || !V || !c || !act_out || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor2d_fwd");
  if (n == 0) return WIRE_OK;
  const Layer2dWs w = layer2d_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer2d_forward_ws(s, w, W_, x, W, b, V, c, omega0, scale0, n, in_features, out_features, is_first))
    return rc;
  LCHK(launch_blocked_to_c64(s, W_ + w.out, n, out_features, w.Pout, (float*)act_out));
  return WIRE_OK;
}

// trainable omega_0 / scale_0 of ComplexGaborLayer2D (modules/wire2d.py:42-43): out2 = {dL/d omega_0, dL/d scale_0}
extern "C" int wire_gabor2d_hparam_grad(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                                        const void* V, const void* c, float omega0, float scale0, int64_t n,
                                        int in_features, int out_features, int is_first, float* out2, void* ws,
                                        int64_t ws_bytes) {
  if (n <= 0 || in_features < 1 || out_features < 1 || !g_act || !x || !W || !b || !V || !c || !out2 || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor2d_hparam_grad");
  const Layer2dWs w = layer2d_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer2d_forward_ws(s, w, W_, x, W, b, V, c, omega0, scale0, n, in_features, out_features, is_first, true))
    return rc;
  LCHK(launch_c64_to_blocked(s, (const float*)g_act, n, out_features, w.Pout, W_ + w.gact));
  LCHK(launch_gabor2d_hparam_grad(s, W_ + w.gact, W_ + w.linsy, W_ + w.out, n, out_features, w.Pout, is_first, scale0,
                                  W_ + w.glinsy, out2));
  return WIRE_OK;
}

extern "C" int wire_gabor2d_bwd(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                                const void* V, const void* c, float omega0, float scale0, int64_t n,
                                int in_features, int out_features, int is_first, void* g_x, void* g_W, void* g_b,
                                void* g_V, void* g_c, void* ws, int64_t ws_bytes) {
  if (n <= 0 || in_features < 1 || out_features < 1 || !g_act || !x || !W || !b || !V || !c || !g_W || !g_b ||
      !g_V || !g_c || !ws)
    return wire_fail_(WIRE_ERR_ARG, "bad argument to wire_gabor2d_bwd");
  const Layer2dWs w = layer2d_ws(n, in_features, out_features);
  if (ws_bytes < w.total * 4) return wire_fail_(WIRE_ERR_SIZE, "layer workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* W_ = (float*)ws;
  if (int rc = layer2d_forward_ws(s, w, W_, x, W, b, V, c, omega0, scale0, n, in_features, out_features, is_first))
    return rc;
  LCHK(launch_c64_to_blocked(s, (const float*)g_act, n, out_features, w.Pout, W_ + w.gact));
  if (is_first) {
    LCHK(launch_gabor2d_bwd_first_point(s, W_ + w.gact, W_ + w.out, (const float*)x, in_features,
                                        (const float*)W, (const float*)b, (const float*)V, (const float*)c, n,
                                        out_features, w.Pout, omega0, scale0, W_ + w.gup, w.ldu));
    LCHK(launch_colreduce(s, W_ + w.gup, 2 * w.ldu, out_features, (const float*)x, in_features, n, W_ + w.crp,
                          (float*)g_W, (float*)g_b));
    LCHK(launch_colreduce(s, W_ + w.gup + w.ldu, 2 * w.ldu, out_features, (const float*)x, in_features, n,
                          W_ + w.crp, (float*)g_V, (float*)g_c));
    return WIRE_OK;
  }
  LCHK(launch_gabor2d_bwd_point(s, W_ + w.gact, W_ + w.linsy, W_ + w.out, n, w.Pout, omega0, scale0,
                                W_ + w.glinsy));
  const LayerGemm g{wire_family_(WIRE_KIND_WIRE2D), W_ + w.x2};
  if (g_x) {
    GemmEpiParams ep; ep.o0 = W_ + w.gxb; ep.ld0 = w.Pin; ep.ld1 = w.Pin;
    LCHK(g.nt(s, EPI_STORE, W_ + w.glinsy, n, w.Pin, 2 * w.Pout, true, W_ + w.btf, W_ + w.btd, W_ + w.btf_x3,
              W_ + w.btd_x3, ep));
    LCHK(launch_blocked_to_c64(s, W_ + w.gxb, n, in_features, w.Pin, (float*)g_x));
  }
  LCHK(g.tn_reduce(s, NK_WIRE2D, W_ + w.glinsy, W_ + w.xb, n, 2 * w.Pout, w.Pin, out_features, in_features,
                   W_ + w.slab, W_ + w.bslab, (float*)g_W, (float*)g_b, (float*)g_V, (float*)g_c));
  return WIRE_OK;
}
