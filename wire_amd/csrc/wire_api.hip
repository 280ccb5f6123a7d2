// wire_api.hip -- the C ABI of libwire_hip.so (include/wire_hip.h): argument
// checking, buffer carving and the launch sequences of the WIRE hot path.
// No device memory is allocated here; every launch goes on the caller's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wire_hip.h"
#include "wire_gemm.h"
#include "wire_point.h"

#ifndef WIRE_AMAX_SLOTS
#define WIRE_AMAX_SLOTS 64            // wire_dev.h (device header): sharded max-|value| slots per operand tensor
#endif

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(WIRE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

int wire_fail_(int code, const char* msg) { return fail(code, "%s", msg); }   // wire_layer_api.hip

extern "C" int wire_abi_version(void) { return WIRE_ABI_VERSION; }
extern "C" const char* wire_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------
// profiling hooks
// ---------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::vector<ProfRec> g_prof_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;

struct ProfScope {
  hipStream_t s; int cls; double flops; bool on; hipEvent_t a{}, b{};
  ProfScope(hipStream_t s_, int cls_, double flops_) : s(s_), cls(cls_), flops(flops_), on(false) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;   // profiling off: no lock on the launch path
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    on = true;
    if (!g_prof_pool.empty()) {
      a = g_prof_pool.back().first; b = g_prof_pool.back().second; g_prof_pool.pop_back();
    } else {
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    }
    (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back({a, b, cls, flops});
  }
};
}  // namespace

extern "C" int wire_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on.store(on != 0);
  return WIRE_OK;
}
extern "C" int wire_prof_read(double* ms_total, int64_t* launches, double* flops_total) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < WIRE_PROF_CLASSES; ++i) { ms_total[i] = 0; launches[i] = 0; flops_total[i] = 0; }
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      ms_total[r.cls] += ms; launches[r.cls] += 1; flops_total[r.cls] += r.flops;
    }
    g_prof_pool.emplace_back(r.a, r.b);
  }
  g_prof_recs.clear();
  return WIRE_OK;
}

// ---------------------------------------------------------------------------
// tuning knobs
// ---------------------------------------------------------------------------
// wire training step on the 16 x 16 x 32 kernels: backward passes evaluate out = act(lin) again instead of reading it
// (the data-gradient epilogues and the fused final stage), and the last hidden layer does not store out at all
static int env_int_(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// (relaxed atomics: wire_tune_set may run on one thread while autograd's thread reads a knob inside a launch sequence)
static std::atomic<int> g_recompute_out{env_int_("WIRE_RECOMPUTE_OUT", 1)};
static std::atomic<int> g_first_sums{env_int_("WIRE_FIRST_SUMS", 1)};
static std::atomic<int> g_complex_3m{1};   // wire: 3-multiplication complex GEMMs on the fp32 MFMA (wire_gemm3m.hip)
static int env_flag(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? (atoi(v) != 0) : dflt;
}
// every net: split-bf16 GEMMs on the bf16 MFMA (wire_gemmx3.hip); overrides complex_3m
static std::atomic<int> g_split_bf16{env_flag("WIRE_SPLIT_BF16", 1)};
// hidden-layer GEMMs of batches >= 4096 rows as a 2 x fp16 split on the f16 MFMA (wire_gemmx2h.hip): 3 instead of 6
// matrix-core products per fp32 product, operand scales from device-side maxima; needs split_bf16 and the 16 x 16 x 32
// kernels (x3_h16) for the net kind, falls back to the 3 x bf16 kernels otherwise
static std::atomic<int> g_split_f16{env_flag("WIRE_SPLIT_F16", 1)};
// with split_f16: activations with an a-priori bound (Gabor, sine, Gaussian outputs) are stored ALREADY SPLIT into fp16
// pairs by the epilogue that produces them (wire_dev.h: wire_store_out4), so the GEMMs that read them -- the next layer's
// forward, the weight gradient -- spend no vector instructions on the split (out_split_scale below)
static std::atomic<int> g_split_out{env_flag("WIRE_SPLIT_OUT", 1)};
// with the fused training forward + data-gradient chain of a sine / Gaussian net: store r = c lin and no out_l ("fused_rstore")
static std::atomic<int> g_fused_rstore{env_flag("WIRE_FUSED_RSTORE", 1)};
// behind the data-gradient chain: the weight gradients of layers 2 .. L as one launch ("wgrad_batch" / WIRE_WGRAD_BATCH)
static std::atomic<int> g_wgrad_batch{env_flag("WIRE_WGRAD_BATCH", 1)};
// family the flags select for a net kind (wire_layer_api.hip): 2 split-bf16, 1 complex 3M (wire only), 0 4M
int wire_family_(int kind) {
  if (g_split_bf16) return 2;
  return (kind == WIRE_KIND_WIRE && g_complex_3m) ? 1 : 0;
}
int wire_split_f16_() { return g_split_f16; }   // wire_layer_api.hip
extern "C" int wire_tune_get(const char* key) {
  if (!key) return fail(WIRE_ERR_ARG, "null key");
  if (!strcmp(key, "complex_3m")) return g_complex_3m;
  if (!strcmp(key, "split_bf16")) return g_split_bf16;
  if (!strcmp(key, "split_out")) return g_split_out;
  if (!strcmp(key, "fused_rstore")) return g_fused_rstore;
  if (!strcmp(key, "wgrad_batch")) return g_wgrad_batch;
  if (!strcmp(key, "x3_h16")) return gemmx3h_mode();
  if (!strcmp(key, "x3_tn16")) return gemmx3_tn16_mode();
  if (!strcmp(key, "recompute_out")) return g_recompute_out;
  if (!strcmp(key, "split_f16")) return g_split_f16;
  if (gemmx2h_tune_get(key) >= 0) return gemmx2h_tune_get(key);
  if (fused_tune_get(key) >= 0) return fused_tune_get(key);
  return fail(WIRE_ERR_ARG, "unknown tuning key: %s", key);
}
extern "C" int wire_tune_set(const char* key, int value) {
  if (!key) return fail(WIRE_ERR_ARG, "null key");
  if (!strcmp(key, "complex_3m")) { g_complex_3m = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "split_bf16")) { g_split_bf16 = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "recompute_out")) { g_recompute_out = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "first_sums")) { g_first_sums = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "split_f16")) { g_split_f16 = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "split_out")) { g_split_out = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "fused_rstore")) { g_fused_rstore = value ? 1 : 0; return WIRE_OK; }
  if (!strcmp(key, "wgrad_batch")) { g_wgrad_batch = value ? 1 : 0; return WIRE_OK; }
  if (gemmx2h_tune_set(key, value) == 0) return WIRE_OK;
  if (fused_tune_set(key, value) == 0) return WIRE_OK;
  if (gemm_tune_set(key, value) == 0) return WIRE_OK;
  if (gemmx3_tune_set(key, value) == 0) return WIRE_OK;
  return fail(WIRE_ERR_ARG, "unknown tuning key or bad value: %s=%d", key, value);
}

// ---------------------------------------------------------------------------
// network plan
// ---------------------------------------------------------------------------
namespace {
inline int rup(int v, int m) { return (v + m - 1) / m * m; }

struct Plan {
  int kind, D, K, L, O, F;
  float w1, w, s;
  bool cplx, first_gemm, m3, x3, x2;
  int P, Pl, Din, Pin0, ldu, ntens, per_layer, Kp;
  // packed image offsets (floats); index l = 0..L (l = 0 only when first_gemm)
  std::vector<int64_t> off_fwd, off_dg, off_bias, off_fwd_x3, off_dg_x3, off_fwd_3m, off_dg_3m, off_fwd_x2, off_dg_x2;
  int64_t off_wf, off_bf, off_first, off_wamax, total_packed;   // off_wamax: max-|weight| slots, WIRE_AMAX_SLOTS per layer
  int64_t off_fx;    // k-permuted 2 x fp16 images of the hidden layers for the fused forward (wire_fused.hip), -1 = no such shape
  bool k_split_out, k_recompute_out, k_first_sums, k_rstore, k_wgrad_batch, k_fused_fwd, k_fused_train, k_fused_bwd, k_fused_final;
  int64_t off_fxd;   // the same of the TRANSPOSED weights of layers L .. 1 (in that order) for the data-gradient chain, -1 = none
  std::vector<int64_t> tfloats;
};

int make_plan(const wire_net_desc* d, Plan& p) {
  if (!d) return fail(WIRE_ERR_ARG, "null descriptor");
  p.kind = d->kind; p.D = d->in_features; p.K = d->width; p.L = d->hidden_layers;
  p.O = d->out_features; p.F = d->posenc_freqs;
  p.w1 = d->first_omega0; p.w = d->hidden_omega0; p.s = d->scale0;
  if (p.kind < WIRE_KIND_WIRE || p.kind > WIRE_KIND_RELU) return fail(WIRE_ERR_ARG, "unknown kind %d", p.kind);
  if (p.D < 1 || p.D > 4) return fail(WIRE_ERR_ARG, "in_features %d outside 1..4", p.D);
  if (p.K < 1 || p.K > 4096) return fail(WIRE_ERR_ARG, "width %d outside 1..4096", p.K);
  if (p.L < 0 || p.L > 64) return fail(WIRE_ERR_ARG, "hidden_layers %d outside 0..64", p.L);
  if (p.O < 1 || p.O > 8) return fail(WIRE_ERR_ARG, "out_features %d outside 1..8", p.O);
  if (p.F < 0 || (p.F > 0 && p.kind != WIRE_KIND_RELU)) return fail(WIRE_ERR_ARG, "posenc only with relu");
  p.cplx = (p.kind == WIRE_KIND_WIRE || p.kind == WIRE_KIND_WIRE2D);
  p.P = p.cplx ? rup(2 * p.K, 64) : rup(p.K, 64);
  p.Pl = (p.kind == WIRE_KIND_WIRE2D) ? 2 * p.P : p.P;
  p.ldu = p.P / 2;
  p.Kp = p.P / 2;
  p.x3 = g_split_bf16 != 0;
  p.x2 = p.x3 && g_split_f16 != 0;
  p.m3 = (p.kind == WIRE_KIND_WIRE) && g_complex_3m && !p.x3;
  // one snapshot of the knobs that decide FORMATS (what the forward stores and the backward of the same call reads): a
  // wire_tune_set from another thread in the middle of wire_train_fwd_bwd must not split a call between two formats
  // (the forward and the backward of the autograd path are separate calls: knobs must not change between them)
  p.k_split_out = g_split_out != 0; p.k_recompute_out = g_recompute_out != 0; p.k_first_sums = g_first_sums != 0;
  p.k_rstore = g_fused_rstore != 0; p.k_wgrad_batch = g_wgrad_batch != 0;
  p.k_fused_fwd = fused_fwd_enabled(); p.k_fused_train = fused_train_enabled(); p.k_fused_bwd = fused_bwd_enabled();
  p.k_fused_final = fused_final_enabled();
  p.first_gemm = p.F > 0;
  p.Din = p.first_gemm ? p.D + 2 * p.D * p.F : p.D;
  p.Pin0 = p.first_gemm ? rup(p.Din, 64) : 0;
  p.per_layer = (p.kind == WIRE_KIND_WIRE2D) ? 4 : 2;
  p.ntens = p.per_layer * (p.L + 1) + 2;
  p.tfloats.assign(p.ntens, 0);
  const int64_t K = p.K, cm = p.cplx ? 2 : 1;
  for (int l = 0; l <= p.L; ++l) {
    const int64_t in = (l == 0) ? p.Din : K;
    const int64_t m = (l == 0) ? 1 : cm;   // first layer is real-valued
    for (int q = 0; q < p.per_layer; q += 2) {
      p.tfloats[p.per_layer * l + q] = m * K * in;
      p.tfloats[p.per_layer * l + q + 1] = m * K;
    }
  }
  p.tfloats[p.ntens - 2] = cm * p.O * K;
  p.tfloats[p.ntens - 1] = cm * p.O;
  // packed image
  int64_t off = 0;
  p.off_fwd.assign(p.L + 1, -1); p.off_dg.assign(p.L + 1, -1); p.off_bias.assign(p.L + 1, -1);
  p.off_fwd_x3.assign(p.L + 1, -1); p.off_dg_x3.assign(p.L + 1, -1);
  p.off_fwd_3m.assign(p.L + 1, -1); p.off_dg_3m.assign(p.L + 1, -1);
  p.off_fwd_x2.assign(p.L + 1, -1); p.off_dg_x2.assign(p.L + 1, -1);
  for (int l = p.first_gemm ? 0 : 1; l <= p.L; ++l) {
    const int64_t pin = (l == 0) ? p.Pin0 : p.P;
    // every family's image is ALWAYS written by wire_pack_params (real-expanded fp32, its split-bf16 form and,
    // for `wire`, the blocked complex planes of the 3M kernels), so a packed buffer stays valid whatever the
    // tuning flags are when it is used (e.g. changed between a forward and its backward)
    const int64_t img = (int64_t)p.Pl * pin;
    p.off_fwd[l] = off; off += img;
    p.off_dg[l] = off; off += img;
    p.off_bias[l] = off; off += p.Pl;
    p.off_fwd_x3[l] = off; off += gemmx3_b_image_floats(p.Pl, (int)pin);
    p.off_dg_x3[l] = off; off += gemmx3_b_image_floats((int)pin, p.Pl);
    if (p.kind == WIRE_KIND_WIRE && l >= 1) {
      p.off_fwd_3m[l] = off; off += (int64_t)p.Kp * p.P;
      p.off_dg_3m[l] = off; off += (int64_t)p.Kp * p.P;
    }
    if (l >= 1) {                                         // 2 x fp16 images of the hidden layers
      p.off_fwd_x2[l] = off; off += gemmx2_b_image_floats(p.Pl, (int)pin);
      p.off_dg_x2[l] = off; off += gemmx2_b_image_floats((int)pin, p.Pl);
    }
  }
  p.off_wamax = off; off += (int64_t)(p.L + 1) * WIRE_AMAX_SLOTS;
  p.off_fx = -1;
  // (a positional-encoding net -- relu, 64 padded encoded features -- has its GEMM first layer's image, P x 64, in front)
  const bool fx_ok = p.L >= 1 && fused_fwd_shape(p.kind, p.P) && (!p.first_gemm || (p.kind == WIRE_KIND_RELU && p.Pin0 == 64));
  if (fx_ok) { p.off_fx = off; off += (p.first_gemm ? (int64_t)p.P * p.Pin0 : 0) + (int64_t)p.L * fused_b_image_floats(p.P); }
  p.off_fxd = -1;
  if (fx_ok && p.L >= 1 && fused_bwd_shape(p.kind, p.P)) { p.off_fxd = off; off += (int64_t)p.L * fused_b_image_floats(p.P); }
  p.off_wf = off; off += (int64_t)p.O * p.P;
  p.off_bf = off; off += 64;
  p.off_first = off;   // native copies of the first layer's tensors (W0,b0[,V0,c0])
  if (!p.first_gemm) for (int q = 0; q < p.per_layer; ++q) off += rup((int)p.tfloats[q], 4);
  p.total_packed = off;
  return WIRE_OK;
}

inline int64_t fx_hidden_off(const Plan& p) { return p.off_fx + (p.first_gemm ? (int64_t)p.P * p.Pin0 : 0); }
inline int64_t first_native_off(const Plan& p, int q) {
  int64_t off = p.off_first;
  for (int i = 0; i < q; ++i) off += rup((int)p.tfloats[i], 4);
  return off;
}

// activation carve (floats)
struct ActLayout {
  int64_t pe, out0, lin0, lin1, total;   // out_l = out0 + l*np*P ; lin_l = lin1 + (l-1)*np*Pl
  int64_t ping, pong;                    // inference
  int64_t amax;                          // max |out_l| slots, WIRE_AMAX_SLOTS per layer l = 0..L (2 x fp16 GEMMs)
  int64_t np;                            // rows each saved buffer is spaced by: n rounded up to 128 -- the fused training
                                         // forward (wire_fused.hip) stores whole 128-row workgroup tiles unconditionally
};
ActLayout act_layout(const Plan& p, int64_t n, int save) {
  ActLayout a{};
  int64_t off = 0;
  a.np = (n + 127) / 128 * 128;
  a.amax = off; off += (int64_t)(p.L + 2) * WIRE_AMAX_SLOTS;
  a.pe = off; if (p.first_gemm) off += n * p.Pin0;
  if (save) {
    a.out0 = off; off += a.np * p.P * (p.L + 1);
    a.lin0 = off; if (!p.cplx) off += a.np * p.P;
    a.lin1 = off; off += a.np * p.Pl * p.L;
  } else {
    a.ping = off; off += n * p.P;
    a.pong = off; off += n * p.P;
  }
  a.total = off;
  return a;
}

struct ScratchLayout { int64_t ga, gb, gu, slab, bslab, fpw, fpb, crp, gamax, gch, gch_stride, total; int S; };
ScratchLayout scratch_layout(const Plan& p, int64_t n) {
  ScratchLayout s{};
  int64_t off = 0;
  s.gamax = off; off += (int64_t)(p.L + 2) * WIRE_AMAX_SLOTS;   // max |g_lin_l| slots (2 x fp16 GEMMs)
  s.ga = off; off += n * p.Pl;
  s.gb = off; off += n * p.Pl;
  // the data-gradient chain (wire_fused.hip: fused_bwd_kernel) keeps EVERY g_lin_l (l = 1 .. L; the weight-gradient GEMMs
  // run after it), rows padded to 128: g_lin_l at gch + l * gch_stride
  s.gch = -1; s.gch_stride = (n + 127) / 128 * 128 * p.Pl;
  if (p.off_fxd >= 0) { s.gch = off; off += (int64_t)(p.L + 1) * s.gch_stride; }
  s.gu = off; if (p.cplx) off += n * p.ldu * (p.kind == WIRE_KIND_WIRE2D ? 2 : 1);
  const int64_t pn = p.first_gemm && p.Pin0 > p.P ? p.Pin0 : p.P;
  // slabs sized for the largest split count of the three GEMM families (flag-independent scratch size)
  // (the split-bf16 kernel may split finer: narrow nets have few tiles and would otherwise leave CUs idle)
  const int s_x3k = gemmx3_tn_splits_max(n, p.Pl, (int)pn, 256), s_4m = gemm_tn_splits(n, p.Pl, (int)pn, 64);
  const int s_x2 = gemmx2_tn_splits(n, p.Pl, (int)pn, 256);      // 0 when the 2 x fp16 kernel has no shape for the widths
  const int s_x3 = s_x3k > s_x2 ? s_x3k : s_x2;                  // split-family slabs: room for either kernel's count
  const int s_max = s_x3 > s_4m ? s_x3 : s_4m;
  if (p.m3) {
    s.S = gemm3m_tn_splits(n, p.Kp, p.Kp, 64);
    int64_t need = (int64_t)s.S * 3 * p.Kp * p.Kp, full = (int64_t)s_max * p.Pl * pn;
    s.slab = off; off += need > full ? need : full;
    need = (int64_t)s.S * 2 * p.Kp; full = (int64_t)s_max * p.Pl;
    s.bslab = off; off += need > full ? need : full;
  } else {
    s.S = p.x3 ? s_x3 : s_4m;
    int64_t need3 = 0, needb3 = 0;
    if (p.kind == WIRE_KIND_WIRE) {
      const int s3 = gemm3m_tn_splits(n, p.Kp, p.Kp, 64);
      need3 = (int64_t)s3 * 3 * p.Kp * p.Kp; needb3 = (int64_t)s3 * 2 * p.Kp;
    }
    int64_t full = (int64_t)s_max * p.Pl * pn;
    s.slab = off; off += need3 > full ? need3 : full;
    full = (int64_t)s_max * p.Pl;
    s.bslab = off; off += needb3 > full ? needb3 : full;
  }
  // + pre-reduction scratch; the training forward with the final stage inside writes one block per 128-row workgroup
  const int nbf = (p.off_fx >= 0 && !p.cplx ? 2 : 1) * final_bwd_blocks(n) + 32;
  s.fpw = off; off += (int64_t)nbf * p.O * p.P;
  s.fpb = off; off += (int64_t)nbf * p.O + 64;
  // (wire2d: two sets, one per Linear of the first layer, when the data-gradient epilogue forms the sums itself)
  // (the data-gradient chain writes one block of first-layer sums per 64- or 128-row workgroup: 4 x the 256-row blocks)
  s.crp = off; off += (int64_t)((p.off_fxd >= 0 ? 4 : 1) * colreduce_blocks(n) + 32) * (p.cplx ? p.ldu : p.P) * 5 *
                      (p.kind == WIRE_KIND_WIRE2D ? 2 : 1);
  s.total = off;
  return s;
}

int epi_fwd(int kind);
int epi_bwd(int kind);
// the hidden-layer GEMMs of this call run as 2 x fp16 splits: every forward / data-gradient launch is then a 16 x 16 x 32
// kernel (M >= 4096, x3_h16 bits of the kind), whose epilogues track the maxima the next GEMM scales by
bool use_x2(const Plan& p, int64_t n) {
  return p.x2 && p.L >= 1 && gemmx3_nt_is_h16(epi_fwd(p.kind), n) && gemmx3_nt_is_h16(epi_bwd(p.kind), n);
}
// Scale with which out_l of this call is stored pre-split (wire_dev.h: wire_store_out4), 0 = plain fp32.  Pre-split when
//  * the call runs the 2 x fp16 kernels and every reader of out_l understands the format: the forward GEMM of layer l + 1
//    (pre-split A edition), the weight-gradient GEMM of layer l + 1 (gemmx2_tn16, pre-split Z) and NOTHING else -- the
//    data-gradient epilogue of layer l + 1 must evaluate act(lin_l) again rather than read out_l (recompute_out; sine needs
//    no out), out_L feeds the final linear layer in fp32, relu's out carries its backward's sign decisions;
//  * l >= 1 (out_0 comes from first_fwd_kernel);
//  * max |out_l| has a bound that is known here AND is reached in practice, so that a scale fixed on the host wastes no
//    fp16 range: sine and Gaussian <= 1; Gabor |exp(j w lin - s^2 |lin|^2)| = exp(-w v - s^2 (u^2 + v^2)) <= exp(w^2 / 4 s^2),
//    attained at lin = -j w / 2 s^2 -- accepted up to 16 (w / s <= 3.33: every configuration of the reference's scripts;
//    beyond it the maximum is tracked on the device as before).  bound < 2^e  ->  scale 2^(15 - e): |out| scale < 2^15.
float fused_act_scale(const Plan& p);
// The training forward of this call runs as ONE kernel (wire_fused.hip) that stores lin_l / out_l on the way: the shapes
// that have a kernel (256-feature real nets; wire at P = 192 / 256 -- P = 384 spills in its storing edition), the 2 x fp16
// family with its pre-split activations and recompute_out (the formats that kernel writes), a bound on the activations (all
// kinds but relu).  Decides the FORMAT of the stored out_l (pre-split at scale 1), so the backward asks the same question.
bool fused_train_applies(const Plan& p, int64_t n) {
  if (!p.k_fused_train || p.off_fx < 0 || p.L < 1 || p.L > 8) return false;
  if (!use_x2(p, n) || !gemmx2_tn_applies(p.Pl, p.P) || !p.k_split_out || !p.k_recompute_out) return false;
  if (p.kind == WIRE_KIND_WIRE && p.P > 256 && env_int_("WIRE_FUSED_TRAIN_P384", 0) == 0) return false;
  if (p.kind != WIRE_KIND_RELU && fused_act_scale(p) == 0.f) return false;
  return fused_pre_scale(p.kind, p.w1, p.s) > 0.f && fused_pre_scale(p.kind, p.w, p.s) > 0.f;
}
// ... and its data gradients g_lin_L -> .. -> g_lin_1 -> the first layer's gradient sums as ONE kernel
bool fused_bwd_applies(const Plan& p, int64_t n) {
  return p.k_fused_bwd && p.off_fxd >= 0 && p.L >= 1 && fused_train_applies(p, n);
}
// ... and then the sine / Gaussian nets store NO out_l below L and their lin_l (l < L, layer 0 included) as r = c lin (the
// argument the activation was evaluated on): the chain differentiates on r, the weight-gradient loader evaluates act(r) again
// -- 1.3 GB per step less
bool fused_rstore(const Plan& p, int64_t n) {
  return (p.kind == WIRE_KIND_SIREN || p.kind == WIRE_KIND_GAUSS) && p.k_rstore && fused_bwd_applies(p, n);
}
float out_split_scale(const Plan& p, int64_t n, int l) {
  if (!p.k_split_out || l < 1 || l >= p.L || p.kind == WIRE_KIND_RELU) return 0.f;
  if (!use_x2(p, n) || !gemmx2_tn_applies(p.Pl, p.P)) return 0.f;
  if (!p.k_recompute_out && p.kind != WIRE_KIND_SIREN) return 0.f;
  if (fused_train_applies(p, n)) return 1.f;               // the fused forward splits its bounded activations unscaled
  double bound = 1.0;
  if (p.cplx) {
    if (!(p.s > 0.f)) return 0.f;
    const double r = (double)p.w / (2.0 * (double)p.s);
    if (!(r * r <= 2.7725887)) return 0.f;                 // ln 16
    bound = exp(r * r);
  }
  const int e = ilogb(bound) + 1;
  return ldexpf(1.f, 15 - e);
}
// Power-of-two scale with which the fused forward (wire_fused.hip) splits the activations it keeps in registers: from
// their a-priori bound as above (sine, Gaussian <= 1; Gabor <= exp(w^2 / 4 s^2), accepted up to 16); 0 = none known
// (relu: the kernel takes each wave's own maximum; Gabor beyond the bound: the layer-by-layer path runs)
float fused_act_scale(const Plan& p) {
  if (p.kind == WIRE_KIND_RELU || p.kind == WIRE_KIND_WIRE2D) return 0.f;
  double bound = 1.0;
  if (p.kind == WIRE_KIND_WIRE) {
    if (!(p.s > 0.f)) return 0.f;
    const double r = (double)p.w / (2.0 * (double)p.s);
    if (!(r * r <= 2.7725887)) return 0.f;
    bound = exp(r * r);
  }
  return ldexpf(1.f, 15 - (ilogb(bound) + 1));
}
int epi_fwd(int kind) {
  switch (kind) {
    case WIRE_KIND_WIRE: return EPI_GABOR_FWD;
    case WIRE_KIND_WIRE2D: return EPI_GABOR2D_FWD;
    case WIRE_KIND_SIREN: return EPI_SIREN_FWD;
    case WIRE_KIND_GAUSS: return EPI_GAUSS_FWD;
    default: return EPI_RELU_FWD;
  }
}
int epi_bwd(int kind) {
  switch (kind) {
    case WIRE_KIND_WIRE: return EPI_GABOR_BWD;
    case WIRE_KIND_WIRE2D: return EPI_GABOR2D_BWD;
    case WIRE_KIND_SIREN: return EPI_SIREN_BWD;
    case WIRE_KIND_GAUSS: return EPI_GAUSS_BWD;
    default: return EPI_RELU_BWD;
  }
}
}  // namespace

// ---------------------------------------------------------------------------
// size queries
// ---------------------------------------------------------------------------
extern "C" int wire_num_param_tensors(const wire_net_desc* d) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  return p.ntens;
}
extern "C" int64_t wire_param_tensor_floats(const wire_net_desc* d, int t) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  if (t < 0 || t >= p.ntens) return fail(WIRE_ERR_ARG, "tensor index %d out of range", t);
  return p.tfloats[t];
}
extern "C" int64_t wire_packed_floats(const wire_net_desc* d) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  return p.total_packed;
}
extern "C" int64_t wire_act_bytes(const wire_net_desc* d, int64_t n, int save_for_bwd) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  if (n < 0) return fail(WIRE_ERR_ARG, "negative n");
  return act_layout(p, n, save_for_bwd).total * 4 + 256;
}
extern "C" int64_t wire_bwd_scratch_bytes(const wire_net_desc* d, int64_t n) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  if (n < 0) return fail(WIRE_ERR_ARG, "negative n");
  return scratch_layout(p, n).total * 4 + 256;
}
extern "C" int wire_blocked_width(int K) { return rup(2 * K, 64); }
// float offset of out_l (rows of P floats, l = 0..L) inside an act buffer laid out for n rows with save_for_bwd = 1
extern "C" int64_t wire_act_out_offset(const wire_net_desc* d, int64_t n, int layer) {
  Plan p; if (make_plan(d, p)) return WIRE_ERR_ARG;
  if (n < 0 || layer < 0 || layer > p.L) return fail(WIRE_ERR_ARG, "bad argument to wire_act_out_offset");
  const ActLayout a = act_layout(p, n, 1);
  return a.out0 + (int64_t)layer * a.np * p.P;
}

// ---------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------
extern "C" int wire_pack_params(void* stream, const wire_net_desc* d, const void* const* params,
                                float* packed) {
  Plan p; if (int rc = make_plan(d, p)) return rc;
  if (!params || !packed) return fail(WIRE_ERR_ARG, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(s, 3, 0);
  for (int i = 0; i < p.ntens; ++i)
    if (!params[i]) return fail(WIRE_ERR_ARG, "params[%d] is null", i);
  HIPCHK(hipMemsetAsync(packed + p.off_wamax, 0, (size_t)(p.L + 1) * WIRE_AMAX_SLOTS * sizeof(float), s));
  if (p.first_gemm) {                                    // layer 0 as a GEMM (positional encoding): its own shape
    const float* W = (const float*)params[0];
    const float* b = (const float*)params[1];
    HIPCHK(launch_pack_hidden(s, p.kind, W, b, nullptr, nullptr, p.K, p.Din, p.P, p.Pin0, packed + p.off_fwd[0],
                              packed + p.off_dg[0], packed + p.off_bias[0]));
    HIPCHK(launch_x3_split_b(s, packed + p.off_fwd[0], p.Pin0, p.Pl, p.Pin0, packed + p.off_fwd_x3[0]));
    HIPCHK(launch_x3_split_b(s, packed + p.off_dg[0], p.Pl, p.Pin0, p.Pl, packed + p.off_dg_x3[0]));
    if (p.off_fx >= 0) {                                   // the fused forward's image of the GEMM first layer + its maximum
      unsigned* slots0 = reinterpret_cast<unsigned*>(packed + p.off_wamax);
      HIPCHK(launch_amax(s, packed + p.off_fwd[0], (int64_t)p.Pl * p.Pin0, slots0));
      FxSplitBatch f0{};
      f0.src[0] = packed + p.off_fwd[0]; f0.dst[0] = packed + p.off_fx; f0.slots[0] = slots0;
      HIPCHK(launch_fx_split_b_batch(s, f0, 1, p.Pin0, p.P, 1.f, p.Pin0));
    }
  }
  // hidden layers share one shape: every family's image of up to PACK_MAXB layers per launch (3 - 4 launches per step
  // instead of 4 per layer; this runs once per optimizer step and is all launch gaps)
  for (int l0 = 1; l0 <= p.L; l0 += PACK_MAXB) {
    const int nb = (p.L - l0 + 1) < PACK_MAXB ? (p.L - l0 + 1) : PACK_MAXB;
    PackBatch pb{}, p3{};
    X3SplitBatch sf{}, sd{};
    for (int i = 0; i < nb; ++i) {
      const int l = l0 + i;
      pb.W[i] = (const float*)params[p.per_layer * l];
      pb.b[i] = (const float*)params[p.per_layer * l + 1];
      pb.V[i] = p.per_layer == 4 ? (const float*)params[p.per_layer * l + 2] : nullptr;
      pb.c[i] = p.per_layer == 4 ? (const float*)params[p.per_layer * l + 3] : nullptr;
      pb.fwd[i] = packed + p.off_fwd[l]; pb.dg[i] = packed + p.off_dg[l]; pb.bias[i] = packed + p.off_bias[l];
      sf.src[i] = packed + p.off_fwd[l]; sf.dst[i] = packed + p.off_fwd_x3[l];
      sd.src[i] = packed + p.off_dg[l]; sd.dst[i] = packed + p.off_dg_x3[l];
      if (p.off_fwd_3m[l] >= 0) {
        p3.W[i] = pb.W[i]; p3.b[i] = pb.b[i];
        p3.fwd[i] = packed + p.off_fwd_3m[l]; p3.dg[i] = packed + p.off_dg_3m[l]; p3.bias[i] = packed + p.off_bias[l];
      }
    }
    HIPCHK(launch_pack_hidden_batch(s, p.kind, pb, nb, p.K, p.K, p.P, p.P));
    if (p.Pl == p.P) {                                   // forward and transposed images have one shape: one launch
      for (int i = 0; i < nb; ++i) { sf.src[nb + i] = sd.src[i]; sf.dst[nb + i] = sd.dst[i]; }
      HIPCHK(launch_x3_split_b_batch(s, sf, 2 * nb, p.P, p.Pl, p.P));
    } else {
      HIPCHK(launch_x3_split_b_batch(s, sf, nb, p.P, p.Pl, p.P));
      HIPCHK(launch_x3_split_b_batch(s, sd, nb, p.Pl, p.P, p.Pl));
    }
    if (p.off_fwd_3m[l0] >= 0)    // same bias image (blocked planar) as the hidden pack wrote
      HIPCHK(launch_pack3m_batch(s, p3, nb, p.K, p.K, p.Kp, p.Kp));
    {
      // 2 x fp16 images: max |weight| of each layer (the forward and the transposed image hold the same values), then
      // the scaled split of both images
      X2AmaxBatch ab{};
      X2SplitBatch xf{}, xd{};
      for (int i = 0; i < nb; ++i) {
        const int l = l0 + i;
        unsigned* slots = reinterpret_cast<unsigned*>(packed + p.off_wamax + (int64_t)l * WIRE_AMAX_SLOTS);
        ab.src[i] = packed + p.off_fwd[l]; ab.slots[i] = slots;
        xf.src[i] = packed + p.off_fwd[l]; xf.dst[i] = packed + p.off_fwd_x2[l]; xf.slots[i] = slots;
        xd.src[i] = packed + p.off_dg[l]; xd.dst[i] = packed + p.off_dg_x2[l]; xd.slots[i] = slots;
      }
      HIPCHK(launch_amax_batch(s, ab, nb, (int64_t)p.Pl * p.P));
      HIPCHK(launch_x2_split_b_batch(s, xf, nb, p.P, p.Pl, p.P));
      HIPCHK(launch_x2_split_b_batch(s, xd, nb, p.Pl, p.P, p.Pl));
      if (p.off_fx >= 0 && fused_pre_scale(p.kind, p.w, p.s) > 0.f) {   // the fused forward's edition of the forward image
        FxSplitBatch fx{};
        for (int i = 0; i < nb; ++i) {
          const int l = l0 + i;
          fx.src[i] = packed + p.off_fwd[l];
          fx.dst[i] = packed + fx_hidden_off(p) + (int64_t)(l - 1) * fused_b_image_floats(p.P);
          fx.slots[i] = ab.slots[i];
        }
        HIPCHK(launch_fx_split_b_batch(s, fx, nb, p.P, p.P, fused_pre_scale(p.kind, p.w, p.s)));
      }
      if (p.off_fxd >= 0) {                                // the data-gradient chain's: transposed images, layers L .. 1
        FxSplitBatch fd{};
        int m = 0;
        for (int i = 0; i < nb; ++i) {
          const int l = l0 + i;
          fd.src[m] = packed + p.off_dg[l];
          fd.dst[m] = packed + p.off_fxd + (int64_t)(p.L - l) * fused_b_image_floats(p.P);
          fd.slots[m] = ab.slots[i];
          ++m;
        }
        if (m > 0) HIPCHK(launch_fx_split_b_batch(s, fd, m, p.Pl, p.P, 1.f));
      }
    }
  }
  HIPCHK(launch_pack_final(s, p.kind, (const float*)params[p.ntens - 2],
                           (const float*)params[p.ntens - 1], p.K, p.P, p.O, packed + p.off_wf,
                           packed + p.off_bf));
  if (!p.first_gemm)
    for (int q = 0; q < p.per_layer; ++q)
      HIPCHK(hipMemcpyAsync(packed + first_native_off(p, q), params[q], p.tfloats[q] * 4,
                            hipMemcpyDeviceToDevice, s));
  return WIRE_OK;
}

// ---------------------------------------------------------------------------
// whole-network forward
// ---------------------------------------------------------------------------
static int mlp_fwd_core(void* stream, const Plan& p, const float* packed, const float* coords, int64_t n,
                        float* y, void* act, int64_t act_bytes, int save_for_bwd, bool do_final,
                        bool skip_last_out = false, const FusedFwdParams* loss = nullptr, bool* loss_done = nullptr) {
  if (n < 0) return fail(WIRE_ERR_ARG, "negative n");
  if (n == 0) return WIRE_OK;
  if (!packed || !coords || (do_final && !y) || !act) return fail(WIRE_ERR_ARG, "null pointer");
  const ActLayout a = act_layout(p, n, save_for_bwd);
  if (act_bytes < a.total * 4) return fail(WIRE_ERR_SIZE, "act buffer %lld < %lld bytes",
                                           (long long)act_bytes, (long long)a.total * 4);
  hipStream_t s = (hipStream_t)stream;
  float* A = (float*)act;
  const bool x2 = use_x2(p, n);
  if (!save_for_bwd && do_final && x2) {
    // forward-only: the whole net in one kernel, activations in registers (wire_fused.hip)
    const float a_scale = fused_act_scale(p);
    if (p.off_fx >= 0 && p.k_fused_fwd && p.O <= 4 && p.L <= 8 && (a_scale != 0.f || p.kind == WIRE_KIND_RELU) &&
        fused_pre_scale(p.kind, p.w1, p.s) > 0.f && fused_pre_scale(p.kind, p.w, p.s) > 0.f) {
      FusedFwdParams fp;
      fp.coords = coords; fp.n = n;
      if (p.first_gemm) {
        fp.pe_F = p.F; fp.bias0 = packed + p.off_bias[0]; fp.wamax0 = reinterpret_cast<const unsigned*>(packed + p.off_wamax);
      } else {
        fp.W0 = packed + first_native_off(p, 0); fp.b0 = packed + first_native_off(p, 1);
      }
      fp.wimg = reinterpret_cast<const unsigned char*>(packed + p.off_fx);
      fp.bias = packed + p.off_bias[1]; fp.bias_stride = p.L >= 2 ? p.off_bias[2] - p.off_bias[1] : 0;
      fp.wamax = reinterpret_cast<const unsigned*>(packed + p.off_wamax) + WIRE_AMAX_SLOTS; fp.wamax_stride = WIRE_AMAX_SLOTS;
      fp.wf = packed + p.off_wf; fp.bfr = packed + p.off_bf; fp.y = y;
      fp.D = p.D; fp.K = p.K; fp.L = p.L; fp.O = p.O; fp.w1 = p.w1; fp.w = p.w; fp.s = p.s;
      fp.c_first = fused_pre_scale(p.kind, p.w1, p.s); fp.c_hidden = fused_pre_scale(p.kind, p.w, p.s);
      fp.k2_first = p.s * p.s * 1.44269502f / (fp.c_first * fp.c_first);
      fp.k2 = p.s * p.s * 1.44269502f / (fp.c_hidden * fp.c_hidden);
      ProfScope ps(s, 0, 2.0 * n * p.Pl * p.P * p.L);
      HIPCHK(launch_fused_fwd(s, p.kind, p.P, fp));
      return WIRE_OK;
    }
  }
  unsigned* const amax = reinterpret_cast<unsigned*>(A + a.amax);            // slots of out_l at amax + 64 l
  auto wamax = [&](int l) { return reinterpret_cast<const unsigned*>(packed + p.off_wamax + (int64_t)l * WIRE_AMAX_SLOTS); };
  if (x2) HIPCHK(hipMemsetAsync(amax, 0, (size_t)(p.L + 2) * WIRE_AMAX_SLOTS * sizeof(unsigned), s));
  if (save_for_bwd && !do_final && p.O <= 4 && fused_train_applies(p, n) &&
      (skip_last_out || p.kind == WIRE_KIND_RELU)) {
    // training forward of wire_train_fwd_bwd: the hidden layers in one kernel that stores lin_l (not relu), out_0 (fp32 +
    // its maximum), out_1 .. out_{L-1} (pre-split pairs; relu: fp32 + maxima, and out_L) -- what the fused final stage,
    // the data-gradient epilogues and the weight-gradient GEMMs of mlp_bwd_core read (wire_fused.hip)
    FusedFwdParams fp;
    fp.coords = coords; fp.n = n;
    if (p.first_gemm) {
      // the encoded features themselves are still written: the first layer's weight gradient reads them (mlp_bwd_core)
      { ProfScope ps(s, 3, 0);
        HIPCHK(launch_posenc(s, coords, n, p.D, p.F, p.Pin0, A + a.pe)); }
      fp.pe_F = p.F; fp.bias0 = packed + p.off_bias[0]; fp.wamax0 = reinterpret_cast<const unsigned*>(packed + p.off_wamax);
    } else {
      fp.W0 = packed + first_native_off(p, 0); fp.b0 = packed + first_native_off(p, 1);
    }
    fp.wimg = reinterpret_cast<const unsigned char*>(packed + p.off_fx);
    fp.bias = packed + p.off_bias[1]; fp.bias_stride = p.L >= 2 ? p.off_bias[2] - p.off_bias[1] : 0;
    fp.wamax = reinterpret_cast<const unsigned*>(packed + p.off_wamax) + WIRE_AMAX_SLOTS; fp.wamax_stride = WIRE_AMAX_SLOTS;
    fp.D = p.D; fp.K = p.K; fp.L = p.L; fp.O = p.O; fp.w1 = p.w1; fp.w = p.w; fp.s = p.s;
    fp.c_first = fused_pre_scale(p.kind, p.w1, p.s); fp.c_hidden = fused_pre_scale(p.kind, p.w, p.s);
    fp.k2_first = p.s * p.s * 1.44269502f / (fp.c_first * fp.c_first);
    fp.k2 = p.s * p.s * 1.44269502f / (fp.c_hidden * fp.c_hidden);
    fp.inv_c_first = 1.f / fp.c_first; fp.inv_c_hidden = 1.f / fp.c_hidden;
    fp.lin0 = p.cplx ? nullptr : A + a.lin0;
    fp.lin = p.kind == WIRE_KIND_RELU ? nullptr : A + a.lin1; fp.lin_stride = a.np * p.Pl;
    fp.out = A + a.out0; fp.out_stride = a.np * p.P;
    fp.amax_out = amax;
    fp.rstore = fused_rstore(p, n) ? 1 : 0;
    if (loss && loss_done && !p.cplx && p.k_fused_final) {
      // the final stage inside this kernel (wire_fused.hip: fx_tail_loss): lin_L / out_L are not stored at all
      fp.wf = packed + p.off_wf; fp.bfr = packed + p.off_bf;
      fp.target = loss->target; fp.idx = loss->idx; fp.first = loss->first; fp.gscale = loss->gscale;
      fp.y = loss->y; fp.rec = loss->rec; fp.g_lin = loss->g_lin; fp.part_w = loss->part_w; fp.part_b = loss->part_b;
      fp.loss_partial = loss->loss_partial; fp.amax_g = loss->amax_g;
      *loss_done = true;
    }
    ProfScope ps(s, 0, 2.0 * n * p.Pl * p.P * p.L);
    HIPCHK(launch_fused_fwd(s, p.kind, p.P, fp));
    return WIRE_OK;
  }
  auto out_l = [&](int l) { return save_for_bwd ? A + a.out0 + (int64_t)l * a.np * p.P
                                                : A + ((l & 1) ? a.pong : a.ping); };
  auto lin_l = [&](int l) -> float* {
    // relu: out = max(lin, 0) carries everything its backward needs (lin > 0 <=> out > 0): lin is never written
    if (!save_for_bwd || p.kind == WIRE_KIND_RELU) return nullptr;
    return l == 0 ? A + a.lin0 : A + a.lin1 + (int64_t)(l - 1) * a.np * p.Pl;
  };
  // ---- layer 0
  if (p.first_gemm) {
    { ProfScope ps(s, 3, 0);
      HIPCHK(launch_posenc(s, coords, n, p.D, p.F, p.Pin0, A + a.pe)); }
    GemmEpiParams ep; ep.bias = packed + p.off_bias[0]; ep.o0 = lin_l(0); ep.o1 = out_l(0);
    ep.ld0 = p.Pl; ep.ld1 = p.P; ep.omega = p.w1; ep.scale = p.s; ep.kvalid = p.K;
    if (x2) ep.amax_out = amax;                          // (the 3 x bf16 16 x 16 x 32 kernel tracks the maximum too)
    ProfScope ps(s, 0, 2.0 * n * p.Pl * p.Pin0);
    if (p.x3)
      HIPCHK(launch_gemmx3_nt(s, epi_fwd(p.kind), A + a.pe, p.Pin0, packed + p.off_fwd_x3[0], n, p.Pl,
                              p.Pin0, ep));
    else
      HIPCHK(launch_gemm_nt(s, epi_fwd(p.kind), A + a.pe, p.Pin0, packed + p.off_fwd[0], p.Pin0, n,
                            p.Pl, p.Pin0, ep));
  } else {
    const float* W0 = packed + first_native_off(p, 0);
    const float* b0 = packed + first_native_off(p, 1);
    const float* V0 = p.per_layer == 4 ? packed + first_native_off(p, 2) : nullptr;
    const float* c0 = p.per_layer == 4 ? packed + first_native_off(p, 3) : nullptr;
    ProfScope ps(s, 3, 0);
    HIPCHK(launch_first_fwd(s, p.kind, coords, n, p.D, W0, b0, V0, c0, p.K, p.P, p.w1, p.s,
                            p.cplx ? nullptr : lin_l(0), out_l(0), x2 ? amax : nullptr));
  }
  // ---- hidden layers
  for (int l = 1; l <= p.L; ++l) {
    GemmEpiParams ep; ep.bias = packed + p.off_bias[l]; ep.o0 = lin_l(l); ep.o1 = out_l(l);
    ep.ld0 = p.Pl; ep.ld1 = p.P; ep.omega = p.w; ep.scale = p.s; ep.kvalid = p.K;
    if (skip_last_out && l == p.L) ep.o1 = nullptr;       // wire_train_fwd_bwd: the final stage recomputes it
    ProfScope ps(s, 0, 2.0 * n * p.Pl * p.P);
    if (x2) {
      ep.amax_a = amax + (l - 1) * WIRE_AMAX_SLOTS; ep.amax_b = wamax(l); ep.amax_out = amax + l * WIRE_AMAX_SLOTS;
      // pre-split activations: out_{l-1} read as such, out_l written as such (its maximum slots stay zero: not read)
      const float s_in = out_split_scale(p, n, l - 1), s_out = ep.o1 ? out_split_scale(p, n, l) : 0.f;
      if (s_in != 0.f) ep.a_split_inv = 1.f / s_in;
      if (s_out != 0.f) { ep.o1_split = s_out; ep.amax_out = nullptr; }
      HIPCHK(launch_gemmx2h_nt(s, epi_fwd(p.kind), out_l(l - 1), p.P, packed + p.off_fwd_x2[l], n, p.Pl, p.P, ep));
    } else if (p.m3)
      HIPCHK(launch_gemm3m_nt(s, EPI_GABOR_FWD, out_l(l - 1), p.P, packed + p.off_fwd_3m[l], p.P, n, p.Kp,
                              p.Kp, ep));
    else if (p.x3)
      HIPCHK(launch_gemmx3_nt(s, epi_fwd(p.kind), out_l(l - 1), p.P, packed + p.off_fwd_x3[l], n, p.Pl,
                              p.P, ep));
    else
      HIPCHK(launch_gemm_nt(s, epi_fwd(p.kind), out_l(l - 1), p.P, packed + p.off_fwd[l], p.P, n,
                            p.Pl, p.P, ep));
  }
  if (do_final) {
    ProfScope ps(s, 3, 0);
    HIPCHK(launch_final_fwd(s, out_l(p.L), n, p.P, p.O, packed + p.off_wf, packed + p.off_bf, y));
  }
  return WIRE_OK;
}

extern "C" int wire_mlp_fwd(void* stream, const wire_net_desc* d, const float* packed,
                            const float* coords, int64_t n, float* y, void* act, int64_t act_bytes,
                            int save_for_bwd) {
  Plan p; if (int rc = make_plan(d, p)) return rc;
  return mlp_fwd_core(stream, p, packed, coords, n, y, act, act_bytes, save_for_bwd, true);
}

// ---------------------------------------------------------------------------
// whole-network backward
// ---------------------------------------------------------------------------
// do_final = false: the caller (wire_train_fwd_bwd) has already run the fused final stage, i.e.
// g_lin of layer L is in the scratch's first gradient buffer and the final-layer partials are there.
static int mlp_bwd_core(void* stream, const Plan& p, const float* packed, const float* coords, int64_t n,
                        const float* g_y, const void* act, int64_t act_bytes, void* scratch,
                        int64_t scratch_bytes, void* const* grads, bool do_final,
                        wire_grad_ready_fn ready = nullptr, void* user = nullptr, int final_blocks = 0) {
  if (n <= 0) return fail(WIRE_ERR_ARG, "backward needs n > 0");
  if (!packed || !coords || (do_final && !g_y) || !act || !scratch || !grads) return fail(WIRE_ERR_ARG, "null pointer");
  for (int i = 0; i < p.ntens; ++i) if (!grads[i]) return fail(WIRE_ERR_ARG, "grads[%d] is null", i);
  const ActLayout a = act_layout(p, n, 1);
  const ScratchLayout sc = scratch_layout(p, n);
  if (act_bytes < a.total * 4) return fail(WIRE_ERR_SIZE, "act buffer too small");
  if (scratch_bytes < sc.total * 4) return fail(WIRE_ERR_SIZE, "scratch %lld < %lld bytes",
                                                (long long)scratch_bytes, (long long)sc.total * 4);
  hipStream_t s = (hipStream_t)stream;
  const float* A = (const float*)act;
  float* Sx = (float*)scratch;
  auto out_l = [&](int l) { return A + a.out0 + (int64_t)l * a.np * p.P; };
  auto lin_l = [&](int l) { return l == 0 ? A + a.lin0 : A + a.lin1 + (int64_t)(l - 1) * a.np * p.Pl; };
  float* gcur = Sx + sc.ga;
  float* gnext = Sx + sc.gb;
  const bool x2 = use_x2(p, n);
  unsigned* const gamax = reinterpret_cast<unsigned*>(Sx + sc.gamax);          // slots of g_lin_l at gamax + 64 l
  const unsigned* const amax = reinterpret_cast<const unsigned*>(A + a.amax);  // slots of out_l, filled by the forward
  auto wamax = [&](int l) { return reinterpret_cast<const unsigned*>(packed + p.off_wamax + (int64_t)l * WIRE_AMAX_SLOTS); };
  // (the fused path zeroed the slots before its final stage published max |g_lin_L|)
  if (x2 && do_final) HIPCHK(hipMemsetAsync(gamax, 0, (size_t)(p.L + 2) * WIRE_AMAX_SLOTS * sizeof(unsigned), s));
  const bool first_sums = p.k_first_sums && p.cplx && p.x3 && p.L >= 1 &&
                          gemmx3_nt_is_h16(p.kind == WIRE_KIND_WIRE ? EPI_GABOR_BWD_FIRST : EPI_GABOR2D_BWD_FIRST, n);
  const int64_t crp_set = (int64_t)(colreduce_blocks(n) + 32) * p.ldu * 5;   // wire2d: second set of partial sums
  // the same for siren / gauss / relu with a native first layer: the epilogue of the layer-1 data gradient sums g_lin_0 [x | 1]
  const bool first_sums_real = p.k_first_sums && !p.cplx && !p.first_gemm && p.x3 && p.L >= 1 &&
                               gemmx3_nt_is_h16(epi_bwd(p.kind), n);

  // ---- final linear + activation gradient of layer L
  // (final_blocks: the training forward formed the final layer's partial sums itself, one block per workgroup)
  const int nbf = final_blocks > 0 ? final_blocks : final_bwd_blocks(n);
  if (!do_final) {
    ProfScope ps(s, 3, 0);
    HIPCHK(launch_final_reduce(s, p.kind, Sx + sc.fpw, Sx + sc.fpb, nbf, p.O, p.K, p.P,
                               (float*)grads[p.ntens - 2], (float*)grads[p.ntens - 1]));
    if (ready) ready(user, p.ntens - 2, 2);
  } else {
    ProfScope ps(s, 3, 0);
    const float wL = (p.L == 0) ? p.w1 : p.w;
    if (p.L == 0 && p.cplx) {
      // no hidden layer: g_out0 is needed raw; handled below through the raw path
      HIPCHK(launch_final_bwd(s, p.kind, 1, g_y, n, p.O, packed + p.off_wf, nullptr, out_l(0), p.K,
                              p.P, wL, p.s, gcur, Sx + sc.fpw, Sx + sc.fpb));
    } else {
      HIPCHK(launch_final_bwd(s, p.kind, 0, g_y, n, p.O, packed + p.off_wf, lin_l(p.L), out_l(p.L),
                              p.K, p.P, wL, p.s, gcur, Sx + sc.fpw, Sx + sc.fpb,
                              x2 ? gamax + p.L * WIRE_AMAX_SLOTS : nullptr));
    }
    HIPCHK(launch_final_reduce(s, p.kind, Sx + sc.fpw, Sx + sc.fpb, nbf, p.O, p.K, p.P,
                               (float*)grads[p.ntens - 2], (float*)grads[p.ntens - 1]));
    if (ready) ready(user, p.ntens - 2, 2);
  }
  if (p.L == 0 && p.cplx) {
    // no hidden layer (net = first Gabor layer + final linear): gcur holds the raw g_out0; the first layer's
    // activation gradient is an elementwise pass (u recomputed from the coordinates), then the column sums below
    ProfScope ps(s, 3, 0);
    const float* W0 = packed + first_native_off(p, 0);
    const float* b0 = packed + first_native_off(p, 1);
    if (p.kind == WIRE_KIND_WIRE)
      HIPCHK(launch_gabor_bwd_first_point(s, gcur, out_l(0), coords, p.D, W0, b0, n, p.K, p.P, p.w1, p.s,
                                          Sx + sc.gu, p.ldu));
    else
      HIPCHK(launch_gabor2d_bwd_first_point(s, gcur, out_l(0), coords, p.D, W0, b0, packed + first_native_off(p, 2),
                                            packed + first_native_off(p, 3), n, p.K, p.P, p.w1, p.s, Sx + sc.gu,
                                            p.ldu));
  }

  // ---- the data gradients of layers L .. 1 as one chain (wire_fused.hip): every g_lin_l (l >= 1) lands in its own buffer,
  // the weight-gradient GEMMs below read them; the last link forms the first layer's sums g_lin_0^T [x | 1] per workgroup
  // (positional encoding: stores g_lin_0 for the first layer's weight-gradient GEMM)
  const bool chain = !do_final && fused_bwd_applies(p, n);
  int chain_rows = 0;
  if (chain) {
    FusedBwdParams bp;
    bp.n = n;
    bp.g = Sx + sc.gch; bp.g_stride = sc.gch_stride;
    bp.gamax = gamax;
    bp.aux = p.kind == WIRE_KIND_RELU ? A + a.out0 : A + a.lin1 - a.np * p.Pl;   // lin_l at lin1 + (l - 1) * np * Pl
    bp.aux_stride = p.kind == WIRE_KIND_RELU ? a.np * p.P : a.np * p.Pl;
    bp.wimg = reinterpret_cast<const unsigned char*>(packed + p.off_fxd);
    bp.wamax = reinterpret_cast<const unsigned*>(packed + p.off_wamax) + WIRE_AMAX_SLOTS; bp.wamax_stride = WIRE_AMAX_SLOTS;
    bp.L = p.L; bp.w = p.w; bp.s = p.s;
    bp.rstore = fused_rstore(p, n) ? 1 : 0; bp.c_hidden = fused_pre_scale(p.kind, p.w, p.s);
    bp.aux0 = p.kind == WIRE_KIND_RELU ? A + a.out0 : A + a.lin0;
    bp.w1 = p.w1;
    if (!p.first_gemm) { bp.coords = coords; bp.D = p.D; bp.crp = Sx + sc.crp; bp.C = p.K; }
    ProfScope ps(s, 1, 2.0 * n * p.Pl * p.P * p.L);
    HIPCHK(launch_fused_bwd(s, p.kind, p.P, bp, &chain_rows));
  }
  // ---- behind the chain every g_lin_l exists before the first weight gradient starts: those of layers 2 .. L (same shape,
  // operands a fixed step apart) run as ONE launch, each member accumulating L - 1 times the rows into a third of the slabs
  // (layer 1 joins when its activation operand has the form of the others': r_0 of the sine / Gaussian nets, relu's fp32 out_0)
  int wbatch = 0, wbatch_S = 0, wbatch_l0 = 2;
  if (chain && p.k_wgrad_batch && x2 && gemmx2_tn_applies(p.Pl, p.P)) {
    const bool rs = fused_rstore(p, n);
    const int l0 = (rs || p.kind == WIRE_KIND_RELU) ? 1 : 2;
    const int nb = p.L - l0 + 1;
    const int S = nb >= 2 ? gemmx2_tn_batch_splits(n, p.Pl, p.P, sc.S, nb) : 0;
    if (S >= 1 && (int64_t)S * nb <= sc.S) {
      const float s_z = rs ? 0.f : out_split_scale(p, n, l0);
      ProfScope ps(s, 2, 2.0 * n * p.Pl * p.P * nb);
      HIPCHK(launch_gemmx2_tn(s, Sx + sc.gch + l0 * sc.gch_stride, p.Pl, rs ? lin_l(l0 - 1) : out_l(l0 - 1), rs ? p.Pl : p.P, n,
                              p.Pl, p.P, S, Sx + sc.slab, Sx + sc.bslab, gamax + l0 * WIRE_AMAX_SLOTS,
                              amax + (l0 - 1) * WIRE_AMAX_SLOTS, rs ? 1.f / 16384.f : (s_z != 0.f ? 1.f / s_z : 0.f),
                              rs ? (p.kind == WIRE_KIND_SIREN ? 2 : 3) : 0, nb, sc.gch_stride,
                              rs ? a.np * p.Pl : a.np * p.P, WIRE_AMAX_SLOTS));
      wbatch = nb; wbatch_S = S; wbatch_l0 = l0;
    }
  }
  // ---- hidden layers L..1
  for (int l = p.L; l >= 1; --l) {
    if (chain) gcur = Sx + sc.gch + (int64_t)l * sc.gch_stride;
    float* gW = (float*)grads[p.per_layer * l];
    float* gb = (float*)grads[p.per_layer * l + 1];
    float* gV = p.per_layer == 4 ? (float*)grads[p.per_layer * l + 2] : nullptr;
    float* gc = p.per_layer == 4 ? (float*)grads[p.per_layer * l + 3] : nullptr;
    if (p.m3) {
      const int S = sc.S;
      { ProfScope ps(s, 2, 2.0 * n * p.Pl * p.P);
        HIPCHK(launch_gemm3m_tn(s, gcur, p.P, out_l(l - 1), p.P, n, p.Kp, p.Kp, S, Sx + sc.slab,
                                Sx + sc.bslab)); }
      ProfScope ps(s, 3, 0);
      HIPCHK(launch_wgrad3m_reduce(s, Sx + sc.slab, Sx + sc.bslab, S, p.K, p.K, p.Kp, p.Kp, gW, gb));
      if (ready) ready(user, p.per_layer * l, p.per_layer);
    } else if (wbatch > 0 && l >= wbatch_l0) {
      // the batch above wrote this layer's slabs: member l - l0
      ProfScope ps(s, 3, 0);
      HIPCHK(launch_wgrad_reduce(s, p.kind, Sx + sc.slab + (int64_t)(l - wbatch_l0) * wbatch_S * p.Pl * p.P,
                                 Sx + sc.bslab + (int64_t)(l - wbatch_l0) * wbatch_S * p.Pl, wbatch_S, p.K, p.K, p.Pl, p.P, gW, gb,
                                 gV, gc));
      if (ready) ready(user, p.per_layer * l, p.per_layer);
    } else {
      const bool x2tn = x2 && gemmx2_tn_applies(p.Pl, p.P);
      const int S = x2tn ? gemmx2_tn_splits(n, p.Pl, p.P, sc.S)
                         : (p.x3 ? gemmx3_tn_splits(n, p.Pl, p.P, sc.S) : gemm_tn_splits(n, p.Pl, p.P, sc.S));
      { ProfScope ps(s, 2, 2.0 * n * p.Pl * p.P);
        if (x2tn && chain && fused_rstore(p, n)) {
          // Z = act(r_{l-1}) evaluated by the loader from the stored pre-activation (no out_{l-1} exists), scale 2^14
          HIPCHK(launch_gemmx2_tn(s, gcur, p.Pl, lin_l(l - 1), p.Pl, n, p.Pl, p.P, S, Sx + sc.slab, Sx + sc.bslab,
                                  gamax + l * WIRE_AMAX_SLOTS, nullptr, 1.f / 16384.f,
                                  p.kind == WIRE_KIND_SIREN ? 2 : 3));
        } else if (x2tn) {
          const float s_z = out_split_scale(p, n, l - 1);
          HIPCHK(launch_gemmx2_tn(s, gcur, p.Pl, out_l(l - 1), p.P, n, p.Pl, p.P, S, Sx + sc.slab, Sx + sc.bslab,
                                  gamax + l * WIRE_AMAX_SLOTS, amax + (l - 1) * WIRE_AMAX_SLOTS,
                                  s_z != 0.f ? 1.f / s_z : 0.f));
        }
        else if (p.x3)
          HIPCHK(launch_gemmx3_tn(s, gcur, p.Pl, out_l(l - 1), p.P, n, p.Pl, p.P, S, Sx + sc.slab,
                                  Sx + sc.bslab));
        else
          HIPCHK(launch_gemm_tn(s, gcur, p.Pl, out_l(l - 1), p.P, n, p.Pl, p.P, S, Sx + sc.slab,
                                Sx + sc.bslab)); }
      ProfScope ps(s, 3, 0);
      HIPCHK(launch_wgrad_reduce(s, p.kind, Sx + sc.slab, Sx + sc.bslab, S, p.K, p.K, p.Pl, p.P, gW,
                                 gb, gV, gc));
      if (ready) ready(user, p.per_layer * l, p.per_layer);
    }
    GemmEpiParams ep;
    ep.scale = p.s; ep.kvalid = p.K; ep.ld1 = p.P; ep.i1 = out_l(l - 1);
    int epi;
    if (l > 1 || !p.cplx) {
      epi = epi_bwd(p.kind);
      ep.omega = (l - 1 == 0) ? p.w1 : p.w;
      ep.i0 = lin_l(l - 1); ep.o0 = gnext; ep.ld0 = p.Pl;
      // hidden Gabor layer whose forward ran the lean 16 x 16 x 32 epilogue: out = act(lin) again, 8 B / element less
      // (real nets: only below a hidden layer -- the first layer's out comes from first_fwd_kernel's precise form)
      ep.recompute_out = p.k_recompute_out && p.x3 && (p.cplx || l - 1 >= 1) && gemmx3_nt_is_h16(epi_fwd(p.kind), n);
      if (l == 1 && first_sums_real) { ep.coords = coords; ep.D = p.D; ep.cr_partial = Sx + sc.crp; ep.cr_C = p.K; }
    } else {
      epi = (p.kind == WIRE_KIND_WIRE) ? EPI_GABOR_BWD_FIRST : EPI_GABOR2D_BWD_FIRST;
      ep.omega = p.w1;
      ep.coords = coords; ep.D = p.D; ep.ldu = p.ldu; ep.o0 = Sx + sc.gu;
      // wire on the 16 x 16 x 32 kernel: its epilogue sums g_u [x | 1] per 256-row tile itself (the first layer's weight
      // and bias gradient partials) instead of storing g_u for a separate pass
      if (first_sums) { ep.cr_partial = Sx + sc.crp; ep.cr_C = p.K; ep.cr_set = crp_set; }
      // (and re-evaluates out_0 from the u it recomputes anyway: first_fwd_kernel's own form, the same bits)
      ep.recompute_out = p.k_recompute_out && p.x3 && p.kind == WIRE_KIND_WIRE && gemmx3_nt_is_h16(EPI_GABOR_BWD_FIRST, n);
      ep.W0 = packed + first_native_off(p, 0); ep.b0 = packed + first_native_off(p, 1);
      if (p.per_layer == 4) { ep.W0b = packed + first_native_off(p, 2); ep.b0b = packed + first_native_off(p, 3); }
    }
    if (!p.cplx && l == 1) ep.ld0 = p.P;
    if (chain) continue;                                    // g_lin_{l-1} (l = 1: the first layer's sums) is already there
    { ProfScope ps(s, 1, 2.0 * n * p.Pl * p.P);
      if (x2) {
        ep.amax_a = gamax + l * WIRE_AMAX_SLOTS; ep.amax_b = wamax(l);
        ep.amax_out = l >= 2 ? gamax + (l - 1) * WIRE_AMAX_SLOTS : nullptr;   // g_lin_0 feeds no 2 x fp16 GEMM
        HIPCHK(launch_gemmx2h_nt(s, epi, gcur, p.Pl, packed + p.off_dg_x2[l], n, p.P, p.Pl, ep));
      } else if (p.m3)
        HIPCHK(launch_gemm3m_nt(s, epi, gcur, p.P, packed + p.off_dg_3m[l], p.P, n, p.Kp, p.Kp, ep));
      else if (p.x3)
        HIPCHK(launch_gemmx3_nt(s, epi, gcur, p.Pl, packed + p.off_dg_x3[l], n, p.P, p.Pl, ep));
      else
        HIPCHK(launch_gemm_nt(s, epi, gcur, p.Pl, packed + p.off_dg[l], p.Pl, n, p.P, p.Pl, ep)); }
    float* t = gcur; gcur = gnext; gnext = t;
  }

  // ---- first layer parameter gradients
  ProfScope ps(s, 3, 0);
  if (p.cplx) {
    const float* gu = Sx + sc.gu;
    if (p.kind == WIRE_KIND_WIRE && first_sums) {
      HIPCHK(launch_colreduce_final(s, p.K, p.D, n, Sx + sc.crp, (float*)grads[0], (float*)grads[1]));
    } else if (p.kind == WIRE_KIND_WIRE) {
      HIPCHK(launch_colreduce(s, gu, p.ldu, p.K, coords, p.D, n, Sx + sc.crp, (float*)grads[0],
                              (float*)grads[1]));
    } else if (first_sums) {
      HIPCHK(launch_colreduce_final(s, p.K, p.D, n, Sx + sc.crp, (float*)grads[0], (float*)grads[1]));
      HIPCHK(launch_colreduce_final(s, p.K, p.D, n, Sx + sc.crp + crp_set, (float*)grads[2], (float*)grads[3]));
    } else {
      HIPCHK(launch_colreduce(s, gu, 2 * p.ldu, p.K, coords, p.D, n, Sx + sc.crp, (float*)grads[0],
                              (float*)grads[1]));
      HIPCHK(launch_colreduce(s, gu + p.ldu, 2 * p.ldu, p.K, coords, p.D, n, Sx + sc.crp,
                              (float*)grads[2], (float*)grads[3]));
    }
  } else if (!p.first_gemm && chain) {
    HIPCHK(launch_colreduce_final_blocks(s, p.K, p.D, (int)((n + chain_rows - 1) / chain_rows), Sx + sc.crp, (float*)grads[0],
                                         (float*)grads[1]));
  } else if (!p.first_gemm && first_sums_real) {
    HIPCHK(launch_colreduce_final(s, p.K, p.D, n, Sx + sc.crp, (float*)grads[0], (float*)grads[1]));
  } else if (!p.first_gemm) {
    // gcur holds g_lin_0 [n][P]
    const float* g0 = (p.L == 0) ? Sx + sc.ga : gcur;
    HIPCHK(launch_colreduce(s, g0, p.P, p.K, coords, p.D, n, Sx + sc.crp, (float*)grads[0],
                            (float*)grads[1]));
  } else {
    const float* g0 = (p.L == 0) ? Sx + sc.ga : (chain ? Sx + sc.gch : gcur);   // the chain stored g_lin_0 in its slot 0
    const int S = p.x3 ? gemmx3_tn_splits(n, p.P, p.Pin0, sc.S) : gemm_tn_splits(n, p.P, p.Pin0, sc.S);
    if (p.x3)
      HIPCHK(launch_gemmx3_tn(s, g0, p.P, A + a.pe, p.Pin0, n, p.P, p.Pin0, S, Sx + sc.slab,
                              Sx + sc.bslab));
    else
      HIPCHK(launch_gemm_tn(s, g0, p.P, A + a.pe, p.Pin0, n, p.P, p.Pin0, S, Sx + sc.slab,
                            Sx + sc.bslab));
    HIPCHK(launch_wgrad_reduce(s, p.kind, Sx + sc.slab, Sx + sc.bslab, S, p.K, p.Din, p.P, p.Pin0,
                               (float*)grads[0], (float*)grads[1], nullptr, nullptr));
  }
  if (ready) ready(user, 0, p.per_layer);
  return WIRE_OK;
}

extern "C" int wire_mlp_bwd(void* stream, const wire_net_desc* d, const float* packed,
                            const float* coords, int64_t n, const float* g_y, const void* act,
                            int64_t act_bytes, void* scratch, int64_t scratch_bytes,
                            void* const* grads) {
  Plan p; if (int rc = make_plan(d, p)) return rc;
  return mlp_bwd_core(stream, p, packed, coords, n, g_y, act, act_bytes, scratch, scratch_bytes, grads, true);
}

// ---------------------------------------------------------------------------
// fused training core: forward -> MSE -> backward in one call
// ---------------------------------------------------------------------------
extern "C" int wire_train_fwd_bwd_hooked(void* stream, const wire_net_desc* d, const float* packed,
                                         const float* coords, int64_t n, const float* target, const int64_t* idx,
                                         int64_t first, float weight, float* y, float* g_y, float* loss_out,
                                         float* rec, float* partial, void* act, int64_t act_bytes, void* scratch,
                                         int64_t scratch_bytes, void* const* grads, wire_grad_ready_fn ready,
                                         void* user) {
  Plan p; if (int rc = make_plan(d, p)) return rc;
  if (n <= 0) return fail(WIRE_ERR_ARG, "wire_train_fwd_bwd needs n > 0");
  if (!target || !y || !g_y || !loss_out || !partial) return fail(WIRE_ERR_ARG, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  // every net kind with at least one hidden layer: final linear + loss + final backward + activation gradient of layer L
  // in one pass (wire_point.hip: final_fused_kernel)
  const bool fuse = p.L >= 1 && p.O <= 4 && final_fused_supported(p.P, p.O);
  if (!fuse) {
    if (int rc = mlp_fwd_core(stream, p, packed, coords, n, y, act, act_bytes, 1, true)) return rc;
    { ProfScope ps(s, 3, 0);
      HIPCHK(launch_mse_grad(s, y, target, idx, first, n, p.O, weight, g_y, loss_out, rec, partial)); }
    return mlp_bwd_core(stream, p, packed, coords, n, g_y, act, act_bytes, scratch, scratch_bytes, grads, true, ready,
                        user);
  }
  // with layer L on the 16 x 16 x 32 forward kernel (lean epilogue) the final stage evaluates out_L from lin_L itself,
  // bit for bit what that epilogue would have stored: out_L is neither written nor read (1 GB less HBM traffic)
  const bool recomp = p.k_recompute_out && p.x3 && p.kind != WIRE_KIND_RELU && gemmx3_nt_is_h16(epi_fwd(p.kind), n);
  const ActLayout a = act_layout(p, n, 1);
  const ScratchLayout sc = scratch_layout(p, n);
  if (!scratch || scratch_bytes < sc.total * 4) return fail(WIRE_ERR_SIZE, "scratch too small");
  const float* A = (const float*)act;
  float* Sx = (float*)scratch;
  const bool x2 = use_x2(p, n);
  unsigned* const gamax = reinterpret_cast<unsigned*>(Sx + sc.gamax);
  if (x2) HIPCHK(hipMemsetAsync(gamax, 0, (size_t)(p.L + 2) * WIRE_AMAX_SLOTS * sizeof(unsigned), s));
  float* gL = fused_bwd_applies(p, n) ? Sx + sc.gch + (int64_t)p.L * sc.gch_stride : Sx + sc.ga;
  // real nets on the whole-net training kernel: the final stage runs inside it (fx_tail_loss)
  FusedFwdParams lp;
  const int fblocks = (int)((n + 127) / 128);
  const double inv_no = 1.0 / ((double)n * (double)p.O);
  lp.target = target; lp.idx = idx; lp.first = first; lp.gscale = (float)(weight * 2.0 * inv_no);
  lp.y = y; lp.rec = rec; lp.g_lin = gL; lp.part_w = Sx + sc.fpw; lp.part_b = Sx + sc.fpb; lp.loss_partial = Sx + sc.crp;
  lp.amax_g = gamax + p.L * WIRE_AMAX_SLOTS;
  bool loss_done = false;
  if (int rc = mlp_fwd_core(stream, p, packed, coords, n, nullptr, act, act_bytes, 1, false, recomp,
                            x2 && fused_bwd_applies(p, n) ? &lp : nullptr, &loss_done)) return rc;
  if (loss_done) {
    { ProfScope ps(s, 3, 0);
      HIPCHK(launch_mse_final(s, Sx + sc.crp, fblocks, (float)(weight * inv_no), loss_out)); }
    return mlp_bwd_core(stream, p, packed, coords, n, nullptr, act, act_bytes, scratch, scratch_bytes, grads, false, ready,
                        user, fblocks);
  }
  {
    // final linear forward + MSE (loss, rec) + final linear backward + Gabor gradient of layer L:
    // one pass over out_L / lin_L instead of three
    ProfScope ps(s, 3, 0);
    const float* linL = p.kind == WIRE_KIND_RELU ? nullptr : A + a.lin1 + (int64_t)(p.L - 1) * a.np * p.Pl;
    HIPCHK(launch_final_fused(s, p.kind, recomp ? nullptr : A + a.out0 + (int64_t)p.L * a.np * p.P, linL, n, p.P, p.O,
                              p.K, packed + p.off_wf, packed + p.off_bf, target, idx, first, weight,
                              p.w, p.s, y, rec, gL, Sx + sc.fpw, Sx + sc.fpb, Sx + sc.crp, loss_out,
                              x2 ? gamax + p.L * WIRE_AMAX_SLOTS : nullptr));
  }
  return mlp_bwd_core(stream, p, packed, coords, n, nullptr, act, act_bytes, scratch, scratch_bytes, grads, false, ready,
                      user);
}
extern "C" int wire_train_fwd_bwd(void* stream, const wire_net_desc* d, const float* packed,
                                  const float* coords, int64_t n, const float* target, const int64_t* idx,
                                  int64_t first, float weight, float* y, float* g_y, float* loss_out,
                                  float* rec, float* partial, void* act, int64_t act_bytes, void* scratch,
                                  int64_t scratch_bytes, void* const* grads) {
  return wire_train_fwd_bwd_hooked(stream, d, packed, coords, n, target, idx, first, weight, y, g_y, loss_out, rec, partial,
                                   act, act_bytes, scratch, scratch_bytes, grads, nullptr, nullptr);
}

// ---------------------------------------------------------------------------
// training glue
// ---------------------------------------------------------------------------
extern "C" int wire_coords_from_index(void* stream, const int64_t* idx, int64_t first, int64_t n,
                                      const float* tx, int W, const float* ty, int H,
                                      const float* tz, int T, float* coords) {
  if (n < 0 || !tx || !ty || !coords || W < 1 || H < 1 || (tz && T < 1))
    return fail(WIRE_ERR_ARG, "bad argument to wire_coords_from_index");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_coords((hipStream_t)stream, idx, first, n, tx, W, ty, H, tz, T, coords));
  return WIRE_OK;
}
extern "C" int wire_perm_indices(void* stream, uint64_t seed, int64_t n_total, int64_t first, int64_t count,
                                 int64_t* idx_out) {
  if (n_total < 1 || first < 0 || count < 0 || first + count > n_total || (count > 0 && !idx_out))
    return fail(WIRE_ERR_ARG, "bad argument to wire_perm_indices");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_perm_indices((hipStream_t)stream, seed, n_total, first, count, idx_out));
  return WIRE_OK;
}
extern "C" int wire_mse_grad(void* stream, const float* y, const float* target, const int64_t* idx,
                             int64_t first, int64_t n, int O, float weight, float* g_y,
                             float* loss_out, float* rec, float* partial) {
  if (n < 0 || O < 1 || !y || !target || !g_y || !loss_out || !partial)
    return fail(WIRE_ERR_ARG, "bad argument to wire_mse_grad");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_mse_grad((hipStream_t)stream, y, target, idx, first, n, O, weight, g_y, loss_out,
                         rec, partial));
  return WIRE_OK;
}
extern "C" int wire_avgpool_mse_grad(void* stream, const float* y, int H, int W, int O, int scale,
                                     const float* gt_lr, float* g_y, float* rec_lr, float* loss_out,
                                     float* partial) {
  if (H < 1 || W < 1 || O < 1 || scale < 1 || scale > H || scale > W || !y || !gt_lr || !g_y || !loss_out || !partial)
    return fail(WIRE_ERR_ARG, "bad argument to wire_avgpool_mse_grad");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_avgpool_mse_grad((hipStream_t)stream, y, H, W, O, scale, gt_lr, g_y, rec_lr, loss_out, partial));
  return WIRE_OK;
}
extern "C" int wire_adam_step_flat(void* stream, float* param, const float* grad, float* exp_avg,
                                   float* exp_avg_sq, int64_t count, float lr, float beta1,
                                   float beta2, float eps, int64_t step) {
  if (count < 0 || step < 1 || !param || !grad || !exp_avg || !exp_avg_sq)
    return fail(WIRE_ERR_ARG, "bad argument to wire_adam_step_flat");
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_adam((hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, count,
                     (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / std::sqrt(bc2))));
  return WIRE_OK;
}

extern "C" int wire_eval_metric(void* stream, int mode, const float* rec, const float* gt, int64_t count,
                                float thres, float* out2, float* partial) {
  if ((mode != 0 && mode != 1) || count < 1 || !rec || !gt || !out2 || !partial)
    return fail(WIRE_ERR_ARG, "bad argument to wire_eval_metric");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_metric((hipStream_t)stream, mode, rec, gt, count, thres, out2, partial));
  return WIRE_OK;
}

extern "C" int wire_track_best(void* stream, const float* metric, float* best_metric, int force, const float* src,
                               float* dst, int64_t count, int* updated) {
  if (!metric || !best_metric || count < 0 || (count > 0 && (!src || !dst)))
    return fail(WIRE_ERR_ARG, "bad argument to wire_track_best");
  if (count > 0 && (((uintptr_t)src | (uintptr_t)dst) & 15)) return fail(WIRE_ERR_ARG, "src / dst must be 16-byte aligned");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_track_best((hipStream_t)stream, metric, best_metric, force, src, dst, count, updated));
  return WIRE_OK;
}
extern "C" int wire_radon_fwd(void* stream, const float* img, const float* angles_deg, int H, int W, int nangles,
                              float* sino) {
  if (H < 1 || W < 1 || nangles < 1 || !img || !angles_deg || !sino) return fail(WIRE_ERR_ARG, "bad argument to wire_radon_fwd");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_radon_fwd((hipStream_t)stream, img, angles_deg, H, W, nangles, sino));
  return WIRE_OK;
}
extern "C" int wire_radon_bwd(void* stream, const float* g_sino, const float* angles_deg, int H, int W, int nangles,
                              float* g_img) {
  if (H < 1 || W < 1 || nangles < 1 || !g_sino || !angles_deg || !g_img) return fail(WIRE_ERR_ARG, "bad argument to wire_radon_bwd");
  ProfScope ps((hipStream_t)stream, 3, 0);
  HIPCHK(launch_radon_bwd((hipStream_t)stream, g_sino, angles_deg, H, W, nangles, g_img));
  return WIRE_OK;
}
extern "C" int wire_posenc_fwd(void* stream, const float* coords, int64_t n, int D, int F, float* out) {
  if (n < 0 || D < 1 || D > 4 || F < 0 || F > 30 || (n > 0 && (!coords || !out)))
    return fail(WIRE_ERR_ARG, "bad argument to wire_posenc_fwd");
  HIPCHK(launch_posenc((hipStream_t)stream, coords, n, D, F, D + 2 * D * F, out));
  return WIRE_OK;
}
extern "C" int wire_sigmoid_inplace(void* stream, float* x, int64_t count) {
  if (count < 0 || (count > 0 && !x)) return fail(WIRE_ERR_ARG, "bad argument to wire_sigmoid_inplace");
  HIPCHK(launch_sigmoid((hipStream_t)stream, x, count));
  return WIRE_OK;
}

// ---------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------
extern "C" int wire_c64_to_blocked(void* stream, const void* src, int64_t n, int K, float* dst) {
  if (n < 0 || K < 1 || !src || !dst) return fail(WIRE_ERR_ARG, "bad argument");
  HIPCHK(launch_c64_to_blocked((hipStream_t)stream, (const float*)src, n, K, rup(2 * K, 64), dst));
  return WIRE_OK;
}
extern "C" int wire_blocked_to_c64(void* stream, const float* src, int64_t n, int K, void* dst) {
  if (n < 0 || K < 1 || !src || !dst) return fail(WIRE_ERR_ARG, "bad argument");
  HIPCHK(launch_blocked_to_c64((hipStream_t)stream, src, n, K, rup(2 * K, 64), (float*)dst));
  return WIRE_OK;
}
