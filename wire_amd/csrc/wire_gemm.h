// wire_gemm.h -- host-visible declarations of the fp32-MFMA GEMM launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Epilogue selector of the NT GEMM  C[M][Nc] = A[M][Kd] * Bt[Nc][Kd]^T.
enum WireEpi {
  EPI_STORE = 0,            // o0 = C
  EPI_GABOR_FWD = 1,        // C = lin(re|im): o0 = lin (optional), o1 = gabor(lin + bias)
  EPI_GABOR_BWD = 2,        // C = g_out: i0 = lin, i1 = out -> o0 = g_lin
  EPI_GABOR_BWD_FIRST = 3,  // C = g_out0: i1 = out0, coords/W0/b0 -> u; o0 = g_u [M][ldu]
  EPI_SIREN_FWD = 4, EPI_GAUSS_FWD = 5, EPI_RELU_FWD = 6,   // o0 = lin, o1 = act
  EPI_SIREN_BWD = 7, EPI_GAUSS_BWD = 8, EPI_RELU_BWD = 9,   // i0 = lin, i1 = out -> o0 = g_lin
  EPI_GABOR2D_FWD = 10,     // C = (lin|sy)(re|im) 128-col groups: o0 = linsy [M][2P], o1 = out [M][P]
  EPI_GABOR2D_BWD = 11,     // C = g_out: i0 = linsy, i1 = out -> o0 = g_linsy [M][2P]
  EPI_GABOR2D_BWD_FIRST = 12  // real first layer of wire2d: o0 = g_(u|p) [M][2*ldu]
};

struct GemmEpiParams {
  const float* bias = nullptr;   // [Nc], GEMM column order
  float* o0 = nullptr;
  float* o1 = nullptr;
  const float* i0 = nullptr;
  const float* i1 = nullptr;
  int ld0 = 0;                   // leading dim (floats) of o0 / i0
  int ld1 = 0;                   // leading dim of o1 / i1
  float omega = 0.f, scale = 0.f;
  int kvalid = 0;                // valid features (complex count for Gabor, real count otherwise)
  // first-layer backward extras
  const float* coords = nullptr; // [M][D]
  const float* W0 = nullptr;     // [K][D] (wire2d: followed by V0 via W0b)
  const float* b0 = nullptr;     // [K]
  const float* W0b = nullptr;    // wire2d scale_orth weight [K][D]
  const float* b0b = nullptr;
  int D = 0;
  int ldu = 0;
  int wide = 0;                  // set by the launcher when 32-bit byte offsets could overflow
  float* cr_partial = nullptr;   // EPI_GABOR_BWD_FIRST of wire_gemmx3h.hip: per-256-row-tile sums [tile][cr_C][5] of
  int cr_C = 0;                  //   g_u [x | 1] (the first layer's weight / bias gradient) instead of storing g_u
  int64_t cr_set = 0;            //   wire2d: floats between the sums of the layer's two Linears (g_u, g_p)
  int recompute_out = 0;         // EPI_GABOR_BWD of wire_gemmx3h.hip: out = act(lin) again instead of reading i1
  int stagger = 0, stagger_lo = 0, stagger_hi = 0;   // wire_gemmx3g.hip: late start (100 MHz ticks) of blocks [lo, hi)
  // 2 x fp16 split GEMM (wire_gemmx2h.hip): sharded max-|value| slots (WIRE_AMAX_SLOTS unsigned each, wire_dev.h)
  const unsigned* amax_a = nullptr;   //   of the activation operand A (filled by A's producer)
  const unsigned* amax_b = nullptr;   //   of the weight image (filled by launch_amax over the fp32 image)
  unsigned* amax_out = nullptr;       //   of the tensor this epilogue writes for the next GEMM (out / g_lin); may be null
  // pre-split activations (wire_dev.h: wire_store_out4), 2 x fp16 kernels only:
  float o1_split = 0.f;               //   != 0: the forward epilogue stores o1 (out) split with this power-of-two scale
  float a_split_inv = 0.f;            //   != 0: A is such a pre-split tensor; 1 / its scale (amax_a is not read)
#ifdef WIRE_ABLATE
  int ablate = 0;                // tools/gemm_tune only: 1 no global loads, 2 no LDS writes, 4 no barrier
#endif
};

// C = A * Bt^T with fused epilogue.  A [M][lda], Bt [Nc][ldb]; Nc % 64 == 0,
// Kd % 32 == 0, lda/ldb % 4 == 0.
hipError_t launch_gemm_nt(hipStream_t s, int epi, const float* A, int lda, const float* Bt,
                          int ldb, int64_t M, int Nc, int Kd, const GemmEpiParams& ep);

int gemm_tune_set(const char* key, int value);

// Split-over-rows TN GEMM  slab[s][Pm][Pn] = G[rows_s][Pm]^T * Z[rows_s][Pn]
// (+ optional column sums of G into bslab[s][Pm]).  Returns the number of
// row splits used through *splits (<= max_splits).
int gemm_tn_splits(int64_t n, int Pm, int Pn, int max_splits);
hipError_t launch_gemm_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz,
                          int64_t n, int Pm, int Pn, int splits, float* slab, float* bslab);

// ---- 3-multiplication complex GEMMs (wire_gemm3m.hip).  Kp_* = padded complex widths (P/2).
// NT: C[M][Kp_out] = A[M][Kp_in] * B[Kp_out][Kp_in]^T, all blocked-planar complex rows.
hipError_t launch_gemm3m_nt(hipStream_t s, int epi, const float* A, int lda, const float* B, int ldb,
                            int64_t M, int Kp_out, int Kp_in, const GemmEpiParams& ep);
int gemm3m_tune_set(const char* key, int value);
// TN: slab[split][3][Kp_o][Kp_i] planes (P1,P2,P3), bslab[split][2][Kp_o] column sums of G (re, im)
int gemm3m_tn_splits(int64_t n, int Kp_o, int Kp_i, int max_splits);
hipError_t launch_gemm3m_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                            int Kp_o, int Kp_i, int splits, float* slab, float* bslab);

// ---- split-bf16 GEMMs (wire_gemmx3.hip): fp32 operands split exactly into 3 bf16 terms, 6 partial
// products on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Same operand / epilogue / slab conventions as
// launch_gemm_nt / launch_gemm_tn, except that the weight image is the pre-split stage-major image that
// launch_x3_split_b writes from the fp32 image Bt[Nc][ldb] (gemmx3_b_image_floats(Nc, Kd) floats).
int64_t gemmx3_b_image_floats(int Nc, int Kd);
hipError_t launch_x3_split_b(hipStream_t s, const float* Bt, int ldb, int Nc, int Kd, void* Bx3);
#define X3_SPLIT_MAXB 32
struct X3SplitBatch { const float* src[X3_SPLIT_MAXB]; void* dst[X3_SPLIT_MAXB]; };
hipError_t launch_x3_split_b_batch(hipStream_t s, const X3SplitBatch& sb, int nb, int ldb, int Nc, int Kd);
hipError_t launch_gemmx3_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx3, int64_t M,
                            int Nc, int Kd, const GemmEpiParams& ep);
int gemmx3_tune_set(const char* key, int value);
bool gemmx3_nt_is_h16(int epi, int64_t M);
int gemmx3_tn16_mode();
int gemmx3_tn_splits(int64_t n, int Pm, int Pn, int max_splits);
int gemmx3_tn_splits_max(int64_t n, int Pm, int Pn, int max_splits);   // over the "x3_tn16" settings (scratch sizing)
hipError_t launch_gemmx3_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                            int Pm, int Pn, int splits, float* slab, float* bslab);

// ---- LDS-DMA 32 x 32 x 16 edition of the split-bf16 NT GEMM (tools/wire_gemmx3g.hip, harness builds only; it registers
// itself with launch_gemmx3_nt's dispatch); same operands and image
void gemmx3_register_glds(bool (*handles)(int, int64_t),
                          hipError_t (*launch)(hipStream_t, int, const float*, int, const void*, int64_t, int, int,
                                               const GemmEpiParams&),
                          int (*tune)(const char*, int));
bool gemmx3g_handles(int epi, int64_t M);
int gemmx3g_tune_set(const char* key, int value);
int gemmx3g_mode();
hipError_t launch_gemmx3g_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx3, int64_t M,
                             int Nc, int Kd, const GemmEpiParams& ep);

// ---- 16 x 16 x 32 edition (wire_gemmx3h.hip): reads the second (unswizzled) half of the split image
bool gemmx3h_handles(int epi, int64_t M);
int gemmx3h_tune_set(const char* key, int value);
int gemmx3h_mode();
hipError_t launch_gemmx3h_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx3, int64_t M,
                             int Nc, int Kd, const GemmEpiParams& ep);

// ---- 2 x fp16 split GEMMs on v_mfma_f32_16x16x32_f16 (wire_gemmx2h.hip): 3 partial products per fp32 product, operands
// scaled by powers of two taken from their producers' max-|value| slots (WIRE_AMAX_SLOTS unsigned per tensor, wire_dev.h)
#define X2_AMAX_MAXB 32
struct X2AmaxBatch { const float* src[X2_AMAX_MAXB]; unsigned* slots[X2_AMAX_MAXB]; };
hipError_t launch_amax_batch(hipStream_t s, const X2AmaxBatch& ab, int nb, int64_t count);
hipError_t launch_amax(hipStream_t s, const float* src, int64_t count, unsigned* slots);
int64_t gemmx2_b_image_floats(int Nc, int Kd);
#define X2_SPLIT_MAXB 32
struct X2SplitBatch { const float* src[X2_SPLIT_MAXB]; void* dst[X2_SPLIT_MAXB]; const unsigned* slots[X2_SPLIT_MAXB]; };
hipError_t launch_x2_split_b_batch(hipStream_t s, const X2SplitBatch& sb, int nb, int ldb, int Nc, int Kd);
hipError_t launch_gemmx2h_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx2, int64_t M, int Nc,
                             int Kd, const GemmEpiParams& ep);
int gemmx2h_tune_set(const char* key, int value);
int gemmx2h_tune_get(const char* key);     // "x2_amode" -> value; -1 = unknown key
bool gemmx2_tn_applies(int Pm, int Pn);
int gemmx2_tn_splits(int64_t n, int Pm, int Pn, int max_splits);
hipError_t launch_gemmx2_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n, int Pm,
                            int Pn, int splits, float* slab, float* bslab, const unsigned* amax_g,
                            const unsigned* amax_z, float z_pre_inv = 0.f, int z_act = 0, int batch = 1,
                            long long g_step = 0, long long z_step = 0, int amax_step = 0);
int gemmx2_tn_batch_splits(int64_t n, int Pm, int Pn, int max_splits, int batch);

// ---- whole-net forward in one kernel (wire_fused.hip): activations stay in the wave's registers from the coordinates to
// the output; the hidden layers' weights come from a k-permuted edition of the 2 x fp16 image (same maximum slots)
#define FX_SPLIT_MAXB 32
struct FxSplitBatch { const float* src[FX_SPLIT_MAXB]; void* dst[FX_SPLIT_MAXB]; const unsigned* slots[FX_SPLIT_MAXB]; };
int64_t fused_b_image_floats(int P);                      // floats of one hidden layer's image
// c: the constant folded into the weights (fused_pre_scale of the net kind and its omega_0 / scale_0)
hipError_t launch_fx_split_b_batch(hipStream_t s, const FxSplitBatch& sb, int nb, int ldb, int P, float c, int Kd = 0);
struct FusedFwdParams {
  const float* coords = nullptr; long long n = 0;         // [n][D]
  const float* W0 = nullptr; const float* b0 = nullptr;   // first layer, native [K][D], [K]
  const unsigned char* wimg = nullptr;                    // images of layers 1 .. L back to back (fused_b_image_floats each)
  const float* bias = nullptr; long long bias_stride = 0; // blocked bias of layer l at bias + (l - 1) * bias_stride, [P]
  const unsigned* wamax = nullptr; int wamax_stride = 0;  // max |W_l| slots of layer l at wamax + (l - 1) * wamax_stride
  const float* wf = nullptr; const float* bfr = nullptr;  // final linear image [O][P] and bias [O]
  float* y = nullptr;                                     // [n][O]
  int D = 0, K = 0, L = 0, O = 0;
  float w1 = 0.f, w = 0.f, s = 0.f;                       // first omega_0, hidden omega_0, scale_0
  // constants folded into the pre-activations: r = c lin with c = omega_0 / 2 pi (sine, Gabor), scale_0 sqrt(log2 e)
  // (Gaussian), 1 (relu) -- c_first for layer 0 (first omega_0; applied to W0, b0 on the fly), c_hidden in the hidden
  // layers' images and biases; Gabor envelope: k2 = scale_0^2 log2 e / c^2
  float c_first = 1.f, c_hidden = 1.f, k2_first = 0.f, k2 = 0.f;
  // training forward (out != null): what the backward reads is stored on the way -- rows of P floats, buffers padded to a
  // multiple of 128 rows (every store is unconditional)
  float* lin0 = nullptr;                                  // lin_0 [n][P] (sine, Gaussian)
  float* lin = nullptr; long long lin_stride = 0;         // lin_l, l = 1 .. L, at lin + (l - 1) * lin_stride (not relu)
  float* out = nullptr; long long out_stride = 0;         // out_l at out + l * out_stride: l = 0 fp32; 1 .. L - 1 pre-split
                                                          // pairs at scale 1 (relu: fp32, and out_L too)
  unsigned* amax_out = nullptr;                           // max |out_l| slots at amax_out + l * WIRE_AMAX_SLOTS (fp32-stored layers)
  float inv_c_first = 1.f, inv_c_hidden = 1.f;            // 1 / c: lin is stored in the reference's units
  // positional-encoding nets (relu, modules/relu.py:62-75): pe_F > 0 frequencies -- layer 0 is a GEMM over the 64 padded
  // encoded features, its image (P x 64, same format) lies in front of the hidden layers' at wimg, W0 / b0 are not read
  int pe_F = 0; const float* bias0 = nullptr; const unsigned* wamax0 = nullptr;
  int rstore = 0;                                         // sine / Gaussian nets with the data-gradient chain: lin_0 .. lin_{L-1} are
                                                          // stored AS r = c lin (lin0, lin) and out_0 .. out_{L-1} not at all -- the
                                                          // chain and the weight-gradient loader (launch_gemmx2_tn, z_act) evaluate on r
  // the final stage inside the training forward (real nets; target != null): from the accumulators of layer L the wave forms
  // y = h_L W_f^T + b_f, the MSE terms against target[src] (src = idx ? idx[row] : first + row), dL/dy = gscale (y - t),
  // g_lin_L = (dL/dy W_f) act'(lin_L) -- stored, its maximum published -- and per workgroup the sums dL/dy^T h_L, sum dL/dy,
  // sum (y - t)^2 (what final_fused_kernel of wire_point.hip does from the stored lin_L / out_L, which are then not written)
  const float* target = nullptr; const int64_t* idx = nullptr; long long first = 0; float gscale = 0.f;
  float* rec = nullptr;                                   // optional scatter of y to rec[src]
  float* g_lin = nullptr;                                 // g_lin_L rows of P floats (padded to 128 rows)
  float* part_w = nullptr; float* part_b = nullptr;       // [block][O][P], [block][O]: launch_final_reduce adds them up
  float* loss_partial = nullptr;                          // [block]
  unsigned* amax_g = nullptr;                             // max |g_lin_L| slots
};
bool fused_final_enabled();
bool fused_train_enabled();
float fused_pre_scale(int kind, float omega0, float scale0);   // the c of a layer with these hyper-parameters
bool fused_fwd_shape(int kind, int P);
bool fused_fwd_enabled();
int fused_tune_get(const char* key);                      // "fused_fwd"; -1 = unknown key
int fused_tune_set(const char* key, int value);
hipError_t launch_fused_fwd(hipStream_t s, int kind, int P, const FusedFwdParams& fp);

// ---- the data-gradient chain of the real nets in one kernel (wire_fused.hip: fused_bwd_kernel): g_lin_L .. g_lin_1
struct FusedBwdParams {
  long long n = 0;
  float* g = nullptr; long long g_stride = 0;             // g_lin_l at g + l * g_stride, rows of P floats, l = 1 .. L: g_lin_L is
                                                          // read, g_lin_{L-1} .. g_lin_1 are written (buffers padded to 128 rows)
  unsigned* gamax = nullptr;                              // max |g_lin_l| slots at gamax + l * WIRE_AMAX_SLOTS (L: read; below: written)
  const float* aux = nullptr; long long aux_stride = 0;   // lin_l (rstore: r_l; relu: out_l) at aux + l * aux_stride, l = 1 .. L - 1
  const float* aux0 = nullptr;                            // lin_0 (rstore: r_0 = c_first lin_0; relu: out_0), rows of P floats
  const unsigned char* wimg = nullptr;                    // k-permuted images of the TRANSPOSED weights, layers L, L - 1, .. 1 back to back
  const unsigned* wamax = nullptr; int wamax_stride = 0;  // max |W_l| slots of layer l at wamax + (l - 1) * wamax_stride
  int L = 0;
  float w = 0.f, s = 0.f;                                 // hidden omega_0, scale_0
  int rstore = 0; float c_hidden = 1.f;                   // aux holds r = c lin (FusedFwdParams::rstore), c
  // the last link (layer 1): g_lin_0 = g_out_0 act'(lin_0) with the FIRST layer's omega_0.  crp != null (native first layer):
  // g_lin_0 is not stored, its sums with [x | 1] per workgroup go to crp[blockIdx][C][5] (launch_colreduce_final_blocks adds
  // them up: the first layer's weight and bias gradient); crp == null (positional encoding): g_lin_0 is stored at g + 0
  float w1 = 0.f;
  const float* coords = nullptr; int D = 0;
  float* crp = nullptr; int C = 0;
};
bool fused_bwd_enabled();
bool fused_bwd_shape(int kind, int P);
hipError_t launch_fused_bwd(hipStream_t s, int kind, int P, const FusedBwdParams& fp, int* tile_rows);
