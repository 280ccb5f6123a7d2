// wire_fused.hip -- the whole net forward as ONE kernel: a wave carries its 16 rows from the coordinates to the output, the
// activations never leave its registers.
//
// Replaces `self.net(coords)` of modules/siren.py:90-96, gauss.py:71-74, relu.py:124-130 and modules/wire.py:161-165 for the
// widths whose row fits a wave's registers twice (padded width P <= 384 floats): the layer-by-layer path writes and reads every
// out_l through HBM (2 P floats per sample and layer), this one moves D + O floats per sample (the fused
// algorithmic minimum) and streams the weights from the L2.
//
// How a row stays in registers across layers.  The 16 x 16 x 32 MFMAs are issued with the operands swapped (weights first), so
// a 16-column output block leaves the accumulator as  lane (r = lane & 15, g = lane >> 4), register q  ->  row r, column
// 4 g + q  (wire_gemmh_epi.h).  The NEXT layer's activation fragment of a 32-deep stage wants, in the same lane, 8 reduction
// indices of row r.  The reduction order inside an MFMA is free as long as both operands agree on it -- so the stage's k
// slots are DEFINED as what the lane already holds: slot (g, e) of stage j = column 32 j + 4 g + e (e < 4) or
// 32 j + 16 + 4 g + (e - 4), i.e. the lane's registers of blocks 2 j and 2 j + 1 -- and the weight image is stored with its
// k indices in that order (fx_split_b_kernel).  After bias + activation + fp16 split the 8 accumulator registers of two
// blocks ARE the two fragments (h, l) of one stage: no LDS round trip, no cross-lane traffic, no extra registers.
//
// Structure.  Workgroup = 8 waves x 16 rows = 128 rows; NB = P / 16 column blocks per wave (64 accumulator registers at
// P = 256, 96 at P = 384) in two sets that alternate between layers; the weights of all hidden layers are one linear stream of
// (layer, stage) pieces of NB KB x 2 planes, brought into a ring of LDS stage buffers by LDS-DMA (all 8 waves share a
// stage: one barrier per stage) and read as lane * 16-byte fragments.  While the MFMAs of stage j run, the vector unit
// prepares stage j + 1's fragment from the previous layer's accumulators (bias, activation, split): the epilogue of layer l
// lies under the matrix work of layer l + 1 inside ONE wave, on top of the overlap between the two waves of a SIMD.
// Layer 0 (D <= 4 inputs) is the same producer with the coordinates as its source; the final linear layer is a dot product
// per lane + two cross-lane adds.
//
// The file's kernels:
//   fused_fwd_kernel<KIND, NB, RING, ABL, TRAIN, PE>   the forward: TRAIN = 0 forward-only (coordinates in, y out); 1 / 2 the
//       training forward, storing what the backward reads on the way (2: sine / Gaussian nets behind the chain store r = c lin
//       of every layer and no out at all); PE: positional-encoding nets, layer 0 as a GEMM layer on features evaluated in
//       the lanes; its tails: fx_final (y), fx_tail_train (lin_L), fx_tail_loss (the final stage inside: knob "fused_final")
//   fused_bwd_kernel<KIND, NB, RING, W, ABL>           the data-gradient chain of the real nets g_lin_L -> ... -> g_lin_1 ->
//       first-layer gradient sums, the same register chaining with the transposed weight images
//   fx_split_b_kernel                                  the k-permuted, pre-scaled 2 x fp16 weight images of both
// (ABL != 0: timing probes, instantiated in harness builds only -- EXTRA=-DWIRE_FX_ABLATE, tools/fused_ablate.py,
//  tools/fused_bwd_ablate.py.)
//
// Arithmetic = the 2 x fp16 split of wire_gemmx2h.hip (three partial products per fp32 product, fp32 accumulate, power-of-two
// operand scales: the weights' from their maximum slots, the activations' from their a-priori bound or the wave's own
// maximum), hardware transcendentals on pre-scaled arguments for every layer (below: "activations from pre-scaled ...").
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"
#include "wire_point.h"

typedef _Float16 fx_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fx_u32x4 __attribute__((ext_vector_type(4)));

#define FX_WAVES 8
#define FX_ROWS (16 * FX_WAVES)
#define FX_LMAX 8                         // hidden layers whose biases fit the LDS table
#define FX_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#ifndef FX_PFD
#define FX_PFD 3                          // column blocks by which the weight-fragment LDS reads run ahead of their MFMAs
#endif

// LDS-DMA of 16 bytes per lane (1 KB per wave instruction) -- as inline assembly, not __builtin_amdgcn_global_load_lds:
// with the builtin in a loop the compiler's wait-count pass treats every later LDS read as possibly completing out of order
// and emits `s_waitcnt lgkmcnt(0)` for ALL of them (measured on a 20-line kernel: 14 x lgkmcnt(2) without the builtin, 7 x
// lgkmcnt(0) with it), which turns a fragment prefetch of two blocks into none.  The stream is synchronised by hand anyway
// (fx_stage_top: vmcnt + barrier).  M0 (the LDS base of the DMA) is saved and restored around it.
WIRE_DEVINL void fx_dma16(const void* gsrc, unsigned char* lds_piece) {
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_piece;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(la) : "memory");
}

// ---- activations from PRE-SCALED pre-activations.  The ablations of the first edition (profiles/r04_fused_ablation.txt)
// put a third of the kernel's time into the producer's vector instructions (they do not hide under the MFMAs: the chip
// runs this loop at its power limit), none into the LDS reads.  So the producer is cut to the bone:
//   * the constant that the activation multiplies its argument with is folded into the weights and the bias when the
//     image is packed: r = c lin with c = omega0 / 2 pi (sine, Gabor: v_sin / v_cos take revolutions), c = sigma0
//     sqrt(log2 e) (Gaussian: exp(-(s lin)^2) = exp2(-r^2)), c = 1 (relu).  The accumulation error of r is the same
//     RELATIVE error as that of lin (same products, same sums, scaled by c), i.e. the same absolute error in the argument of
//     the sine as the layer-by-layer path's fl(omega0 lin) carries;
//   * activations bounded by 16 (everything but relu) are split into (h, l) WITHOUT a power-of-two scale: h = fp16(x),
//     l = fp16(x - h) is exact to 2^-25 absolutely (fp16 subnormals), i.e. to fp32's own epsilon relative to the bound --
//     two multiplications per pair less.
// sine: fma, fract, sin + 2 for the split = 5 vector instructions per element (12 before).
template <int ACT>
WIRE_DEVINL float fx_act(float r) {
  if (ACT == ACT_SIREN) return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
  if (ACT == ACT_GAUSS) return __builtin_amdgcn_exp2f(-(r * r));
  return r > 0.f ? r : 0.f;
}
// Gabor exp(j w u - w v - s^2 (u^2 + v^2)) from (u', v') = c (u, v), c = w / 2 pi:  k1 = 2 pi log2 e, k2 = s^2 log2 e / c^2
WIRE_DEVINL void fx_gabor(float up, float vp, float k2, float& o_re, float& o_im) {
  const float q = __builtin_fmaf(up, up, vp * vp);
  const float t = __builtin_fmaf(-k2, q, -9.06472028f * vp);
  const float e = __builtin_amdgcn_exp2f(t);
  const float fr = __builtin_amdgcn_fractf(up);
  o_re = e * __builtin_amdgcn_cosf(fr);
  o_im = e * __builtin_amdgcn_sinf(fr);
}
// (x0, x1) = H + L in packed fp16 pairs, no scale: 4 vector instructions
WIRE_DEVINL void fx_split2(float x0, float x1, unsigned& H, unsigned& L) {
  H = x2_cvt_pk(x0, x1);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(H));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(H));
  L = x2_cvt_pk(r0, r1);
}
// the split of a pair: unscaled for the bounded activations, with the wave's scale for relu
template <int KIND>
WIRE_DEVINL void fx_split_pair(float x0, float x1, float a_scale, unsigned& H, unsigned& L) {
  if constexpr (KIND == NK_RELU) x2_split2(x0, x1, a_scale, H, L);
  else fx_split2(x0, x1, H, L);
}

// ---------------------------------------------------------------------------
// weight image of one hidden layer for the fused kernel, from the fp32 forward image Bt[P][ldb] (row = output column):
//   Fx[stage kt (32 k)][plane (h, l)][16-column block cb][k slot ks (4)][column (16)][8 halves]
// with the k indices of a stage permuted as the header describes: slot (ks, e) holds k = 32 kt + 4 ks + e (e < 4) or
// 32 kt + 16 + 4 ks + (e - 4).  Scaled by the layer's power-of-two scale (its maximum slots), split into (h, l).
// ---------------------------------------------------------------------------
// power-of-two scale of a weight image whose entries are c W: from max |W| (the layer's slots) times c -- evaluated the
// same way where the image is written and where the kernel undoes it
WIRE_DEVINL void fx_weight_scales(const unsigned* slots, int lane, float c, float& s, float& inv) {
  const float m = __uint_as_float(wire_amax_read(slots, lane)) * c;
  wire_x2_scales(__float_as_uint(m), s, inv);
}
__global__ void fx_split_b_kernel(FxSplitBatch sb, int ldb, int P, int Kd, float c) {
  const float* __restrict__ Bt = sb.src[blockIdx.z];
  unsigned short* __restrict__ Fx = (unsigned short*)sb.dst[blockIdx.z];
  float s, inv;
  fx_weight_scales(sb.slots[blockIdx.z], threadIdx.x & 63, c, s, inv);
  const int k2 = (blockIdx.x * blockDim.x + threadIdx.x) * 2;   // pair of reduction indices (same group of four)
  const int j = blockIdx.y;                                     // output column (P of them; Kd reduction indices)
  if (k2 >= Kd) return;
  const float x0 = c * Bt[(size_t)j * ldb + k2], x1 = c * Bt[(size_t)j * ldb + k2 + 1];
  unsigned H, L;
  x2_split2(x0, x1, s, H, L);
  const int NB = P >> 4;
  const int kt = k2 >> 5, kk = k2 & 31;
  const int ks = (kk & 15) >> 2, e = (kk & 3) + ((kk >> 4) << 2);
  const int cb = j >> 4, col = j & 15;
  const size_t base = ((size_t)(kt * 2) * NB + cb) * 512 + (ks * 16 + col) * 8 + e;
  *reinterpret_cast<unsigned*>(Fx + base) = H;
  *reinterpret_cast<unsigned*>(Fx + base + (size_t)NB * 512) = L;
}
int64_t fused_b_image_floats(int P) { return (int64_t)P * P; }   // 2 planes of fp16 = 4 bytes per weight
hipError_t launch_fx_split_b_batch(hipStream_t s, const FxSplitBatch& sb, int nb, int ldb, int P, float c, int Kd) {
  if (Kd <= 0) Kd = P;                                     // square hidden layers by default
  if ((P & 15) || (Kd & 31) || nb < 1 || nb > FX_SPLIT_MAXB || !(c > 0.f)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((Kd / 2 + 127) / 128), (unsigned)P, (unsigned)nb);
  hipLaunchKernelGGL(fx_split_b_kernel, grid, dim3(128), 0, s, sb, ldb, P, Kd, c);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
template <int KIND>
struct FxKind {
  static constexpr bool CPLX = KIND == NK_WIRE;
  static constexpr int ACT = KIND == NK_SIREN ? ACT_SIREN : (KIND == NK_GAUSS ? ACT_GAUSS : ACT_RELU);
  static constexpr int U = CPLX ? 4 : 2;          // column blocks of one epilogue unit (complex: re | im of 32 features)
};

// per-wave state that the layer function threads through
struct FxCtx {
  int lane, wave, g;
  float x[4];                 // the row's coordinates
  long long row, n;           // this lane's row (not clamped), number of rows
  unsigned char* ring;        // LDS stage buffers
  const float* sW0;           // LDS: first layer weights, transposed [4][PF], zero-padded
  const float* sb0;           // LDS: first layer bias [PF]
  const float* sbias;         // LDS: hidden biases [L][P] (GEMM column order)
  const float* swf;           // LDS: final image [4][P]
  const long long* ssrc;      // LDS: source index of the workgroup's rows (final stage inside the training forward)
  int t, T, buf;              // weight stream: next stage to consume, number of stages, its ring buffer
  float inv_prev;             // 1 / (s_A s_B) of the accumulators in the source set
  // training forward: where the layer in production (l_src) is stored -- row pointers of THIS lane (row * P applied)
  float* st_lin;              // lin_{l_src} [n][P] fp32, null = not stored
  float* st_out;              // out_{l_src} [n][P]: fp32 (st_split = false) or pre-split pairs (wire_dev.h: wire_store_out4, scale 1)
  bool st_split;
  float st_inv_c;             // 1 / c: lin is stored in the reference's units
  float amx;                  // max |out| of this wave's fp32-stored activations (the consumer GEMM's operand scale)
  int ys;                     // vector-memory operations issued since the last weight-stream issue (stores: counted waits)
};

template <int NB, int RING, typename PT, int W = FX_WAVES>
WIRE_DEVINL void fx_issue(const PT& fp, const FxCtx& c, int t, int buf) {
  constexpr int STAGE = NB * 2048, PIECES = 2 * NB / W;   // 1 KB pieces per wave: the W waves of a workgroup share a stage
  const unsigned char* src = fp.wimg + (size_t)t * STAGE + c.wave * 1024 + c.lane * 16;
  unsigned char* dst = c.ring + buf * STAGE + c.wave * 1024;
#pragma unroll
  for (int j = 0; j < PIECES; ++j) fx_dma16(src + j * (W * 1024), dst + j * (W * 1024));
}

// top of a stage: this wave's pieces of stage c.t have landed, then everybody's (barrier); the buffer that the barrier
// proves free (every wave has consumed stage c.t - 1) is refilled with stage c.t + RING - 1
// (ABL: timing probes of the harness build -- results wrong -- bit 0 no producer, 1 no fragment reads, 2 no weight stream
//  after the prologue, 3 no barrier; 0 in the product)
// s_waitcnt vmcnt(n) for a value that is a constant only after unrolling (the immediate must be a literal)
WIRE_DEVINL void fx_wait_vm(const int n) {
#define FX_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  // (vmcnt has 6 bits; a count above 48 is clamped -- waiting for more than necessary is always safe)
  switch (n < 0 ? 0 : (n > 48 ? 48 : n)) {
    FX_VM_CASE(0) FX_VM_CASE(1) FX_VM_CASE(2) FX_VM_CASE(3) FX_VM_CASE(4) FX_VM_CASE(5) FX_VM_CASE(6) FX_VM_CASE(7)
    FX_VM_CASE(8) FX_VM_CASE(9) FX_VM_CASE(10) FX_VM_CASE(11) FX_VM_CASE(12) FX_VM_CASE(13) FX_VM_CASE(14) FX_VM_CASE(15)
    FX_VM_CASE(16) FX_VM_CASE(17) FX_VM_CASE(18) FX_VM_CASE(19) FX_VM_CASE(20) FX_VM_CASE(21) FX_VM_CASE(22) FX_VM_CASE(23)
    FX_VM_CASE(24) FX_VM_CASE(25) FX_VM_CASE(26) FX_VM_CASE(27) FX_VM_CASE(28) FX_VM_CASE(29) FX_VM_CASE(30) FX_VM_CASE(31)
    FX_VM_CASE(32) FX_VM_CASE(33) FX_VM_CASE(34) FX_VM_CASE(35) FX_VM_CASE(36) FX_VM_CASE(37) FX_VM_CASE(38) FX_VM_CASE(39)
    FX_VM_CASE(40) FX_VM_CASE(41) FX_VM_CASE(42) FX_VM_CASE(43) FX_VM_CASE(44) FX_VM_CASE(45) FX_VM_CASE(46) FX_VM_CASE(47)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
#undef FX_VM_CASE
}
// `younger` = vector-memory operations that were CERTAINLY issued after the pieces of stage c.t (the training forward's
// stores and, with RING = 3, the pieces of stage c.t + 1): the wait may leave exactly those in flight.  Never more than were
// issued -- a count that is too high would let a piece of stage c.t stay in flight.
template <int NB, int RING, int ABL, typename PT, int W = FX_WAVES>
WIRE_DEVINL void fx_stage_top(const PT& fp, FxCtx& c, const int younger_stores) {
  constexpr int PIECES = 2 * NB / W;
  if constexpr (!(ABL & 4)) {
    if (RING == 3 && c.t + 1 < c.T) fx_wait_vm(PIECES + younger_stores);
    else fx_wait_vm(younger_stores);
  }
  if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
  if (c.t + RING - 1 < c.T) {
    int nb = c.buf + RING - 1;
    nb = nb >= RING ? nb - RING : nb;
    // (the probe keeps the branch: it is what ends the scheduling region of a stage)
    if constexpr (ABL & 4) asm volatile("s_nop 0");
    else fx_issue<NB, RING, PT, W>(fp, c, c.t + RING - 1, nb);
  }
}

// ---- the producer: one PART = the bias + activation (+ fp16 split) of 4 consecutive columns of this lane's row
//   real nets: part cb = block cb (4 values); stage j consumes parts 2 j and 2 j + 1 as the two halves of its fragment
//   wire:      part q = (G = q >> 1, b = q & 1): the 4 complex features 32 G + 16 b + 4 g + (0..3), i.e. blocks 4 G + b (re) and
//              4 G + 2 + b (im); stage 2 G consumes the re halves of parts 2 G, 2 G + 1, stage 2 G + 1 their im halves
// from the coordinates (layer 0: first_fwd_kernel's arithmetic for u, then the lean activation forms) or from the previous
// layer's accumulators.  Pad features (>= K) are NOT zeroed here: their pre-activation is exactly 0 (zero weights and
// bias), their activation finite, and every weight that multiplies them -- next layer's image, final image -- is 0.
template <int KIND, int NB, bool FROM_COORDS>
WIRE_DEVINL void fx_part(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB], const int q, const int l_src,
                         f32x4& o_re, f32x4& o_im) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB, PF = CPLX ? P / 2 : P;
  if constexpr (CPLX) {
    const int G = q >> 1, b = q & 1;
    f32x4 uu, vv;
    if constexpr (FROM_COORDS) {
      const int f0 = 32 * G + 16 * b + 4 * c.g;
      uu = *reinterpret_cast<const f32x4*>(c.sb0 + f0);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const f32x4 wd = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + f0);
#pragma unroll
        for (int e = 0; e < 4; ++e) uu[e] = __builtin_fmaf(c.x[d], wd[e], uu[e]);
      }
      vv = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      const float* bl = c.sbias + l_src * P + 64 * G + 16 * b + 4 * c.g;
      const f32x4 b_re = *reinterpret_cast<const f32x4*>(bl), b_im = *reinterpret_cast<const f32x4*>(bl + 32);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uu[e] = __builtin_fmaf(src[4 * G + b][e], c.inv_prev, b_re[e]);
        vv[e] = __builtin_fmaf(src[4 * G + 2 + b][e], c.inv_prev, b_im[e]);
      }
    }
    const float k2 = FROM_COORDS ? fp.k2_first : fp.k2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a, bq;
      fx_gabor(uu[e], vv[e], k2, a, bq);
      o_re[e] = a; o_im[e] = bq;
    }
  } else {
    constexpr int ACT = FxKind<KIND>::ACT;
    const int col = 16 * q + 4 * c.g;
    f32x4 lin;
    if constexpr (FROM_COORDS) {
      lin = *reinterpret_cast<const f32x4*>(c.sb0 + col);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const f32x4 wd = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) lin[e] = __builtin_fmaf(c.x[d], wd[e], lin[e]);
      }
    } else {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(c.sbias + l_src * P + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) lin[e] = __builtin_fmaf(src[q][e], c.inv_prev, bv[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o_re[e] = fx_act<ACT>(lin[e]);
    o_im = o_re;
  }
}

template <int KIND, int NB> constexpr int fx_nparts() { return FxKind<KIND>::CPLX ? NB / 2 : NB; }

// ---- the producer in SLICES.  The production of one part is cut into steps -- pre-activation (LDS reads of the bias row),
// one activation per element, the fp16 splits -- and step k of the part(s) a stage produces runs behind the MFMAs of
// column block cb = k * stride of that stage, so that every block of the stage carries three MFMAs, two weight-fragment
// reads (of the next block) and a few vector instructions: the schedule is in the source, block by block.
//   real nets: two parts per stage (q = 2 + 2 j + i, i = cb / (NB / 2)), 7 steps each: lin | act 0..3 | split 01 | split 23
//   wire:      one part per stage (q = 2 + j), 9 steps: lin | gabor 0..3 | split re01 | re23 | im01 | im23
// `w` holds the part in flight: real [lin, out]; wire [u, v, o_re, o_im].
// TRAIN: the part is also STORED for the backward -- lin (in the reference's units) at its pre-activation step, out after
// its last activation (fp32) or after its last split (the very registers of the fragment: wire_store_out4's format at scale
// 1).  Stores are unconditional (rows beyond n land in the padding rows of the act buffer: wire_api.hip act_layout): their
// number per stage is what the counted wait of the weight stream relies on.  Returns the number of stores it issued.
template <int KIND, int NB, bool FROM_COORDS, int TRAIN>
WIRE_DEVINL int fx_slice(const FusedFwdParams& fp, FxCtx& c, const f32x4 (&src)[NB], const int l_src,
                         const float a_scale, const int j, const int cb, f32x4 (&w)[4], fx_u32x4 (&Fh)[NB / 2],
                         fx_u32x4 (&Fl)[NB / 2]) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB, PF = CPLX ? P / 2 : P, NP = fx_nparts<KIND, NB>();
  // what a training forward stores of the layer in production (compile-time: the store count per stage is part of the
  // weight stream's wait): lin unless relu / wire's real layer 0; out fp32 for layer 0 and relu, else the split pairs
  // (TRAIN = 2, sine / Gaussian nets whose backward is the chain: EVERY layer below L stores r itself and no out at all)
  constexpr bool ST_LIN = TRAIN && KIND != NK_RELU && (!FROM_COORDS || !CPLX);
  constexpr bool ST_F32 = TRAIN && (KIND == NK_RELU || (FROM_COORDS && TRAIN != 2));
  constexpr bool ST_SPLIT = TRAIN == 1 && !ST_F32;
  if constexpr (CPLX) {
    constexpr int ST = NB / 9;                            // block stride between steps
    const int q = 2 + j;
    if (q >= NP || cb % ST != 0 || cb / ST > 8) return 0;
    const int k = cb / ST, G = q >> 1, b = q & 1;
    const int col = 64 * G + 16 * b + 4 * c.g;            // re column of the part; im = + 32
    int stores = 0;
    if (k == 0) {
      if constexpr (FROM_COORDS) {
        const int f0 = 32 * G + 16 * b + 4 * c.g;
        f32x4 uu = *reinterpret_cast<const f32x4*>(c.sb0 + f0);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const f32x4 wd = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + f0);
#pragma unroll
          for (int e = 0; e < 4; ++e) uu[e] = __builtin_fmaf(c.x[d], wd[e], uu[e]);
        }
        w[0] = uu; w[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const float* bl = c.sbias + l_src * P + 64 * G + 16 * b + 4 * c.g;
        const f32x4 b_re = *reinterpret_cast<const f32x4*>(bl), b_im = *reinterpret_cast<const f32x4*>(bl + 32);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w[0][e] = __builtin_fmaf(src[4 * G + b][e], c.inv_prev, b_re[e]);
          w[1][e] = __builtin_fmaf(src[4 * G + 2 + b][e], c.inv_prev, b_im[e]);
        }
      }
      if constexpr (ST_LIN) {
        f32x4 lu, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lu[e] = w[0][e] * c.st_inv_c; lv[e] = w[1][e] * c.st_inv_c; }
        *reinterpret_cast<f32x4*>(c.st_lin + col) = lu;
        *reinterpret_cast<f32x4*>(c.st_lin + col + 32) = lv;
        stores += 2;
      }
    } else if (k <= 4) {
      const int e = k - 1;
      float a, bq;
      fx_gabor(w[0][e], w[1][e], FROM_COORDS ? fp.k2_first : fp.k2, a, bq);
      w[2][e] = a; w[3][e] = bq;
      if constexpr (ST_F32) {
        if (k == 4) {
          *reinterpret_cast<f32x4*>(c.st_out + col) = w[2];
          *reinterpret_cast<f32x4*>(c.st_out + col + 32) = w[3];
          stores += 2;
#pragma unroll
          for (int x = 0; x < 4; ++x) c.amx = __builtin_fmaxf(c.amx, __builtin_fmaxf(__builtin_fabsf(w[2][x]), __builtin_fabsf(w[3][x])));
        }
      }
    } else {
      const int im = (k - 5) >> 1, pr = (k - 5) & 1;      // re / im plane, element pair
      unsigned h, lo;
      fx_split_pair<KIND>(w[2 + im][2 * pr], w[2 + im][2 * pr + 1], a_scale, h, lo);
      Fh[2 * G + im][2 * b + pr] = h;
      Fl[2 * G + im][2 * b + pr] = lo;
      if constexpr (ST_SPLIT) {
        if (pr == 1) {
          *reinterpret_cast<fx_u32x4*>(c.st_out + col + 32 * im) =
              fx_u32x4{Fh[2 * G + im][2 * b], Fh[2 * G + im][2 * b + 1], Fl[2 * G + im][2 * b], Fl[2 * G + im][2 * b + 1]};
          stores += 1;
        }
      }
    }
    return stores;
  } else {
    constexpr int ACT = FxKind<KIND>::ACT;
    constexpr int HB = NB / 2;                            // blocks per part
    static_assert(CPLX || HB >= 7, "seven producer steps per part");
    const int i = cb / HB, k = cb % HB;
    const int q = 2 + 2 * j + i;
    if (q >= NP || k > 6) return 0;
    const int col = 16 * q + 4 * c.g;
    int stores = 0;
    if (k == 0) {
      if constexpr (FROM_COORDS) {
        f32x4 lin = *reinterpret_cast<const f32x4*>(c.sb0 + col);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const f32x4 wd = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) lin[e] = __builtin_fmaf(c.x[d], wd[e], lin[e]);
        }
        w[0] = lin;
      } else {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(c.sbias + l_src * P + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[0][e] = __builtin_fmaf(src[q][e], c.inv_prev, bv[e]);
      }
      if constexpr (ST_LIN) {
        f32x4 lu;
#pragma unroll
        for (int e = 0; e < 4; ++e) lu[e] = w[0][e] * c.st_inv_c;
        *reinterpret_cast<f32x4*>(c.st_lin + col) = lu;
        stores += 1;
      }
    } else if (k <= 4) {
      w[1][k - 1] = fx_act<ACT>(w[0][k - 1]);
      if constexpr (ST_F32) {
        if (k == 4) {
          *reinterpret_cast<f32x4*>(c.st_out + col) = w[1];
          stores += 1;
#pragma unroll
          for (int x = 0; x < 4; ++x) c.amx = __builtin_fmaxf(c.amx, __builtin_fabsf(w[1][x]));
        }
      }
    } else {
      const int pr = k - 5;
      unsigned h, lo;
      fx_split_pair<KIND>(w[1][2 * pr], w[1][2 * pr + 1], a_scale, h, lo);
      Fh[q >> 1][2 * (q & 1) + pr] = h;
      Fl[q >> 1][2 * (q & 1) + pr] = lo;
      if constexpr (ST_SPLIT) {
        if (pr == 1) {
          const int jj = q >> 1, bb = q & 1;
          *reinterpret_cast<fx_u32x4*>(c.st_out + col) = fx_u32x4{Fh[jj][2 * bb], Fh[jj][2 * bb + 1], Fl[jj][2 * bb], Fl[jj][2 * bb + 1]};
          stores += 1;
        }
      }
    }
    return stores;
  }
}

// the wave's own bound on |h| for activations without an a-priori one (relu): max |value| over the wave's tile, evaluated
// once more (relu: a compare) -> the power-of-two scale that maps it into [2^14, 2^15)
template <int KIND, int NB, bool FROM_COORDS>
WIRE_DEVINL float fx_wave_scale(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB], const int l_src,
                                float& inv) {
  float m = 0.f;
#pragma unroll
  for (int q = 0; q < fx_nparts<KIND, NB>(); ++q) {
    f32x4 o_re, o_im;
    fx_part<KIND, NB, FROM_COORDS>(fp, c, src, q, l_src, o_re, o_im);
#pragma unroll
    for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(o_re[e]), __builtin_fabsf(o_im[e])));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
  float s;
  wire_x2_scales((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(m)), s, inv);
  return s;
}

// hidden layer l (1 .. L):  dst = h_{l-1} W_l^T  with the operands' scales still on it; h_{l-1} comes from the coordinates
// (l = 1) or from `src`, the accumulators of layer l - 1.  During the MFMAs of stage j the vector unit produces what stage
// j + 1 (real nets: parts 2 j + 2, 2 j + 3) or the stage pair after the current one (wire: part j + 2) will consume.
template <int KIND, int NB, int RING, bool FROM_COORDS, int ABL, int TRAIN>
WIRE_DEVINL void fx_layer(const FusedFwdParams& fp, FxCtx& c, const f32x4 (&src)[NB], f32x4 (&dst)[NB], const int l) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int S = NB / 2, STAGE = NB * 2048, P = 16 * NB;
  if constexpr (TRAIN) {
    // layer l - 1 is produced (and stored) under this layer's MFMAs: lin_{l-1} (real nets but relu: also lin_0; wire: l - 1
    // >= 1), out_{l-1} fp32 for layer 0 and for relu, pre-split for the inner layers of the other kinds
    const size_t ro = (size_t)c.row * P;
    const int ls = l - 1;
    c.st_lin = nullptr;
    if (KIND != NK_RELU && (!FROM_COORDS || !CPLX))
      c.st_lin = (FROM_COORDS ? fp.lin0 : fp.lin + (size_t)(ls - 1) * fp.lin_stride) + ro;
    c.st_out = fp.out + (size_t)ls * fp.out_stride + ro;
    c.st_split = KIND != NK_RELU && !FROM_COORDS;
    c.st_inv_c = TRAIN == 2 ? 1.f : (FROM_COORDS ? fp.inv_c_first : fp.inv_c_hidden);
    c.amx = 0.f;
  }
  float sA = 1.f, invA = 1.f;                          // (bounded activations are split unscaled)
  if constexpr (KIND == NK_RELU) sA = fx_wave_scale<KIND, NB, FROM_COORDS>(fp, c, src, l - 1, invA);
  float sB, invB;
  fx_weight_scales(fp.wamax + (size_t)(l - 1) * fp.wamax_stride, c.lane, fp.c_hidden, sB, invB);
  (void)sB;
  fx_u32x4 Fh[S], Fl[S];
  f32x4 w[4];
  // parts 0 and 1 (what stage 0 -- wire: stages 0 and 1 -- consumes) before the first stage.  (PIPE = false -- ALL parts
  // before the first stage, the stages carrying MFMAs and fragment reads only -- was tried for P = 384, whose two accumulator
  // sets of 96 registers leave the pipelined producer 92 (forward-only) to 305 (training) spilled registers: it spilled
  // 324 / 604 instead and stays off.  The P = 384 training forward is therefore not dispatched: with its spills it ran the
  // K = 181 step 10 % SLOWER than the layer-by-layer kernels, 4.85 against 4.39 ms on one box.)
  // ys[0] / ys[1]: stores issued since the last / the last but one weight-stream issue (the counted wait of fx_stage_top)
  constexpr bool PIPE = true;
  int ys0 = c.ys, ys1 = 0;
#pragma unroll
  for (int jj = (CPLX ? -2 : -1); jj < (PIPE ? 0 : S); ++jj) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) ys0 += fx_slice<KIND, NB, FROM_COORDS, TRAIN>(fp, c, src, l - 1, sA, jj, cb, w, Fh, Fl);
    if constexpr (!PIPE) __builtin_amdgcn_sched_barrier(0);   // one part at a time: the scheduler must not open them all
  }
#pragma unroll
  for (int j = 0; j < S; ++j) {
    // RING 3: the pieces of stage t were issued two tops ago -- younger: the stores since then (ys1 + ys0) and the pieces of
    // stage t + 1 (added in fx_stage_top); RING 2: one top ago -- younger: ys0
    fx_stage_top<NB, RING, ABL>(fp, c, RING == 3 ? ys1 + ys0 : ys0);
    ys1 = ys0; ys0 = 0;
    const unsigned char* Sb = c.ring + c.buf * STAGE + c.lane * 16;
    const fx_f16x8 ah = __builtin_bit_cast(fx_f16x8, Fh[j]), al = __builtin_bit_cast(fx_f16x8, Fl[j]);
    // weight fragments FX_PFD blocks ahead, in a rotating set of registers
    fx_f16x8 bh[FX_PFD + 1], bl[FX_PFD + 1];
#pragma unroll
    for (int cb = 0; cb < FX_PFD; ++cb) {
      bh[cb] = *reinterpret_cast<const fx_f16x8*>(Sb + cb * 1024);
      bl[cb] = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb) * 1024);
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if (cb + FX_PFD < NB && !(ABL & 2)) {
        bh[(cb + FX_PFD) % (FX_PFD + 1)] = *reinterpret_cast<const fx_f16x8*>(Sb + (cb + FX_PFD) * 1024);
        bl[(cb + FX_PFD) % (FX_PFD + 1)] = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb + FX_PFD) * 1024);
      }
      fx_f16x8 xh = bh[(ABL & 2) ? 0 : cb % (FX_PFD + 1)], xl = bl[(ABL & 2) ? 0 : cb % (FX_PFD + 1)];
      if constexpr (ABL & 2) { asm volatile("" : "+v"(xh)); asm volatile("" : "+v"(xl)); }   // (probe: no reads, no CSE)
      // small terms first; weights as the first operand: the block comes out transposed (row = lane & 15)
      if (j == 0) dst[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else FX_MFMA(xl, ah, dst[cb]);
      FX_MFMA(xh, al, dst[cb]);
      FX_MFMA(xh, ah, dst[cb]);
      if constexpr (!(ABL & 1) && PIPE) ys0 += fx_slice<KIND, NB, FROM_COORDS, TRAIN>(fp, c, src, l - 1, sA, j, cb, w, Fh, Fl);
      __builtin_amdgcn_sched_barrier(0);
    }
    c.t += 1;
    c.buf = c.buf + 1 == RING ? 0 : c.buf + 1;
  }
  // (what the next layer's first wait may count: only the stores since the LAST issue -- the one before it lies two tops
  //  back by then; RING 3 looks two tops back, so it also needs ys1 of the last stage: both are 0 there, the last stages of
  //  a layer produce nothing)
  c.ys = ys0;
  c.inv_prev = invA * invB;
  if constexpr (TRAIN) {
    // max |out_{l-1}| of the fp32-stored activations for the GEMMs that read them as operands (weight gradient)
    if (!c.st_split && fp.amax_out && !(TRAIN == 2 && FROM_COORDS))
      wire_amax_publish(fp.amax_out + (size_t)(l - 1) * WIRE_AMAX_SLOTS, c.amx, c.lane);
  }
}

// training forward, after the last hidden layer: lin_L (and relu's out_L, which its backward reads) from the accumulators
template <int KIND, int NB>
WIRE_DEVINL void fx_tail_train(const FusedFwdParams& fp, FxCtx& c, const f32x4 (&src)[NB]) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB;
  const size_t ro = (size_t)c.row * P;
  float* lin = KIND == NK_RELU ? nullptr : fp.lin + (size_t)(fp.L - 1) * fp.lin_stride + ro;
  float* out = KIND == NK_RELU ? fp.out + (size_t)fp.L * fp.out_stride + ro : nullptr;
  float amx = 0.f;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int col = 16 * cb + 4 * c.g;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(c.sbias + fp.L * P + col);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(src[cb][e], c.inv_prev, bv[e]);
    if (lin) {
      f32x4 lu;
#pragma unroll
      for (int e = 0; e < 4; ++e) lu[e] = r[e] * fp.inv_c_hidden;
      *reinterpret_cast<f32x4*>(lin + col) = lu;
    }
    if (out) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = r[e] > 0.f ? r[e] : 0.f; amx = __builtin_fmaxf(amx, o[e]); }
      *reinterpret_cast<f32x4*>(out + col) = o;
    }
  }
  (void)CPLX;
  if (out && fp.amax_out) wire_amax_publish(fp.amax_out + (size_t)fp.L * WIRE_AMAX_SLOTS, amx, c.lane);
}

// sums over the 16 lanes of a DPP row (the 16 rows a wave holds of one column) by quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror: every lane ends with the total.  N independent values, one butterfly step of all of them at a
// time (a DPP operand wants two wait states after the instruction that wrote it: N chains side by side hide them)
#define FX_DPP_ADD(v, ctrl) \
  ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
template <int N>
__device__ __forceinline__ void fx_row16_sum_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = FX_DPP_ADD(v[i], 0xB1);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = FX_DPP_ADD(v[i], 0x4E);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = FX_DPP_ADD(v[i], 0x141);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = FX_DPP_ADD(v[i], 0x140);
}

// training forward of a real net with the FINAL STAGE inside (fp.target != null): final linear layer, MSE terms, dL/dy, the
// final layer's gradient sums and g_lin_L from the accumulators of layer L (modules/siren.py:84-88, gauss.py:64-69,
// relu.py:112-118 forward; the MSE of wire_image_denoise.py:152 and autograd's backward of both).  Neither lin_L nor out_L
// is stored; the arithmetic is final_fused_kernel's with act / act' evaluated on r = c lin
template <int KIND, int NB>
WIRE_DEVINL void fx_tail_loss(const FusedFwdParams& fp, FxCtx& c, const f32x4 (&src)[NB]) {
  constexpr int ACT = FxKind<KIND>::ACT;
  constexpr int P = 16 * NB;
  static_assert(!FxKind<KIND>::CPLX, "real nets");
  const bool valid = c.row < c.n;
  const int r16 = c.lane & 15;
  const long long srow = c.ssrc[c.wave * 16 + r16];
  float tg[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < fp.O) tg[o] = fp.target[srow * fp.O + o];
  }
  f32x4 rr[NB], oo[NB];
  float yo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int col = 16 * cb + 4 * c.g;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(c.sbias + fp.L * P + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      rr[cb][e] = __builtin_fmaf(src[cb][e], c.inv_prev, bv[e]);
      oo[cb][e] = fx_act<ACT>(rr[cb][e]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o < fp.O) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(c.swf + o * P + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) yo[o] = __builtin_fmaf(oo[cb][e], wv[e], yo[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    yo[o] += __shfl_xor(yo[o], 16);
    yo[o] += __shfl_xor(yo[o], 32);
  }
  float gy[4], lsum = 0.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    gy[o] = 0.f;
    if (o < fp.O) {
      const float yy = yo[o] + fp.bfr[o];
      const float dlt = valid ? yy - tg[o] : 0.f;
      gy[o] = fp.gscale * dlt;
      lsum = __builtin_fmaf(dlt, dlt, lsum);
      if (c.g == o && valid) {
        fp.y[c.row * fp.O + o] = yy;
        if (fp.rec) fp.rec[srow * fp.O + o] = yy;
      }
    }
  }
  __syncthreads();                                       // every wave has read its last weight stage: the ring is free
  float* red = reinterpret_cast<float*>(c.ring);         // [waves][O][P], then [waves][8]
  float* red2 = red + FX_WAVES * 4 * P;
  float amx = 0.f;
  float* gp = fp.g_lin + (size_t)c.row * P;
  const float dk = ACT == ACT_GAUSS ? -2.f * fp.s * fp.s / fp.c_hidden : fp.w;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int col = 16 * cb + 4 * c.g;
    f32x4 go = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o < fp.O) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(c.swf + o * P + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) go[e] = __builtin_fmaf(gy[o], wv[e], go[e]);
      }
    }
    f32x4 gl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r = rr[cb][e];
      float v;
      if (ACT == ACT_SIREN) v = go[e] * dk * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(r));   // omega cos(omega lin)
      else if (ACT == ACT_GAUSS) v = go[e] * oo[cb][e] * dk * r;                                   // -2 s^2 lin out
      else v = r > 0.f ? go[e] : 0.f;
      gl[e] = v;
      amx = __builtin_fmaxf(amx, __builtin_fabsf(v));
    }
    *reinterpret_cast<f32x4*>(gp + col) = gl;
    // dL/dy^T h_L of the wave's 16 rows: every lane of a DPP row ends with the sum, lane r = 0 stores it
    float sv[16];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[4 * o + e] = gy[o] * oo[cb][e];
    fx_row16_sum_n<16>(sv);
    if (r16 == 0) {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < fp.O)
          *reinterpret_cast<f32x4*>(red + ((size_t)c.wave * fp.O + o) * P + col) =
              f32x4{sv[4 * o], sv[4 * o + 1], sv[4 * o + 2], sv[4 * o + 3]};
    }
  }
  {
    float sv[5] = {gy[0], gy[1], gy[2], gy[3], lsum};
    fx_row16_sum_n<5>(sv);
    if (c.lane == 0) {
#pragma unroll
      for (int o = 0; o < 5; ++o) red2[c.wave * 8 + o] = sv[o];
    }
  }
  __syncthreads();
  const int tid = c.wave * 64 + c.lane;
  for (int e = tid; e < fp.O * P; e += 64 * FX_WAVES) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < FX_WAVES; ++w) v += red[(size_t)w * fp.O * P + e];
    fp.part_w[(size_t)blockIdx.x * fp.O * P + e] = v;
  }
  if (tid < 5) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < FX_WAVES; ++w) v += red2[w * 8 + tid];
    if (tid < fp.O) fp.part_b[(size_t)blockIdx.x * fp.O + tid] = v;
    if (tid == 4) fp.loss_partial[blockIdx.x] = v;
  }
  wire_amax_publish(fp.amax_g, amx, c.lane);
}

// layer 0 of a positional-encoding net (modules/relu.py:62-75 + :28-29) as a GEMM layer: the 64 padded input features
// [c, {sin(2^i pi c_j), cos(2^i pi c_j)}] of this lane's row are evaluated in the lane (posenc_kernel's arithmetic), in the
// order the two stages' k slots want them; dst = features W_0^T with the weight scale still on it
template <int NB, int RING, int ABL>
WIRE_DEVINL void fx_layer_pe(const FusedFwdParams& fp, FxCtx& c, f32x4 (&dst)[NB]) {
  constexpr int STAGE = NB * 2048;
  fx_u32x4 Fh[2], Fl[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * j + (e < 4 ? 4 * c.g + e : 16 + 4 * c.g + (e - 4));
      float x = 0.f;
      if (k < fp.D) {
#pragma unroll
        for (int d = 0; d < 4; ++d) x = k == d ? c.x[d] : x;
      } else if (k < fp.D + 2 * fp.D * fp.pe_F) {
        const int ee = k - fp.D;
        const int i = ee / (2 * fp.D), jj = (ee % (2 * fp.D)) >> 1;
        float cj = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) cj = jj == d ? c.x[d] : cj;
        const float freq = (float)((double)(1 << i) * 3.14159265358979323846);
        float sn, cs;
        wire_sincos(freq * cj, sn, cs);
        x = (ee & 1) ? cs : sn;
      }
      v[e] = x;
    }
    unsigned h0, l0, h1, l1, h2, l2, h3, l3;
    fx_split2(v[0], v[1], h0, l0); fx_split2(v[2], v[3], h1, l1);
    fx_split2(v[4], v[5], h2, l2); fx_split2(v[6], v[7], h3, l3);
    Fh[j] = fx_u32x4{h0, h1, h2, h3};
    Fl[j] = fx_u32x4{l0, l1, l2, l3};
  }
  float sB, invB;
  fx_weight_scales(fp.wamax0, c.lane, 1.f, sB, invB);
  (void)sB;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    fx_stage_top<NB, RING, ABL>(fp, c, 0);
    const unsigned char* Sb = c.ring + c.buf * STAGE + c.lane * 16;
    const fx_f16x8 ah = __builtin_bit_cast(fx_f16x8, Fh[j]), al = __builtin_bit_cast(fx_f16x8, Fl[j]);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const fx_f16x8 xh = *reinterpret_cast<const fx_f16x8*>(Sb + cb * 1024);
      const fx_f16x8 xl = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb) * 1024);
      if (j == 0) dst[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else FX_MFMA(xl, ah, dst[cb]);
      FX_MFMA(xh, al, dst[cb]);
      FX_MFMA(xh, ah, dst[cb]);
    }
    c.t += 1;
    c.buf = c.buf + 1 == RING ? 0 : c.buf + 1;
  }
  c.inv_prev = invB;
}

// final linear layer on h_L = act(accumulators of layer L): y[o] = sum_c h_L[c] wf[o][c] + bf[o]
// (modules/wire.py:156-157,164-165: the image holds (W_re | -W_im), i.e. the real part of the complex product)
template <int KIND, int NB>
WIRE_DEVINL void fx_final(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB]) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB;
  float yo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < fx_nparts<KIND, NB>(); ++q) {
    f32x4 o_re, o_im;
    fx_part<KIND, NB, false>(fp, c, src, q, fp.L, o_re, o_im);
    const int col = CPLX ? 64 * (q >> 1) + 16 * (q & 1) + 4 * c.g : 16 * q + 4 * c.g;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o < fp.O) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(c.swf + o * P + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) yo[o] = __builtin_fmaf(o_re[e], wv[e], yo[o]);
        if constexpr (CPLX) {
          const f32x4 wi = *reinterpret_cast<const f32x4*>(c.swf + o * P + col + 32);
#pragma unroll
          for (int e = 0; e < 4; ++e) yo[o] = __builtin_fmaf(o_im[e], wi[e], yo[o]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    yo[o] += __shfl_xor(yo[o], 16);
    yo[o] += __shfl_xor(yo[o], 32);
  }
  if (c.g == 0 && c.row < c.n) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < fp.O) fp.y[c.row * fp.O + o] = yo[o] + fp.bfr[o];
  }
}

template <int NB, int RING>
constexpr int fx_lds_bytes() {
  return RING * NB * 2048 + (4 * 16 * NB + 16 * NB + (FX_LMAX + 1) * 16 * NB + 4 * 16 * NB) * 4 +   // (first layer sized for PF = P)
         FX_ROWS * 8;                                     // source indices of the rows (final stage inside the training forward)
}

template <int KIND, int NB, int RING, int ABL = 0, int TRAIN = 0, bool PE = false>
__global__ __launch_bounds__(64 * FX_WAVES) void fused_fwd_kernel(const FusedFwdParams fp) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB, PF = CPLX ? P / 2 : P, S = NB / 2, STAGE = NB * 2048;
  static_assert(NB % FxKind<KIND>::U == 0 && NB % 4 == 0, "whole epilogue units, whole DMA rounds");
  extern __shared__ __attribute__((aligned(1024))) unsigned char fx_smem[];
  float* const sW0 = reinterpret_cast<float*>(fx_smem + RING * STAGE);
  float* const sb0 = sW0 + 4 * P;
  float* const sbias = sb0 + P;
  float* const swf = sbias + (FX_LMAX + 1) * P;
  long long* const ssrc = reinterpret_cast<long long*>(swf + 4 * P);
  const int tid = threadIdx.x;
  FxCtx c;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.g = c.lane >> 4;
  c.n = fp.n;
  c.row = (long long)blockIdx.x * FX_ROWS + c.wave * 16 + (c.lane & 15);
  c.ring = fx_smem;
  c.sW0 = sW0; c.sb0 = sb0; c.sbias = sbias; c.swf = swf; c.ssrc = ssrc;
  c.t = 0; c.T = fp.L * S + (PE ? 2 : 0); c.buf = 0;
  c.inv_prev = 1.f;
  c.st_lin = nullptr; c.st_out = nullptr; c.st_split = false; c.st_inv_c = 1.f; c.amx = 0.f; c.ys = 0;
  // the weight stream starts first: it does not depend on anything below
  fx_issue<NB, RING>(fp, c, 0, 0);
  if (RING == 3 && c.T > 1) fx_issue<NB, RING>(fp, c, 1, 1);
  {
    const long long rowc = c.row < c.n ? c.row : c.n - 1;
#pragma unroll
    for (int d = 0; d < 4; ++d) c.x[d] = d < fp.D ? fp.coords[rowc * fp.D + d] : 0.f;
  }
  if constexpr (!PE) {
    for (int i = tid; i < 4 * PF; i += 64 * FX_WAVES) {
      const int d = i / PF, f = i - d * PF;
      sW0[i] = (d < fp.D && f < fp.K) ? fp.c_first * fp.W0[f * fp.D + d] : 0.f;
    }
    for (int i = tid; i < PF; i += 64 * FX_WAVES) sb0[i] = i < fp.K ? fp.c_first * fp.b0[i] : 0.f;
  }
  for (int i = tid; i < fp.L * P; i += 64 * FX_WAVES) {   // row l = layer l's bias, l = 1 .. L (row 0: a GEMM first layer's)
    const int l = i / P;
    sbias[P + i] = fp.c_hidden * fp.bias[(size_t)l * fp.bias_stride + (i - l * P)];
  }
  if constexpr (PE)
    for (int i = tid; i < P; i += 64 * FX_WAVES) sbias[i] = fp.bias0[i];
  if (!TRAIN || fp.target) {                             // (the training forward without the final stage ends at lin_L)
    for (int i = tid; i < fp.O * P; i += 64 * FX_WAVES) swf[i] = fp.wf[i];
  }
  if constexpr (TRAIN != 0 && !CPLX) {
    if (fp.target) {
      for (int i = tid; i < FX_ROWS; i += 64 * FX_WAVES) {
        const long long row = (long long)blockIdx.x * FX_ROWS + i;
        ssrc[i] = row < fp.n ? (fp.idx ? (long long)fp.idx[row] : fp.first + row) : 0;
      }
    }
  }
  __syncthreads();

  f32x4 accA[NB], accB[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) accB[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int l = 2;
  if constexpr (PE) {
    // positional-encoding net: layer 0 is a GEMM over the 64 encoded features (two stages at the head of the weight stream),
    // layer 1 then reads accumulators like every later one
    fx_layer_pe<NB, RING, ABL>(fp, c, accA);
    l = 1;
  } else {
    fx_layer<KIND, NB, RING, true, ABL, TRAIN>(fp, c, accB, accA, 1);
  }
  for (; l + 1 <= fp.L; l += 2) {
    fx_layer<KIND, NB, RING, false, ABL, TRAIN>(fp, c, accA, accB, l);
    fx_layer<KIND, NB, RING, false, ABL, TRAIN>(fp, c, accB, accA, l + 1);
  }
  if (l <= fp.L) {
    fx_layer<KIND, NB, RING, false, ABL, TRAIN>(fp, c, accA, accB, l);
    if constexpr (TRAIN != 0 && !CPLX) {
      if (fp.target) fx_tail_loss<KIND, NB>(fp, c, accB);
      else fx_tail_train<KIND, NB>(fp, c, accB);
    } else if constexpr (TRAIN) fx_tail_train<KIND, NB>(fp, c, accB);
    else fx_final<KIND, NB>(fp, c, accB);
  } else {
    if constexpr (TRAIN != 0 && !CPLX) {
      if (fp.target) fx_tail_loss<KIND, NB>(fp, c, accA);
      else fx_tail_train<KIND, NB>(fp, c, accA);
    } else if constexpr (TRAIN) fx_tail_train<KIND, NB>(fp, c, accA);
    else fx_final<KIND, NB>(fp, c, accA);
  }
}

// ---------------------------------------------------------------------------
// the data-gradient CHAIN of the real nets: g_lin_L -> g_lin_{L-1} -> ... -> g_lin_1 in one kernel
//
//   g_out_{l-1} = g_lin_l W_l          (the transposed weight image, same k-permuted 2 x fp16 format)
//   g_lin_{l-1} = g_out_{l-1} * act'(lin_{l-1})        (autograd of modules/siren.py:48-49, gauss.py:27-28, relu.py:28-29)
//
// with g_lin_l never re-read: the accumulators of one link, multiplied by the activation derivative, ARE the next link's
// activation operand (the same register chaining as the forward).  Each g_lin_{l-1} is stored once (the weight-gradient GEMM
// of layer l - 1 reads it) with its maximum published for that GEMM's operand scale; lin_{l-1} (relu: out_{l-1}) is
// prefetched into registers while the link's MFMAs run.  The scale of a link's operand is the wave's own maximum of the
// g_lin it has just formed (gradients have no a-priori bound), so a link's epilogue is not pipelined under the next link's
// MFMAs; the two waves of a SIMD overlap instead.  The last link (layer 1) differentiates through the FIRST layer's
// activation (its own omega_0, lin_0 in the reference's units) and, for a native first layer, does not store g_lin_0: the
// first layer's weight and bias gradient sums g_lin_0^T [x | 1] are formed from the accumulators (16 rows of a wave by DPP row
// operations, the waves through LDS) and written per workgroup, as the layer-by-layer kernel's epilogue does per 256-row
// tile (wire_gemmh_epi.h, cr_partial).
// ---------------------------------------------------------------------------
// ABL (harness builds, -DWIRE_FX_ABLATE; results wrong): 1 no lin_{l-1} loads, 2 no g_lin stores, 4 no activation derivative,
// 8 no MFMAs, 16 no stage barrier / weight wait, 32 no weight-fragment LDS reads, 64 no splits
template <int KIND, int NB, int RING, int W, int ABL = 0>
__global__ __launch_bounds__(64 * W, 8 / W) void fused_bwd_kernel(const FusedBwdParams fp) {
  constexpr int ACT = FxKind<KIND>::ACT;
  constexpr int P = 16 * NB, S = NB / 2, STAGE = NB * 2048;
  static_assert(!FxKind<KIND>::CPLX, "real nets");
  extern __shared__ __attribute__((aligned(1024))) unsigned char fx_smem[];
  const int tid = threadIdx.x;
  FxCtx c;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.g = c.lane >> 4;
  c.n = fp.n;
  c.row = (long long)blockIdx.x * (16 * W) + c.wave * 16 + (c.lane & 15);
  c.ring = fx_smem;
  const int links = fp.L;                                // layers L .. 1
  c.t = 0; c.T = links * S; c.buf = 0;
  c.ys = 0;
  fx_issue<NB, RING, FusedBwdParams, W>(fp, c, 0, 0);
  if (RING == 3 && c.T > 1) fx_issue<NB, RING, FusedBwdParams, W>(fp, c, 1, 1);
  const size_t ro = (size_t)c.row * P + 4 * c.g;         // this lane's first column of block 0 (rows are padded to 128)

  // ---- the first operand: g_lin_L from memory, scaled by its tensor maximum (published by the final stage)
  f32x4 acc[NB];
  {
    const float* gp = fp.g + (size_t)fp.L * fp.g_stride + ro;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = *reinterpret_cast<const f32x4*>(gp + 16 * cb);
  }
  // rows beyond n (the last workgroup's overhang into the padding rows) carry zeros: whatever lies in the padding must
  // reach neither the wave's scale nor the published maxima
  const bool valid = c.row < c.n;
  if (!valid) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float sA, invA;
  wire_x2_scales(wire_amax_read(fp.gamax + (size_t)fp.L * WIRE_AMAX_SLOTS, c.lane), sA, invA);
  fx_u32x4 Fh[S], Fl[S];
  int ys0 = 0, ys1 = 0;
  for (int l = fp.L; l >= 1; --l) {
    const bool first = l == 1;                           // the link into the first layer
    // fragments of g_lin_l: blocks 2 j, 2 j + 1 of the lane's registers are stage j
#pragma unroll
    for (int j = 0; j < S; ++j) {
      unsigned h0, l0, h1, l1, h2, l2, h3, l3;
      if constexpr (ABL & 64) {
        h0 = h1 = h2 = h3 = 0x3c003c00u; l0 = l1 = l2 = l3 = 0x14001400u;
        asm volatile("" : "+v"(h0), "+v"(l0));
      } else {
        x2_split2(acc[2 * j][0], acc[2 * j][1], sA, h0, l0);
        x2_split2(acc[2 * j][2], acc[2 * j][3], sA, h1, l1);
        x2_split2(acc[2 * j + 1][0], acc[2 * j + 1][1], sA, h2, l2);
        x2_split2(acc[2 * j + 1][2], acc[2 * j + 1][3], sA, h3, l3);
      }
      Fh[j] = fx_u32x4{h0, h1, h2, h3};
      Fl[j] = fx_u32x4{l0, l1, l2, l3};
    }
    // lin_{l-1} (relu: out_{l-1}) of this lane's 64 columns: in flight under the MFMAs below.  NOT all at the head of the link
    // (the first edition): vector-memory operations complete in order, so the wait for the weight pieces of
    // stage 2 -- issued after these loads -- was a wait for 16 KB of HBM reads per wave, and the link ran as "HBM time, then
    // matrix-core time" (0.22 ms per link where either alone takes 0.11 - 0.13).  Three loads per stage behind the stage's
    // weight issue instead: a weight wait then has at most the loads of three stages ago in front of it
    f32x4 aux[NB];
    const float* ap = (first ? fp.aux0 : fp.aux + (size_t)(l - 1) * fp.aux_stride) + ro;
    float sB, invB;
    wire_x2_scales(wire_amax_read(fp.wamax + (size_t)(l - 1) * fp.wamax_stride, c.lane), sB, invB);
    (void)sB;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      fx_stage_top<NB, RING, (ABL & 16) ? 12 : 0, FusedBwdParams, W>(fp, c, RING == 3 ? ys1 + ys0 : ys0);
      ys1 = ys0; ys0 = 0;
      {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int cb = 3 * j + q;
          if (cb < NB) {
            if constexpr (ABL & 1) { aux[cb] = f32x4{0.3f, 0.2f, 0.1f, 0.4f}; asm volatile("" : "+v"(aux[cb])); }
            else { aux[cb] = *reinterpret_cast<const f32x4*>(ap + 16 * cb); ys0 += 1; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const unsigned char* Sb = c.ring + c.buf * STAGE + c.lane * 16;
      const fx_f16x8 ah = __builtin_bit_cast(fx_f16x8, Fh[j]), al = __builtin_bit_cast(fx_f16x8, Fl[j]);
      fx_f16x8 bh[FX_PFD + 1], bl[FX_PFD + 1];
#pragma unroll
      for (int cb = 0; cb < FX_PFD; ++cb) {
        bh[cb] = *reinterpret_cast<const fx_f16x8*>(Sb + cb * 1024);
        bl[cb] = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb) * 1024);
      }
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        if (cb + FX_PFD < NB && !(ABL & 32)) {
          bh[(cb + FX_PFD) % (FX_PFD + 1)] = *reinterpret_cast<const fx_f16x8*>(Sb + (cb + FX_PFD) * 1024);
          bl[(cb + FX_PFD) % (FX_PFD + 1)] = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb + FX_PFD) * 1024);
        }
        fx_f16x8 xh = bh[(ABL & 32) ? 0 : cb % (FX_PFD + 1)], xl = bl[(ABL & 32) ? 0 : cb % (FX_PFD + 1)];
        if constexpr (ABL & 32) { asm volatile("" : "+v"(xh)); asm volatile("" : "+v"(xl)); }
        if constexpr (ABL & 8) {
          asm volatile("" :: "v"(xh), "v"(xl), "v"(ah), "v"(al));
        } else {
          if (j == 0) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else FX_MFMA(xl, ah, acc[cb]);
          FX_MFMA(xh, al, acc[cb]);
          FX_MFMA(xh, ah, acc[cb]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      c.t += 1;
      c.buf = c.buf + 1 == RING ? 0 : c.buf + 1;
    }
    // ---- epilogue of the link: g_lin_{l-1} = g_out * act'(lin_{l-1}), stored, its maximum tracked
    const float inv = invA * invB;
    float amx = 0.f;
    float* gp = fp.g + (size_t)(l - 1) * fp.g_stride + ro;
    const bool rstore = fp.rstore != 0;                  // aux holds r = c lin (layer 0: c of the first omega_0)
    const float om = first ? fp.w1 : fp.w;
    const bool sums = first && fp.crp != nullptr;
    float xs[4] = {0.f, 0.f, 0.f, 0.f};
    float* red = reinterpret_cast<float*>(fx_smem);      // [W waves][P columns][5]
    if (sums) {
      if (valid)
        for (int d = 0; d < fp.D; ++d) xs[d] = fp.coords[(size_t)c.row * fp.D + d];
      __syncthreads();                                   // every wave has read its last weight stage: the ring is free
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      f32x4 gl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float go = acc[cb][e] * inv;
        float v;
        if (rstore && ACT == ACT_SIREN) {
          // aux = r = omega lin / 2 pi: d sin(omega lin) / d lin = omega cos(2 pi r)
          v = go * om * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(aux[cb][e]));
        } else if (rstore && ACT == ACT_GAUSS) {
          // aux = r = s sqrt(log2 e) lin: out = exp2(-r^2), d out / d lin = -2 s^2 lin out = -(2 s^2 / c) r out
          const float r = aux[cb][e];
          v = go * __builtin_amdgcn_exp2f(-(r * r)) * (-2.f * fp.s * fp.s / fp.c_hidden) * r;
        } else {
          float out = 0.f;
          if (ACT == ACT_GAUSS) out = real_act_fwd_lean<ACT_GAUSS>(aux[cb][e], om, fp.s);
          if (ACT == ACT_RELU) out = aux[cb][e];
          v = real_act_bwd_lean<ACT>(go, aux[cb][e], out, om, fp.s);
        }
        if constexpr (ABL & 4) v = go + aux[cb][e];
        gl[e] = valid ? v : 0.f;
        amx = __builtin_fmaxf(amx, __builtin_fabsf(gl[e]));
      }
      if (!sums) {
        if constexpr (ABL & 2) asm volatile("" :: "v"(gl));
        else *reinterpret_cast<f32x4*>(gp + 16 * cb) = gl;
      }
      acc[cb] = gl;
    }
    if (sums) {
      // g_lin_0^T [x | 1] of the wave's 16 rows through a wave-private LDS transpose, a quarter of the columns at a time: the
      // lanes write their 4-column pieces of 4 blocks as rows of T (stride 68 floats: 16-byte aligned, consecutive rows 4
      // banks apart), then lane c reads column c of the 16 rows (consecutive lanes, consecutive words: conflict-free) beside
      // the rows' [x] (broadcast reads) -- 30 vector instructions per block where sums over the 16 lanes of a DPP row (the
      // first edition: 20 chains of four row operations side by side) took 100: - 0.04 ms per step
      // (profiles/r04_bwd_lsum_ab.txt).  One wave, one in-order LDS queue: no barrier between its writes and its reads.
      float* Tw = red + (size_t)W * P * 5 + (size_t)c.wave * (16 * 68);
      float* rv = red + (size_t)W * P * 5 + (size_t)W * (16 * 68) + (size_t)c.wave * 64;
      const int r16 = c.lane & 15;
      if (c.g == 0) *reinterpret_cast<f32x4*>(rv + 4 * r16) = f32x4{xs[0], xs[1], xs[2], xs[3]};
#pragma unroll
      for (int qt = 0; qt < NB / 4; ++qt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Tw + r16 * 68 + 16 * i + 4 * c.g) = acc[4 * qt + i];
        float sd[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) {
          const float t = Tw[r2 * 68 + c.lane];
          const f32x4 xv = *reinterpret_cast<const f32x4*>(rv + 4 * r2);
#pragma unroll
          for (int d = 0; d < 4; ++d) sd[d] = __builtin_fmaf(t, xv[d], sd[d]);
          sd[4] += t;
        }
        float* rp = red + ((size_t)c.wave * P + 64 * qt + c.lane) * 5;
#pragma unroll
        for (int d = 0; d < 5; ++d) rp[d] = sd[d];
      }
    }
    if (sums) {
      __syncthreads();
      for (int e = tid; e < P * 5; e += 64 * W) {
        const int col = e / 5, d = e - 5 * col;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) v += red[(size_t)w * P * 5 + e];
        if (col < fp.C) fp.crp[((size_t)blockIdx.x * fp.C + col) * 5 + d] = v;
      }
      return;                                            // (first is the last link)
    }
    if constexpr (!(ABL & 2)) ys0 += NB;
    if (first) return;
    // the next link's operand scale from this wave's maximum; the tensor's maximum for the weight-gradient GEMM
#pragma unroll
    for (int o = 32; o; o >>= 1) amx = __builtin_fmaxf(amx, __shfl_xor(amx, o));
    wire_x2_scales((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(amx)), sA, invA);
    if (c.lane == 0) {
      const unsigned bits = __float_as_uint(amx);
      unsigned* slot = fp.gamax + (size_t)(l - 1) * WIRE_AMAX_SLOTS + (blockIdx.x & (WIRE_AMAX_SLOTS - 1));
      if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int fx_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// "fused_fwd" / WIRE_FUSED_FWD: 1 (default) = forward-only calls of the nets below run the fused kernel
static std::atomic<int> g_fused_fwd{fx_env("WIRE_FUSED_FWD", 1)};
#ifdef WIRE_FX_ABLATE
static std::atomic<int> g_fx_ablate{0};
#endif
// "fused_train" / WIRE_FUSED_TRAIN: 1 (default) = training forwards of those nets run it too (storing lin_l / out_l)
static std::atomic<int> g_fused_train{fx_env("WIRE_FUSED_TRAIN", 1)};
bool fused_train_enabled() { return g_fused_train != 0 && g_fused_fwd != 0; }
// "fused_final" / WIRE_FUSED_FINAL: 1 = wire_train_fwd_bwd of the real nets forms loss, dL/dy, g_lin_L and the final layer's
// gradient sums inside the training forward (fx_tail_loss) instead of a pass over the stored lin_L / out_L.  Default 0:
// measured neutral to slower (siren 2.05 / 2.03, gauss 2.02 / 1.94, relu 1.95 / 1.93 ms per step with / without,
// profiles/r04_fused_final_ab.txt) -- the 0.18 ms pass it removes is HBM-bound and cheap, the tail it adds (3500 vector
// instructions per wave, a third of them the 16-row sums of dL/dy^T h_L by DPP) runs with the matrix cores idle
static std::atomic<int> g_fused_final{fx_env("WIRE_FUSED_FINAL", 0)};
bool fused_final_enabled() { return g_fused_final != 0 && fused_train_enabled(); }
int fused_bwd_knob();
void fused_bwd_knob_set(int v);
void fxb_ablate_set(int v);
void fused_bwd_w_set(int v);
int fused_bwd_w_get();
int fused_tune_get(const char* key) {
  if (!strcmp(key, "fused_fwd")) return g_fused_fwd;
  if (!strcmp(key, "fused_train")) return g_fused_train;
  if (!strcmp(key, "fused_bwd")) return fused_bwd_knob();
  if (!strcmp(key, "fused_final")) return g_fused_final;
  if (!strcmp(key, "fused_bwd_w")) return fused_bwd_w_get();
  return -1;
}
int fused_tune_set(const char* key, int value) {
  if (!strcmp(key, "fused_fwd") && value >= 0 && value <= 1) { g_fused_fwd = value; return 0; }
  if (!strcmp(key, "fused_train") && value >= 0 && value <= 1) { g_fused_train = value; return 0; }
  if (!strcmp(key, "fused_bwd") && value >= 0 && value <= 1) { fused_bwd_knob_set(value); return 0; }
  if (!strcmp(key, "fused_final") && value >= 0 && value <= 1) { g_fused_final = value; return 0; }
#ifdef WIRE_FX_ABLATE
  if (!strcmp(key, "fxb_ablate") && value >= 0 && value <= 127) { fxb_ablate_set(value); return 0; }
#endif
  if (!strcmp(key, "fused_bwd_w") && (value == 4 || value == 8)) { fused_bwd_w_set(value); return 0; }
#ifdef WIRE_FX_ABLATE
  if (!strcmp(key, "fx_ablate") && value >= 0 && value <= 15) { g_fx_ablate = value; return 0; }
#endif
  return -1;
}

// shapes with a kernel: the 256-feature real nets (BASELINE.json configs[4]) and `wire` at padded widths 192 / 256 / 384
// (config 1's K = 90, K <= 128, hidden_features = 256 -> K = 181)
bool fused_fwd_shape(int kind, int P) {
  if (kind == NK_SIREN || kind == NK_GAUSS || kind == NK_RELU) return P == 256;
  if (kind == NK_WIRE) return P == 192 || P == 256 || P == 384;
  return false;
}
bool fused_fwd_enabled() { return g_fused_fwd != 0; }
float fused_pre_scale(int kind, float omega0, float scale0) {
  if (kind == NK_SIREN || kind == NK_WIRE) return (float)((double)omega0 / 6.283185307179586);
  if (kind == NK_GAUSS) return (float)((double)scale0 * 1.2011224087864498);      // sqrt(log2 e)
  return 1.f;
}

template <int KIND, int NB, int RING, int ABL = 0>
static hipError_t fx_launch_t(hipStream_t s, const FusedFwdParams& fp) {
  constexpr int LDS = fx_lds_bytes<NB, RING>();
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
  const unsigned grid = (unsigned)((fp.n + FX_ROWS - 1) / FX_ROWS);
  if (fp.out != nullptr) {                               // training forward: stores what the backward reads
    if constexpr (ABL == 0) {
      if constexpr (KIND == NK_SIREN || KIND == NK_GAUSS) {
        if (fp.rstore) {
          const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING, 0, 2>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          if (attr != hipSuccess) return attr;
          hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING, 0, 2>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
          return hipGetLastError();
        }
      }
      if (fp.rstore) return hipErrorInvalidValue;
      if constexpr (KIND == NK_RELU && NB == 16) {
        if (fp.pe_F > 0) {
          const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING, 0, 1, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          if (attr != hipSuccess) return attr;
          hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING, 0, 1, true>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
          return hipGetLastError();
        }
      }
      if (fp.pe_F > 0) return hipErrorInvalidValue;
      const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING, 0, 1>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (attr != hipSuccess) return attr;
      hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING, 0, 1>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (KIND == NK_RELU && NB == 16 && ABL == 0) {
    if (fp.pe_F > 0) {
      const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING, 0, 0, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (attr != hipSuccess) return attr;
      hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING, 0, 0, true>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
      return hipGetLastError();
    }
  }
  if (fp.pe_F > 0) return hipErrorInvalidValue;
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING, ABL>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING, ABL>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
  return hipGetLastError();
}

hipError_t launch_fused_fwd(hipStream_t s, int kind, int P, const FusedFwdParams& fp) {
  if (fp.n <= 0) return hipSuccess;
  if (!fused_fwd_shape(kind, P) || fp.L < 1 || fp.L > FX_LMAX || fp.D < 1 || fp.D > 4 || fp.O < 1 || fp.O > 4 ||
      fp.n > 0x7fffffffLL * FX_ROWS)
    return hipErrorInvalidValue;
  if (!(fp.c_first > 0.f) || !(fp.c_hidden > 0.f)) return hipErrorInvalidValue;
  if (fp.pe_F > 0 && (kind != NK_RELU || !fp.bias0 || !fp.wamax0 || fp.D + 2 * fp.D * fp.pe_F > 64)) return hipErrorInvalidValue;
  if (fp.pe_F == 0 && (!fp.W0 || !fp.b0)) return hipErrorInvalidValue;
  if (fp.out != nullptr && ((kind != NK_RELU && !fp.lin) || ((kind == NK_SIREN || kind == NK_GAUSS) && !fp.lin0) || !fp.amax_out))
    return hipErrorInvalidValue;
  if (fp.target != nullptr && (kind == NK_WIRE || !fp.out || !fp.wf || !fp.bfr || !fp.y || !fp.g_lin || !fp.part_w || !fp.part_b ||
                               !fp.loss_partial || !fp.amax_g))
    return hipErrorInvalidValue;
#ifdef WIRE_FX_ABLATE
  if (kind == NK_SIREN) {
    switch (g_fx_ablate.load()) {
      case 1: return fx_launch_t<NK_SIREN, 16, 3, 1>(s, fp);
      case 2: return fx_launch_t<NK_SIREN, 16, 3, 2>(s, fp);
      case 3: return fx_launch_t<NK_SIREN, 16, 3, 3>(s, fp);
      case 4: return fx_launch_t<NK_SIREN, 16, 3, 4>(s, fp);
      case 8: return fx_launch_t<NK_SIREN, 16, 3, 8>(s, fp);
      case 12: return fx_launch_t<NK_SIREN, 16, 3, 12>(s, fp);
      case 15: return fx_launch_t<NK_SIREN, 16, 3, 15>(s, fp);
      default: break;
    }
  }
#endif
  switch (kind) {
    case NK_SIREN: return fx_launch_t<NK_SIREN, 16, 3>(s, fp);
    case NK_GAUSS: return fx_launch_t<NK_GAUSS, 16, 3>(s, fp);
    case NK_RELU: return fx_launch_t<NK_RELU, 16, 3>(s, fp);
    case NK_WIRE:
      if (P == 192) return fx_launch_t<NK_WIRE, 12, 3>(s, fp);
      if (P == 256) return fx_launch_t<NK_WIRE, 16, 3>(s, fp);
      return fx_launch_t<NK_WIRE, 24, 2>(s, fp);
    default: return hipErrorInvalidValue;
  }
}

#ifdef FX_PROBE_ABL
template __global__ void fused_fwd_kernel<NK_SIREN, 16, 3, FX_PROBE_ABL>(const FusedFwdParams);
#endif

// ---- the data-gradient chain (real nets, P = 256)
static std::atomic<int> g_fused_bwd{fx_env("WIRE_FUSED_BWD", 1)};
bool fused_bwd_enabled() { return g_fused_bwd != 0 && fused_train_enabled(); }
bool fused_bwd_shape(int kind, int P) { return (kind == NK_SIREN || kind == NK_GAUSS || kind == NK_RELU) && P == 256; }
// W = waves per workgroup.  8: one workgroup per CU, ring of three stages.  4: TWO workgroups of 64 rows per CU, each with
// a ring of two stages (2 x 64 KB of LDS) -- they share nothing, so one's epilogue (no MFMAs: the link's activation
// derivative, maxima, splits) runs beside the other's MFMAs, which the two waves of a SIMD inside ONE workgroup cannot do
// (they meet at every stage barrier); price: the weight stream is fetched per 64 rows instead of per 128.
static std::atomic<int> g_fused_bwd_w{fx_env("WIRE_FUSED_BWD_W", 8)};   // (A/B on three nets, two rounds each: no difference -- profiles/r04_fused_bwd_w_ab.txt)
#ifdef WIRE_FX_ABLATE
static std::atomic<int> g_fxb_ablate{0};
#endif
template <int KIND, int W, int ABL = 0>
static hipError_t fxb_launch_w(hipStream_t s, const FusedBwdParams& fp) {
  constexpr int NB = 16, RING = W == 8 ? 3 : 2, LDS = RING * NB * 2048;
  static_assert((W * 16 * NB * 5 + W * 16 * 68 + W * 64) * 4 <= LDS, "sums: partials + transpose tiles + row vectors in the ring");
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_bwd_kernel<KIND, NB, RING, W, ABL>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) return attr;
  const unsigned grid = (unsigned)((fp.n + 16 * W - 1) / (16 * W));
  hipLaunchKernelGGL((fused_bwd_kernel<KIND, NB, RING, W, ABL>), dim3(grid), dim3(64 * W), LDS, s, fp);
  return hipGetLastError();
}
template <int KIND>
static hipError_t fxb_launch_t(hipStream_t s, const FusedBwdParams& fp, int W) {
#ifdef WIRE_FX_ABLATE
  if constexpr (KIND == NK_SIREN) {
    switch (g_fxb_ablate.load()) {
      case 1: return fxb_launch_w<KIND, 8, 1>(s, fp);
      case 2: return fxb_launch_w<KIND, 8, 2>(s, fp);
      case 3: return fxb_launch_w<KIND, 8, 3>(s, fp);
      case 4: return fxb_launch_w<KIND, 8, 4>(s, fp);
      case 7: return fxb_launch_w<KIND, 8, 7>(s, fp);
      case 8: return fxb_launch_w<KIND, 8, 8>(s, fp);
      case 16: return fxb_launch_w<KIND, 8, 16>(s, fp);
      case 23: return fxb_launch_w<KIND, 8, 23>(s, fp);
      case 32: return fxb_launch_w<KIND, 8, 32>(s, fp);
      case 64: return fxb_launch_w<KIND, 8, 64>(s, fp);
      case 55: return fxb_launch_w<KIND, 8, 55>(s, fp);
      case 119: return fxb_launch_w<KIND, 8, 119>(s, fp);
      case 127: return fxb_launch_w<KIND, 8, 127>(s, fp);
      default: break;
    }
  }
#endif
  return W == 8 ? fxb_launch_w<KIND, 8>(s, fp) : fxb_launch_w<KIND, 4>(s, fp);
}
// *tile_rows: rows per workgroup of this launch = rows per block of fp.crp
hipError_t launch_fused_bwd(hipStream_t s, int kind, int P, const FusedBwdParams& fp, int* tile_rows) {
  const int W = g_fused_bwd_w;
  if (tile_rows) *tile_rows = 16 * W;
  if (fp.n <= 0 || fp.L < 1) return hipSuccess;
  if (!fused_bwd_shape(kind, P) || fp.L > FX_LMAX || !fp.g || !fp.aux0 || (fp.L >= 2 && !fp.aux) || !fp.gamax || !fp.wamax ||
      !fp.wimg)
    return hipErrorInvalidValue;
  if (fp.crp && (!fp.coords || fp.D < 1 || fp.D > 4 || fp.C < 1 || fp.C > P)) return hipErrorInvalidValue;
  switch (kind) {
    case NK_SIREN: return fxb_launch_t<NK_SIREN>(s, fp, W);
    case NK_GAUSS: return fxb_launch_t<NK_GAUSS>(s, fp, W);
    case NK_RELU: return fxb_launch_t<NK_RELU>(s, fp, W);
    default: return hipErrorInvalidValue;
  }
}
int fused_bwd_knob() { return g_fused_bwd; }
void fused_bwd_knob_set(int v) { g_fused_bwd = v; }
void fused_bwd_w_set(int v) { g_fused_bwd_w = v; }
int fused_bwd_w_get() { return g_fused_bwd_w; }
#ifdef WIRE_FX_ABLATE
void fxb_ablate_set(int v) { g_fxb_ablate = v; }
#else
void fxb_ablate_set(int) {}
#endif
