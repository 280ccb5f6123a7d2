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
// Arithmetic = the 2 x fp16 split of wire_gemmx2h.hip (three partial products per fp32 product, fp32 accumulate, power-of-two
// operand scales: the weights' from their maximum slots, the activations' from their a-priori bound or the wave's own
// maximum), lean hardware-transcendental epilogues for the hidden layers, first_fwd_kernel's precise forms for layer 0.
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

WIRE_DEVINL void fx_dma16(const void* gsrc, unsigned char* lds_piece) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// weight image of one hidden layer for the fused kernel, from the fp32 forward image Bt[P][ldb] (row = output column):
//   Fx[stage kt (32 k)][plane (h, l)][16-column block cb][k slot ks (4)][column (16)][8 halves]
// with the k indices of a stage permuted as the header describes: slot (ks, e) holds k = 32 kt + 4 ks + e (e < 4) or
// 32 kt + 16 + 4 ks + (e - 4).  Scaled by the layer's power-of-two scale (its maximum slots), split into (h, l).
// ---------------------------------------------------------------------------
__global__ void fx_split_b_kernel(FxSplitBatch sb, int ldb, int P) {
  const float* __restrict__ Bt = sb.src[blockIdx.z];
  unsigned short* __restrict__ Fx = (unsigned short*)sb.dst[blockIdx.z];
  float s, inv;
  wire_x2_scales(wire_amax_read(sb.slots[blockIdx.z], threadIdx.x & 63), s, inv);
  const int k2 = (blockIdx.x * blockDim.x + threadIdx.x) * 2;   // pair of reduction indices (same group of four)
  const int j = blockIdx.y;                                     // output column
  if (k2 >= P) return;
  const float x0 = Bt[(size_t)j * ldb + k2], x1 = Bt[(size_t)j * ldb + k2 + 1];
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
hipError_t launch_fx_split_b_batch(hipStream_t s, const FxSplitBatch& sb, int nb, int ldb, int P) {
  if ((P & 31) || nb < 1 || nb > FX_SPLIT_MAXB) return hipErrorInvalidValue;
  dim3 grid((unsigned)((P / 2 + 127) / 128), (unsigned)P, (unsigned)nb);
  hipLaunchKernelGGL(fx_split_b_kernel, grid, dim3(128), 0, s, sb, ldb, P);
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
  int t, T, buf;              // weight stream: next stage to consume, number of stages, its ring buffer
  float inv_prev;             // 1 / (s_A s_B) of the accumulators in the source set
};

template <int NB, int RING>
WIRE_DEVINL void fx_issue(const FusedFwdParams& fp, const FxCtx& c, int t, int buf) {
  constexpr int STAGE = NB * 2048, PIECES = NB / 4;
  const unsigned char* src = fp.wimg + (size_t)t * STAGE + c.wave * 1024 + c.lane * 16;
  unsigned char* dst = c.ring + buf * STAGE + c.wave * 1024;
#pragma unroll
  for (int j = 0; j < PIECES; ++j) fx_dma16(src + j * 8192, dst + j * 8192);
}

// top of a stage: this wave's pieces of stage c.t have landed, then everybody's (barrier); the buffer that the barrier
// proves free (every wave has consumed stage c.t - 1) is refilled with stage c.t + RING - 1
template <int NB, int RING>
WIRE_DEVINL void fx_stage_top(const FusedFwdParams& fp, FxCtx& c) {
  constexpr int PIECES = NB / 4;
  if (RING == 3 && c.t + 1 < c.T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (c.t + RING - 1 < c.T) {
    int nb = c.buf + RING - 1;
    nb = nb >= RING ? nb - RING : nb;
    fx_issue<NB, RING>(fp, c, c.t + RING - 1, nb);
  }
}

// one unit of the producer: the fragments (h, l) of U / 2 stages from the coordinates (layer 0) or from the previous layer's
// accumulators.  LAST = the values feed the final linear layer instead: no split, returned as fp32 in `of`.
template <int KIND, int NB, bool FROM_COORDS, bool LAST>
WIRE_DEVINL void fx_produce(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB], const int u, const int l_src,
                            const float a_scale, fx_f16x8 (&ah)[FxKind<KIND>::U / 2], fx_f16x8 (&al)[FxKind<KIND>::U / 2],
                            f32x4 (&of)[FxKind<KIND>::U]) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB, PF = CPLX ? P / 2 : P;
  if constexpr (CPLX) {
    // group G = u: blocks 4 G + b (re) and 4 G + 2 + b (im) of features 32 G + 16 b + 4 g + q
    unsigned Hr[4], Lr[4], Hi[4], Li[4];
    const float w0 = fp.w, w0l2e = fp.w * 1.44269502f, ns2l2e = -(fp.s * fp.s) * 1.44269502f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int f0 = 32 * u + 16 * b + 4 * c.g;
      f32x4 o_re, o_im;
      if constexpr (FROM_COORDS) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(c.sb0 + f0);
        f32x4 wd[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) wd[d] = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + f0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float uu = bb[q];
#pragma unroll
          for (int d = 0; d < 4; ++d) uu = __builtin_fmaf(c.x[d], wd[d][q], uu);
          float a, bq;
          gabor_fwd_real(uu, fp.w1, fp.s, a, bq);
          const bool valid = f0 + q < fp.K;
          o_re[q] = valid ? a : 0.f;
          o_im[q] = valid ? bq : 0.f;
        }
      } else {
        const float* bl = c.sbias + (l_src - 1) * P + 64 * u + 16 * b + 4 * c.g;
        const f32x4 b_re = *reinterpret_cast<const f32x4*>(bl), b_im = *reinterpret_cast<const f32x4*>(bl + 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float uu = src[4 * u + b][q] * c.inv_prev + b_re[q];
          const float vv = src[4 * u + 2 + b][q] * c.inv_prev + b_im[q];
          float a, bq;
          gabor_fwd_lean(uu, vv, w0, w0l2e, ns2l2e, a, bq);
          const bool valid = f0 + q < fp.K;
          o_re[q] = valid ? a : 0.f;
          o_im[q] = valid ? bq : 0.f;
        }
      }
      if constexpr (LAST) { of[b] = o_re; of[2 + b] = o_im; }
      else {
        x2_split2(o_re[0], o_re[1], a_scale, Hr[2 * b], Lr[2 * b]);
        x2_split2(o_re[2], o_re[3], a_scale, Hr[2 * b + 1], Lr[2 * b + 1]);
        x2_split2(o_im[0], o_im[1], a_scale, Hi[2 * b], Li[2 * b]);
        x2_split2(o_im[2], o_im[3], a_scale, Hi[2 * b + 1], Li[2 * b + 1]);
      }
    }
    if constexpr (!LAST) {
      ah[0] = __builtin_bit_cast(fx_f16x8, fx_u32x4{Hr[0], Hr[1], Hr[2], Hr[3]});
      al[0] = __builtin_bit_cast(fx_f16x8, fx_u32x4{Lr[0], Lr[1], Lr[2], Lr[3]});
      ah[1] = __builtin_bit_cast(fx_f16x8, fx_u32x4{Hi[0], Hi[1], Hi[2], Hi[3]});
      al[1] = __builtin_bit_cast(fx_f16x8, fx_u32x4{Li[0], Li[1], Li[2], Li[3]});
    }
  } else {
    constexpr int ACT = FxKind<KIND>::ACT;
    unsigned H[4], Lo[4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = 16 * (2 * u + b) + 4 * c.g;
      f32x4 o;
      if constexpr (FROM_COORDS) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(c.sb0 + col);
        f32x4 wd[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) wd[d] = *reinterpret_cast<const f32x4*>(c.sW0 + d * PF + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float uu = bb[q];
#pragma unroll
          for (int d = 0; d < 4; ++d) uu = __builtin_fmaf(c.x[d], wd[d][q], uu);
          const float v = real_act_fwd<ACT>(uu, fp.w1, fp.s);
          o[q] = col + q < fp.K ? v : 0.f;
        }
      } else {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(c.sbias + (l_src - 1) * P + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lin = src[2 * u + b][q] * c.inv_prev + bv[q];
          const float v = real_act_fwd_lean<ACT>(lin, fp.w, fp.s);
          o[q] = col + q < fp.K ? v : 0.f;
        }
      }
      if constexpr (LAST) of[b] = o;
      else {
        x2_split2(o[0], o[1], a_scale, H[2 * b], Lo[2 * b]);
        x2_split2(o[2], o[3], a_scale, H[2 * b + 1], Lo[2 * b + 1]);
      }
    }
    if constexpr (!LAST) {
      ah[0] = __builtin_bit_cast(fx_f16x8, fx_u32x4{H[0], H[1], H[2], H[3]});
      al[0] = __builtin_bit_cast(fx_f16x8, fx_u32x4{Lo[0], Lo[1], Lo[2], Lo[3]});
    }
  }
}

// the wave's own bound on |h| for activations without an a-priori one (relu): max |value| over the wave's tile, evaluated
// once more (relu: a compare) -> the power-of-two scale that maps it into [2^14, 2^15)
template <int KIND, int NB, bool FROM_COORDS>
WIRE_DEVINL float fx_wave_scale(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB], const int l_src,
                                float& inv) {
  constexpr int U = FxKind<KIND>::U, NU = NB / U;
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    fx_f16x8 dh[U / 2], dl[U / 2];
    f32x4 of[U];
    fx_produce<KIND, NB, FROM_COORDS, true>(fp, c, src, u, l_src, 1.f, dh, dl, of);
#pragma unroll
    for (int b = 0; b < U; ++b)
      m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(of[b][0]), __builtin_fabsf(of[b][1])),
                                             __builtin_fmaxf(__builtin_fabsf(of[b][2]), __builtin_fabsf(of[b][3]))));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
  float s;
  wire_x2_scales((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(m)), s, inv);
  return s;
}

// hidden layer l (1 .. L):  dst = h_{l-1} W_l^T  with the operands' scales still on it; h_{l-1} comes from the coordinates
// (l = 1) or from `src`, the accumulators of layer l - 1
template <int KIND, int NB, int RING, bool FROM_COORDS>
WIRE_DEVINL void fx_layer(const FusedFwdParams& fp, FxCtx& c, const f32x4 (&src)[NB], f32x4 (&dst)[NB], const int l) {
  constexpr int U = FxKind<KIND>::U, NU = NB / U, HS = U / 2, STAGE = NB * 2048;
  float sA = fp.a_scale, invA = fp.a_scale_inv;
  if constexpr (KIND == NK_RELU) sA = fx_wave_scale<KIND, NB, FROM_COORDS>(fp, c, src, l - 1, invA);
  float sB, invB;
  wire_x2_scales(wire_amax_read(fp.wamax + (size_t)(l - 1) * fp.wamax_stride, c.lane), sB, invB);
  (void)sB;
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) dst[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  fx_f16x8 ah[HS], al[HS], nh[HS], nl[HS];
  f32x4 unused[U];
  fx_produce<KIND, NB, FROM_COORDS, false>(fp, c, src, 0, l - 1, sA, ah, al, unused);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    // the NEXT unit's fragments: vector work that runs under this unit's MFMAs
    if (u + 1 < NU) fx_produce<KIND, NB, FROM_COORDS, false>(fp, c, src, u + 1, l - 1, sA, nh, nl, unused);
#pragma unroll
    for (int sub = 0; sub < HS; ++sub) {
      fx_stage_top<NB, RING>(fp, c);
      const unsigned char* Sb = c.ring + c.buf * STAGE + c.lane * 16;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const fx_f16x8 bh = *reinterpret_cast<const fx_f16x8*>(Sb + cb * 1024);
        const fx_f16x8 bl = *reinterpret_cast<const fx_f16x8*>(Sb + (NB + cb) * 1024);
        // small terms first; weights as the first operand: the block comes out transposed (row = lane & 15)
        FX_MFMA(bl, ah[sub], dst[cb]);
        FX_MFMA(bh, al[sub], dst[cb]);
        FX_MFMA(bh, ah[sub], dst[cb]);
      }
      c.t += 1;
      c.buf = c.buf + 1 == RING ? 0 : c.buf + 1;
    }
#pragma unroll
    for (int sub = 0; sub < HS; ++sub) { ah[sub] = nh[sub]; al[sub] = nl[sub]; }
  }
  c.inv_prev = invA * invB;
}

// final linear layer on h_L = act(accumulators of layer L): y[o] = sum_c h_L[c] wf[o][c] + bf[o]
// (modules/wire.py:156-157,164-165: the image holds (W_re | -W_im), i.e. the real part of the complex product)
template <int KIND, int NB>
WIRE_DEVINL void fx_final(const FusedFwdParams& fp, const FxCtx& c, const f32x4 (&src)[NB]) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int U = FxKind<KIND>::U, NU = NB / U, P = 16 * NB;
  float yo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    fx_f16x8 dh[U / 2], dl[U / 2];
    f32x4 of[U];
    fx_produce<KIND, NB, false, true>(fp, c, src, u, fp.L, 1.f, dh, dl, of);
#pragma unroll
    for (int b = 0; b < U; ++b) {
      const int col = CPLX ? 64 * u + 32 * (b >> 1) + 16 * (b & 1) + 4 * c.g : 16 * (2 * u + b) + 4 * c.g;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (o < fp.O) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(c.swf + o * P + col);
#pragma unroll
          for (int q = 0; q < 4; ++q) yo[o] = __builtin_fmaf(of[b][q], wv[q], yo[o]);
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
  return RING * NB * 2048 + (4 * 16 * NB + 16 * NB + FX_LMAX * 16 * NB + 4 * 16 * NB) * 4;   // (first layer sized for PF = P)
}

template <int KIND, int NB, int RING>
__global__ __launch_bounds__(64 * FX_WAVES) void fused_fwd_kernel(const FusedFwdParams fp) {
  constexpr bool CPLX = FxKind<KIND>::CPLX;
  constexpr int P = 16 * NB, PF = CPLX ? P / 2 : P, S = NB / 2, STAGE = NB * 2048;
  static_assert(NB % FxKind<KIND>::U == 0 && NB % 4 == 0, "whole epilogue units, whole DMA rounds");
  extern __shared__ __attribute__((aligned(1024))) unsigned char fx_smem[];
  float* const sW0 = reinterpret_cast<float*>(fx_smem + RING * STAGE);
  float* const sb0 = sW0 + 4 * P;
  float* const sbias = sb0 + P;
  float* const swf = sbias + FX_LMAX * P;
  const int tid = threadIdx.x;
  FxCtx c;
  c.lane = tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c.g = c.lane >> 4;
  c.n = fp.n;
  c.row = (long long)blockIdx.x * FX_ROWS + c.wave * 16 + (c.lane & 15);
  c.ring = fx_smem;
  c.sW0 = sW0; c.sb0 = sb0; c.sbias = sbias; c.swf = swf;
  c.t = 0; c.T = fp.L * S; c.buf = 0;
  c.inv_prev = 1.f;
  // the weight stream starts first: it does not depend on anything below
  fx_issue<NB, RING>(fp, c, 0, 0);
  if (RING == 3 && c.T > 1) fx_issue<NB, RING>(fp, c, 1, 1);
  {
    const long long rowc = c.row < c.n ? c.row : c.n - 1;
#pragma unroll
    for (int d = 0; d < 4; ++d) c.x[d] = d < fp.D ? fp.coords[rowc * fp.D + d] : 0.f;
  }
  for (int i = tid; i < 4 * PF; i += 64 * FX_WAVES) {
    const int d = i / PF, f = i - d * PF;
    sW0[i] = (d < fp.D && f < fp.K) ? fp.W0[f * fp.D + d] : 0.f;
  }
  for (int i = tid; i < PF; i += 64 * FX_WAVES) sb0[i] = i < fp.K ? fp.b0[i] : 0.f;
  for (int i = tid; i < fp.L * P; i += 64 * FX_WAVES) {
    const int l = i / P;
    sbias[i] = fp.bias[(size_t)l * fp.bias_stride + (i - l * P)];
  }
  for (int i = tid; i < fp.O * P; i += 64 * FX_WAVES) swf[i] = fp.wf[i];
  __syncthreads();

  f32x4 accA[NB], accB[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) accB[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  fx_layer<KIND, NB, RING, true>(fp, c, accB, accA, 1);
  int l = 2;
  for (; l + 1 <= fp.L; l += 2) {
    fx_layer<KIND, NB, RING, false>(fp, c, accA, accB, l);
    fx_layer<KIND, NB, RING, false>(fp, c, accB, accA, l + 1);
  }
  if (l <= fp.L) {
    fx_layer<KIND, NB, RING, false>(fp, c, accA, accB, l);
    fx_final<KIND, NB>(fp, c, accB);
  } else {
    fx_final<KIND, NB>(fp, c, accA);
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
int fused_tune_get(const char* key) {
  if (!strcmp(key, "fused_fwd")) return g_fused_fwd;
  return -1;
}
int fused_tune_set(const char* key, int value) {
  if (!strcmp(key, "fused_fwd") && value >= 0 && value <= 1) { g_fused_fwd = value; return 0; }
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

template <int KIND, int NB, int RING>
static hipError_t fx_launch_t(hipStream_t s, const FusedFwdParams& fp) {
  constexpr int LDS = fx_lds_bytes<NB, RING>();
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_kernel<KIND, NB, RING>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr != hipSuccess) return attr;
  const unsigned grid = (unsigned)((fp.n + FX_ROWS - 1) / FX_ROWS);
  hipLaunchKernelGGL((fused_fwd_kernel<KIND, NB, RING>), dim3(grid), dim3(64 * FX_WAVES), LDS, s, fp);
  return hipGetLastError();
}

hipError_t launch_fused_fwd(hipStream_t s, int kind, int P, const FusedFwdParams& fp) {
  if (fp.n <= 0) return hipSuccess;
  if (!fused_fwd_shape(kind, P) || fp.L < 1 || fp.L > FX_LMAX || fp.D < 1 || fp.D > 4 || fp.O < 1 || fp.O > 4 ||
      fp.n > 0x7fffffffLL * FX_ROWS)
    return hipErrorInvalidValue;
  if (kind != NK_RELU && !(fp.a_scale > 0.f)) return hipErrorInvalidValue;
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
