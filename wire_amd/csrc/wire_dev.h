// wire_dev.h -- device-side helpers shared by the gfx950 kernels.
//
// Everything here is wave64 / CDNA4 code; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WIRE_DEVINL __device__ __forceinline__

// ---- blocked-planar complex layout --------------------------------------
// feature o, part c (0 = re, 1 = im) -> column in a row of P floats.
WIRE_DEVINL int blk_col(int o, int part) { return ((o >> 5) << 6) + (part << 5) + (o & 31); }
// inverse: column -> (feature, part)
WIRE_DEVINL void blk_decode(int col, int& o, int& part) {
  o = ((col >> 6) << 5) + (col & 31);
  part = (col >> 5) & 1;
}

// ---- transcendental kernels ---------------------------------------------
// exp(a) as exp2 on the hardware unit (v_exp_f32) with the rounding error of
// a*log2(e) folded back in: ~1-2 ulp for every a the Gabor envelope produces.
WIRE_DEVINL float wire_exp(float a) {
  const float L2E_HI = 1.44269502e+00f;   // float(log2 e)
  const float L2E_LO = 1.92596299e-08f;   // log2 e - L2E_HI
  const float LN2 = 6.93147182e-01f;
  float t = a * L2E_HI;
  float tl = __builtin_fmaf(a, L2E_HI, -t) + a * L2E_LO;
  float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, tl * LN2, e);
}

// sin and cos of x with a 3-constant Cody-Waite reduction by pi/2 (fma keeps
// every partial product unrounded) and cephes-grade minimax polynomials on
// [-pi/4, pi/4].  Arguments here are omega0*u with |x| well below 1e5, where
// the reduction error stays under 1 ulp of the reduced argument.
WIRE_DEVINL void wire_sincos(float x, float& sn, float& cs) {
  const float TWO_OVER_PI = 6.36619747e-01f;
  const float C1 = 1.57079637e+00f;    // float(pi/2)
  const float C2 = -4.37113883e-08f;   // float(pi/2 - C1)
  const float C3 = -1.71512451e-15f;   // pi/2 - C1 - C2
  float n = __builtin_rintf(x * TWO_OVER_PI);
  float r = __builtin_fmaf(-n, C1, x);
  r = __builtin_fmaf(-n, C2, r);
  r = __builtin_fmaf(-n, C3, r);
  int q = (int)n;
  float z = r * r;
  float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
  float s = __builtin_fmaf(sp * z, r, r);
  float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
  float c = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  float a = (q & 1) ? c : s;
  float b = (q & 1) ? s : c;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}

// ---- activation math (shared by the GEMM epilogues and the point kernels) --
// Complex Gabor, modules/wire.py:90-93:  out = exp(j w0 lin - |s0 lin|^2)
WIRE_DEVINL void gabor_fwd(float u, float v, float w0, float s0, float& o_re, float& o_im) {
  float su = s0 * u, sv = s0 * v;
  float ar = -(w0 * v) - __builtin_fmaf(su, su, sv * sv);
  float e = wire_exp(ar);
  float sn, cs;
  wire_sincos(w0 * u, sn, cs);
  o_re = e * cs;
  o_im = e * sn;
}
// ---- lean variants for the GEMM epilogues -----------------------------------------------
// On gfx950 v_mfma_f32_32x32x2_f32 and fp32 VALU instructions do NOT overlap (measured with
// tools/mfma_valu_probe.hip: every v_fma beside the MFMA stream costs ~5 cycles of it, every
// v_exp/v_sin ~9), so an epilogue instruction is paid for in matrix-core time.  These versions
// use the hardware transcendental unit (v_sin/v_cos take revolutions) with a two-term 1/(2 pi)
// reduction: max abs error 4e-7 for |x| <= 200 (tools/trig_probe.hip) against 9e-8 for the
// polynomial path -- far inside the 1e-5 per-layer parity bar -- at a third of the issue cost.
WIRE_DEVINL void wire_sincos_hw(float x, float& sn, float& cs) {
  const float INV2PI_HI = 0.15915494f;       // float(1 / 2 pi)
  const float INV2PI_LO = 6.42063833e-9f;    // 1 / 2 pi - INV2PI_HI
  const float t = x * INV2PI_HI;
  float e = __builtin_fmaf(x, INV2PI_HI, -t);
  e = __builtin_fmaf(x, INV2PI_LO, e);
  const float r = __builtin_amdgcn_fractf(t) + e;
  sn = __builtin_amdgcn_sinf(r);
  cs = __builtin_amdgcn_cosf(r);
}
// neg_s2 = -s0^2, w0l2e = w0 * log2(e), ns2l2e = -s0^2 * log2(e): the envelope exponent is formed
// directly in the base-2 domain (one v_exp_f32, no compensation: error <= 2.4e-7 of max(1, value)).
WIRE_DEVINL void gabor_fwd_lean(float u, float v, float w0, float w0l2e, float ns2l2e, float& o_re,
                                float& o_im) {
  const float q = __builtin_fmaf(u, u, v * v);
  const float t = __builtin_fmaf(ns2l2e, q, -(w0l2e * v));   // log2 of the envelope
  const float e = __builtin_amdgcn_exp2f(t);
  float sn, cs;
  wire_sincos_hw(w0 * u, sn, cs);
  o_re = e * cs;
  o_im = e * sn;
}

// first layer: lin is real (v = 0)
WIRE_DEVINL void gabor_fwd_real(float u, float w0, float s0, float& o_re, float& o_im) {
  float su = s0 * u;
  float e = wire_exp(-(su * su));
  float sn, cs;
  wire_sincos(w0 * u, sn, cs);
  o_re = e * cs;
  o_im = e * sn;
}
// backward, SURVEY 8(a) row a4: c = conj(out) g, P = Re c,
//   g_lin = -2 s0^2 P lin - j w0 c
WIRE_DEVINL void gabor_bwd(float gr, float gi, float u, float v, float o_re, float o_im,
                           float w0, float m2s2, float& gl_re, float& gl_im) {
  float c_re = __builtin_fmaf(o_re, gr, o_im * gi);
  float c_im = __builtin_fmaf(o_re, gi, -(o_im * gr));
  float t = m2s2 * c_re;
  gl_re = __builtin_fmaf(t, u, w0 * c_im);
  gl_im = __builtin_fmaf(t, v, -(w0 * c_re));
}
// first layer: g_u = -2 s0^2 u P + w0 Im c
WIRE_DEVINL float gabor_bwd_real(float gr, float gi, float u, float o_re, float o_im,
                                 float w0, float m2s2) {
  float c_re = __builtin_fmaf(o_re, gr, o_im * gi);
  float c_im = __builtin_fmaf(o_re, gi, -(o_im * gr));
  return __builtin_fmaf(m2s2 * c_re, u, w0 * c_im);
}

// 2-D Gabor, modules/wire2d.py:56-67:
//   out = exp(j w0 lin) exp(-s0^2 (|lin|^2 + |sy|^2))
WIRE_DEVINL void gabor2d_fwd(float u, float v, float p, float q, float w0, float s0,
                             float& o_re, float& o_im) {
  // exp(j w0 (u + j v)) = exp(-w0 v) (cos w0 u + j sin w0 u)
  float arg = __builtin_fmaf(u, u, v * v) + __builtin_fmaf(p, p, q * q);
  float e = wire_exp(-(w0 * v) - (s0 * s0) * arg);
  float sn, cs;
  wire_sincos(w0 * u, sn, cs);
  o_re = e * cs;
  o_im = e * sn;
}

// real sweep activations (config 5)
enum { ACT_SIREN = 0, ACT_GAUSS = 1, ACT_RELU = 2 };
template <int ACT>
WIRE_DEVINL float real_act_fwd(float lin, float w0, float s0) {
  if (ACT == ACT_SIREN) {            // modules/siren.py:48-49
    float sn, cs;
    wire_sincos(w0 * lin, sn, cs);
    return sn;
  } else if (ACT == ACT_GAUSS) {     // modules/gauss.py:27-28
    float t = s0 * lin;
    return wire_exp(-(t * t));
  } else {                           // modules/relu.py:28-29
    return lin > 0.f ? lin : 0.f;
  }
}
template <int ACT>
WIRE_DEVINL float real_act_bwd(float g, float lin, float out, float w0, float s0) {
  if (ACT == ACT_SIREN) {
    float sn, cs;
    wire_sincos(w0 * lin, sn, cs);
    return g * w0 * cs;
  } else if (ACT == ACT_GAUSS) {
    return g * out * (-2.f * s0 * s0) * lin;
  } else {
    return out > 0.f ? g : 0.f;        // out = max(lin, 0): out > 0 <=> lin > 0, so relu never needs lin stored
  }
}

// lean forms for the split-bf16 GEMM epilogues: hardware transcendentals (v_sin / v_cos / v_exp), same error
// bounds as gabor_fwd_lean (4e-7 abs for the trigonometric part, 2.4e-7 relative for the exponential)
template <int ACT>
WIRE_DEVINL float real_act_fwd_lean(float lin, float w0, float s0) {
  if (ACT == ACT_SIREN) {
    float sn, cs;
    wire_sincos_hw(w0 * lin, sn, cs);
    return sn;
  } else if (ACT == ACT_GAUSS) {
    const float t = s0 * lin;
    return __builtin_amdgcn_exp2f(-(t * t) * 1.44269502f);
  } else {
    return lin > 0.f ? lin : 0.f;
  }
}
template <int ACT>
WIRE_DEVINL float real_act_bwd_lean(float g, float lin, float out, float w0, float s0) {
  if (ACT == ACT_SIREN) {
    float sn, cs;
    wire_sincos_hw(w0 * lin, sn, cs);
    return g * w0 * cs;
  } else if (ACT == ACT_GAUSS) {
    return g * out * (-2.f * s0 * s0) * lin;
  } else {
    return out > 0.f ? g : 0.f;
  }
}

// 2-D Gabor, lean form (hardware v_exp / v_sin / v_cos, as gabor_fwd_lean)
WIRE_DEVINL void gabor2d_fwd_lean(float u, float v, float p, float q, float w0, float s0, float& o_re,
                                  float& o_im) {
  const float arg = __builtin_fmaf(u, u, v * v) + __builtin_fmaf(p, p, q * q);
  const float t = __builtin_fmaf(-(s0 * s0) * 1.44269502f, arg, -(w0 * 1.44269502f) * v);   // log2 of the envelope
  const float e = __builtin_amdgcn_exp2f(t);
  float sn, cs;
  wire_sincos_hw(w0 * u, sn, cs);
  o_re = e * cs;
  o_im = e * sn;
}

// ---- power-of-two operand scales of the 2 x fp16 split GEMMs (wire_gemmx2h.hip) ------------------------------
// Every tensor that such a GEMM reads as an operand has WIRE_AMAX_SLOTS sharded slots holding the bit pattern of its
// max |value| (unsigned compare = float compare for non-negative floats), filled by its producer with atomicMax and
// zeroed by the host sequence beforehand.  The consumer maps the maximum into [2^14, 2^15) -- below the fp16 overflow
// limit 65504, far above the fp16 subnormal range -- by an exact power of two.
#define WIRE_AMAX_SLOTS 64
WIRE_DEVINL void wire_amax_publish(unsigned* slots, float m, int lane) {
#pragma unroll
  for (int o = 32; o; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) {
    const unsigned bits = __float_as_uint(m);
    unsigned* slot = slots + (blockIdx.x & (WIRE_AMAX_SLOTS - 1));
    // a stale (smaller) value only costs a redundant atomic: the maximum is monotone
    if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
  }
}
// every lane returns the maximum over the slots (the producer finished at a kernel boundary)
WIRE_DEVINL unsigned wire_amax_read(const unsigned* slots, int lane) {
  unsigned m = slots[lane & (WIRE_AMAX_SLOTS - 1)];
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = m > t ? m : t;
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
// scale 2^(14 - e) for a maximum in [2^e, 2^(e+1)) and its inverse; 1 for an all-zero, subnormal-only, vanishing
// (< 2^-112) or non-finite tensor
WIRE_DEVINL void wire_x2_scales(unsigned maxbits, float& s, float& inv) {
  const int E = (int)(maxbits >> 23);
  if (E < 15 || E > 254) { s = 1.f; inv = 1.f; return; }
  s = __uint_as_float((unsigned)(268 - E) << 23);
  inv = __uint_as_float((unsigned)(E - 14) << 23);
}

// ---- the split itself ------------------------------------------------------------------------------
typedef _Float16 wire_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wire_u32x4 __attribute__((ext_vector_type(4)));
WIRE_DEVINL unsigned x2_cvt_pk(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wire_f16x2));   // v_cvt_pk_f16_f32 (round to nearest)
}
// (x0, x1) s = H + L in packed fp16 pairs (low half = first element): 6 vector ops
WIRE_DEVINL void x2_split2(float x0, float x1, float s, unsigned& H, unsigned& L) {
  H = x2_cvt_pk(x0 * s, x1 * s);
  float r0, r1;
  // x s - h in one instruction each: the fp16 half is converted on the fly, the product is not rounded
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(s), "v"(H));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(s), "v"(H));
  L = x2_cvt_pk(r0, r1);
}
// PRE-SPLIT activations ("split_out").  A tensor whose max |value| is bounded a priori (Gabor / sine / Gaussian outputs)
// can be split by its PRODUCER with a scale fixed on the host, once per element, instead of by every GEMM that reads it
// (the forward GEMM of the next layer once per 128-column tile -- 4 x at K = 256 -- and the weight-gradient GEMM once per
// 256-column tile).  Layout: same rows, same leading dimension, same 16 bytes per 4 consecutive columns -- holding
//     [h(c) h(c+1) h(c+2) h(c+3) | l(c) .. l(c+3)]   (fp16 each)    instead of   [x(c) .. x(c+3)]   (fp32 each)
// which is the unit every producer writes (one 16-byte store per lane) and every consumer loads.
WIRE_DEVINL void wire_store_out4(float* p, const f32x4& o, const float split_scale) {
  if (split_scale != 0.f) {
    unsigned h0, l0, h1, l1;
    x2_split2(o[0], o[1], split_scale, h0, l0);
    x2_split2(o[2], o[3], split_scale, h1, l1);
    *reinterpret_cast<wire_u32x4*>(p) = wire_u32x4{h0, h1, l0, l1};
  } else {
    *reinterpret_cast<f32x4*>(p) = o;
  }
}
