// wire_gemmx3h.hip -- split-bf16 NT GEMM on v_mfma_f32_16x16x32_bf16 (large batches of the wire kernels).
//
// Why another edition.  The split-bf16 GEMMs are bound by the rate at which the chip executes bf16 MFMAs on random
// data under its power management, not by staging (DESIGN.md 4.1 "Round 2": three data paths, one time).  On gfx950
// the 16 x 16 x 32 shape sustains more flops at that cap than the 32 x 32 x 16 one (MI355X guide: 1.12 - 1.15 x at
// equal cycles; probed on THIS kernel's instruction mix with tools/gemm_glds_tune16: 0.69 -> 0.60 ms per launch,
// MFMA-only floor 0.53 -> 0.46 ms).
//
// How a 16-deep stage feeds a 32-deep instruction without doubling the stage.  A 16 x 16 x 32 MFMA sums over four
// 8-k slots, slot = lane >> 4.  Lanes 0-31 (slots 0, 1) carry the 16 k of the stage for one plane pair, lanes 32-63
// (slots 2, 3) the SAME 16 k for another plane pair, so one instruction computes TWO of the six partial products:
//     MFMA 1:  A = (a_h | a_l)   B = (b_l | b_h)   ->  h l + l h
//     MFMA 2:  A = (a_m | a_h)   B = (b_m | b_m)   ->  m m + h m
//     MFMA 3:  A = (a_m | a_h)   B = (b_h | b_h)   ->  m h + h h        (lower half | upper half)
// Three instructions of 16 cycles per 16 x 16 block and stage = the same matrix-core cycles as before.
//  * B planes lie pre-split in LDS: the three B fragments are three ds_read_b128 with a per-half plane offset.
//  * A is split on the fragment, in registers, as in wire_gemmx3g.hip -- without redundancy: the lower half of the
//    wave loads and splits rows 0-15 of a 32-row block, the upper half rows 16-31, and v_permlane32_swap_b32
//    exchanges planes between the halves: swap(H, L) gives (h0 | l0) and (h1 | l1), swap(M, H) gives (m0 | h0) and
//    (m1 | h1) -- the A operands of both 16-row blocks from one split each.  112 vector ops per wave and stage (88
//    split + 24 exchange) beside 96 MFMAs.
//  * staging as in wire_gemmx3g.hip mode 1: both operands by LDS-DMA, 256 x 128 tile, 4 waves stacked in M (64 rows
//    x 128 columns each), two 28 KB buffers, 2 workgroups per CU, one raw s_barrier per stage.  The fp32 rows use a
//    Gray-code chunk swizzle and the weight planes the UNSWIZZLED second half of the split image: both are
//    conflict-free for the 16-lane groups of a 16 x 16 x 32 fragment read (wire_gemmx3.hip: x3_split_b_kernel).
//  * the MFMAs take the weights as their first operand, so a 16 x 16 block leaves the accumulator transposed: lane ->
//    activation row lane & 15, four consecutive columns per lane.  Own epilogues below: same arithmetic as
//    wire_gemm_epi.h, but every global access is a 16-byte vector (4 x fewer memory instructions).
//
// Accuracy: the same six products, each exact in fp32, summed in fp32 inside the instruction; the grouping differs
// from the 32 x 32 x 16 path, so results agree to fp32 round-off, not bit for bit.  All parity tests run on it.
//
// Replaces the ATen complex addmm / mm of modules/wire.py:89 and of its autograd backward at M >= 4096.
#include <cstdlib>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"

typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));

#define H_BK 16
#define H_TBM 256
#define H_TBN 128
#define H_ABYTES (H_TBM * 64)
#define H_BPLANE (H_TBN * 32)
#define H_STAGE (H_ABYTES + 3 * H_BPLANE)
#define H_PPW 7                      // (16 + 12) 1-KiB pieces per stage / 4 waves

WIRE_DEVINL unsigned h_cvt_pk_bf16(float a, float b) {
  const hbf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
WIRE_DEVINL float h_bf16_lo(unsigned packed) {
  unsigned r;
  asm("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(packed));
  return __uint_as_float(r);
}
WIRE_DEVINL void h_split2(float x0, float x1, unsigned& H, unsigned& Mi, unsigned& L) {
  H = h_cvt_pk_bf16(x0, x1);
  const float r0 = x0 - h_bf16_lo(H), r1 = x1 - __uint_as_float(H & 0xffff0000u);
  Mi = h_cvt_pk_bf16(r0, r1);
  const float q0 = r0 - h_bf16_lo(Mi), q1 = r1 - __uint_as_float(Mi & 0xffff0000u);
  L = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
// lanes 32-63 of `a` <-> lanes 0-31 of `b`
WIRE_DEVINL void h_swap_halves(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
WIRE_DEVINL int h_gray(int x) { return x ^ (x >> 1); }

#define H_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

WIRE_DEVINL void h_dma16(const void* gsrc, unsigned char* lds_piece) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// epilogues.  The MFMAs are issued with the operands SWAPPED (weights as the "A" operand, activations as "B"), so a
// 16 x 16 block leaves the accumulator transposed: lane -> activation ROW lane & 15, registers q = 0..3 -> four
// CONSECUTIVE output columns 4 (lane >> 4) + q.  Stored like that a wave instruction would write 16 rows x 64 bytes
// -- half cache lines.  Two column-adjacent blocks X (columns 0-15) and Y (16-31) are therefore re-paired inside each
// 16-lane row with two DPP moves per register (row_ror:8 under a bank mask):
//     X' = lanes 0-7: X of rows 0-7,  lanes 8-15: Y of rows 0-7        Y' = lanes 0-7: X of rows 8-15, lanes 8-15: Y of rows 8-15
// after which lane (rr = lane & 7, ch = (lane >> 3) & 1, g = lane >> 4) of X' holds row rr, columns 16 ch + 4 g + q:
// eight lanes x 16 bytes = one whole 128-byte line per row, eight rows per instruction, every access a 16-byte
// vector: full-line traffic with 4 x fewer memory instructions than the dword-per-lane epilogues of the 32 x 32
// kernels.  Same arithmetic as wire_gemm_epi.h (lean Gabor forms of wire_dev.h).
// ---------------------------------------------------------------------------
WIRE_DEVINL void h_pair_rows(const f32x4& X, const f32x4& Y, f32x4& Xp, f32x4& Yp) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = __float_as_int(X[q]), y = __float_as_int(Y[q]);
    // row_ror:8 = 0x128; bank_mask 0xC writes lanes 8-15 of every 16-lane row, 0x3 lanes 0-7
    Xp[q] = __int_as_float(__builtin_amdgcn_update_dpp(x, y, 0x128, 0xf, 0xc, false));
    Yp[q] = __int_as_float(__builtin_amdgcn_update_dpp(y, x, 0x128, 0xf, 0x3, false));
  }
}

template <int EPI>
WIRE_DEVINL void h_epilogue(f32x4 (&acc)[4][8], const GemmEpiParams& ep, const int M, const int m_w, const int n_w,
                            const int Nc, const int lane, unsigned char* lds, const int wave, const int rt) {
  const int rr = lane & 7, ch = (lane >> 3) & 1, g = lane >> 4;
  const int cq = 16 * ch + 4 * g;                                 // this lane's first column inside a 32-column span
  if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {                              // 32-column spans of the 128-column wave tile
      const int col = n_w + 32 * sp + cq;
      if (n_w + 32 * sp >= Nc) continue;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 xp, yp;
        h_pair_rows(acc[rb][2 * sp], acc[rb][2 * sp + 1], xp, yp);
        const int row = m_w + 16 * rb + rr;
        if (row < M) *reinterpret_cast<f32x4*>(ep.o0 + (size_t)row * ep.ld0 + col) = xp;
        if (row + 8 < M) *reinterpret_cast<f32x4*>(ep.o0 + (size_t)(row + 8) * ep.ld0 + col) = yp;
      }
    }
  } else if constexpr (EPI == EPI_SIREN_FWD || EPI == EPI_GAUSS_FWD || EPI == EPI_RELU_FWD) {
    constexpr int ACT = EPI - EPI_SIREN_FWD;
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      if (n_w + 32 * sp >= Nc) continue;
      const int col = n_w + 32 * sp + cq;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + col);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 a2[2];
        h_pair_rows(acc[rb][2 * sp], acc[rb][2 * sp + 1], a2[0], a2[1]);
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          const int row = m_w + 16 * rb + 8 * hr + rr;
          f32x4 lin;
#pragma unroll
          for (int q = 0; q < 4; ++q) lin[q] = a2[hr][q] + bv[q];
          if (ep.o0 && row < M) *reinterpret_cast<f32x4*>(ep.o0 + (size_t)row * ep.ld0 + col) = lin;
          if (ep.o1) {                                             // null: last hidden layer of a fused training step
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = real_act_fwd_lean<ACT>(lin[q], ep.omega, ep.scale);
              o[q] = col + q < ep.kvalid ? v : 0.f;                // pad features are written as 0
            }
            if (row < M) *reinterpret_cast<f32x4*>(ep.o1 + (size_t)row * ep.ld1 + col) = o;
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_SIREN_BWD || EPI == EPI_GAUSS_BWD || EPI == EPI_RELU_BWD) {
    constexpr int ACT = EPI - EPI_SIREN_BWD;
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      if (n_w + 32 * sp >= Nc) continue;
      const int col = n_w + 32 * sp + cq;
      float crs[4][5];                                            // cr_partial: sums of g_lin_0 [x | 1] over this lane's rows
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 5; ++d) crs[q][d] = 0.f;
#pragma unroll
      for (int rp = 0; rp < 4; rp += 2) {
        f32x4 lv[4], ov[4], a2[4];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          h_pair_rows(acc[rp + r2][2 * sp], acc[rp + r2][2 * sp + 1], a2[2 * r2], a2[2 * r2 + 1]);
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            int row = m_w + 16 * (rp + r2) + 8 * hr + rr;
            row = row < M ? row : M - 1;
            // siren needs lin, relu needs out (its lin is never stored), gauss both
            lv[2 * r2 + hr] = (ACT != ACT_RELU) ? *reinterpret_cast<const f32x4*>(ep.i0 + (size_t)row * ep.ld0 + col)
                                                : f32x4{0.f, 0.f, 0.f, 0.f};
            const bool load_out = ACT == ACT_RELU || (ACT == ACT_GAUSS && !ep.recompute_out);
            ov[2 * r2 + hr] = load_out ? *reinterpret_cast<const f32x4*>(ep.i1 + (size_t)row * ep.ld1 + col)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        if (ACT == ACT_GAUSS && ep.recompute_out) {                // out = exp(-(s0 lin)^2) again: the forward's lean form
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = real_act_fwd_lean<ACT>(lv[e][q], ep.omega, ep.scale);
              ov[e][q] = col + q < ep.kvalid ? v : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = m_w + 16 * (rp + (e >> 1)) + 8 * (e & 1) + rr;
          f32x4 gl;
#pragma unroll
          for (int q = 0; q < 4; ++q) gl[q] = real_act_bwd_lean<ACT>(a2[e][q], lv[e][q], ov[e][q], ep.omega, ep.scale);
          if (row < M) {
            if (ep.cr_partial) {
              // first layer of a real net: its weight / bias gradient sums g_lin_0 [x | 1] are formed here (what
              // colreduce_kernel would form from the stored g_lin_0 -- which is then never written)
              float x[4] = {0.f, 0.f, 0.f, 0.f};
              for (int d = 0; d < ep.D; ++d) x[d] = ep.coords[(size_t)row * ep.D + d];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int d = 0; d < 4; ++d) crs[q][d] = __builtin_fmaf(gl[q], x[d], crs[q][d]);
                crs[q][4] += gl[q];
              }
            } else {
              *reinterpret_cast<f32x4*>(ep.o0 + (size_t)row * ep.ld0 + col) = gl;
            }
          }
        }
      }
      if (ep.cr_partial) {                                         // 8 row lanes by shuffles; the 4 waves through LDS below
        float* red = reinterpret_cast<float*>(lds);                // [4 waves][128 columns][5]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            float v = crs[q][d];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            if (rr == 0) red[(wave * 128 + 32 * sp + cq + q) * 5 + d] = v;
          }
      }
    }
    if (ep.cr_partial) {
      __syncthreads();
      const float* red = reinterpret_cast<const float*>(lds);
      for (int e = threadIdx.x; e < 128 * 5; e += 256) {
        const int c = e / 5, d = e - 5 * c;
        const float v = (red[e] + red[128 * 5 + e]) + (red[2 * 128 * 5 + e] + red[3 * 128 * 5 + e]);
        if (n_w + c < ep.cr_C) ep.cr_partial[((size_t)rt * ep.cr_C + n_w + c) * 5 + d] = v;
      }
    }
  } else if constexpr (EPI == EPI_GABOR2D_FWD) {
    // the wave's 128 columns = (lin_re | lin_im | sy_re | sy_im) of 32 features (modules/wire2d.py:56-67)
    const int grp = n_w >> 7;
    const int f0 = (grp << 5) + cq;                               // features f0 .. f0 + 3
    const int oc = (grp << 6) + cq;                               // re column in the P-wide output row; im = + 32
    f32x4 bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bv[k] = *reinterpret_cast<const f32x4*>(ep.bias + n_w + 32 * k + cq);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      f32x4 part[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) h_pair_rows(acc[rb][2 * k], acc[rb][2 * k + 1], part[k][0], part[k][1]);
#pragma unroll
      for (int hr = 0; hr < 2; ++hr) {
        const int row = m_w + 16 * rb + 8 * hr + rr;
        f32x4 u, v, pp, qq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u[q] = part[0][hr][q] + bv[0][q];
          v[q] = part[1][hr][q] + bv[1][q];
          pp[q] = part[2][hr][q] + bv[2][q];
          qq[q] = part[3][hr][q] + bv[3][q];
        }
        if (ep.o0 && row < M) {
          float* Lp = ep.o0 + (size_t)row * ep.ld0 + n_w + cq;
          *reinterpret_cast<f32x4*>(Lp) = u;
          *reinterpret_cast<f32x4*>(Lp + 32) = v;
          *reinterpret_cast<f32x4*>(Lp + 64) = pp;
          *reinterpret_cast<f32x4*>(Lp + 96) = qq;
        }
        if (ep.o1) {                                              // null: last hidden layer of a fused training step
          f32x4 o_re, o_im;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a, b;
            gabor2d_fwd_lean(u[q], v[q], pp[q], qq[q], ep.omega, ep.scale, a, b);
            const bool valid = f0 + q < ep.kvalid;
            o_re[q] = valid ? a : 0.f;
            o_im[q] = valid ? b : 0.f;
          }
          if (row < M) {
            float* Op = ep.o1 + (size_t)row * ep.ld1 + oc;
            *reinterpret_cast<f32x4*>(Op) = o_re;
            *reinterpret_cast<f32x4*>(Op + 32) = o_im;
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_GABOR2D_BWD || EPI == EPI_GABOR2D_BWD_FIRST) {
    // C = g_out (re | im pairs, P wide); writes g_(lin | sy) into the 2P-wide row (or the real g_(u | p) of layer 0)
    const float m2s2 = -2.f * ep.scale * ep.scale, w0 = ep.omega;
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      if (n_w + 64 * G >= Nc) continue;
      const int c0 = n_w + 64 * G + cq;                           // re column of g_out / out; im = + 32
      const int grp = c0 >> 6;
      const int f0 = (grp << 5) + (c0 & 31);
      const int lc = (grp << 7) + (c0 & 31);                      // lin_re column in the 2P row
      float w[4][4], wv[4][4], bb[4], bv2[4];
      float cru[4][5], crp2[4][5];                                // BWD_FIRST with cr_partial: sums of g_u [x | 1], g_p [x | 1]
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 5; ++d) { cru[q][d] = 0.f; crp2[q][d] = 0.f; }
      if constexpr (EPI == EPI_GABOR2D_BWD_FIRST) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool valid = f0 + q < ep.kvalid;
          bb[q] = valid ? ep.b0[f0 + q] : 0.f;
          bv2[q] = valid ? ep.b0b[f0 + q] : 0.f;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            w[q][d] = (valid && d < ep.D) ? ep.W0[(f0 + q) * ep.D + d] : 0.f;
            wv[q][d] = (valid && d < ep.D) ? ep.W0b[(f0 + q) * ep.D + d] : 0.f;
          }
        }
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 are[2], aim[2];
        h_pair_rows(acc[rb][4 * G], acc[rb][4 * G + 1], are[0], are[1]);
        h_pair_rows(acc[rb][4 * G + 2], acc[rb][4 * G + 3], aim[0], aim[1]);
        f32x4 u[2], v[2], pp[2], qq[2], pr[2], pi[2];
        float xs[2][4];                                           // BWD_FIRST: the rows' coordinates
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          int row = m_w + 16 * rb + 8 * hr + rr;
          row = row < M ? row : M - 1;
          const bool recompute = EPI == EPI_GABOR2D_BWD && ep.recompute_out != 0;
          if (!recompute) {
            const float* Op = ep.i1 + (size_t)row * ep.ld1 + c0;
            pr[hr] = *reinterpret_cast<const f32x4*>(Op);
            pi[hr] = *reinterpret_cast<const f32x4*>(Op + 32);
          }
          if constexpr (EPI == EPI_GABOR2D_BWD) {
            const float* Lp = ep.i0 + (size_t)row * ep.ld0 + lc;
            u[hr] = *reinterpret_cast<const f32x4*>(Lp);
            v[hr] = *reinterpret_cast<const f32x4*>(Lp + 32);
            pp[hr] = *reinterpret_cast<const f32x4*>(Lp + 64);
            qq[hr] = *reinterpret_cast<const f32x4*>(Lp + 96);
            if (recompute) {                                      // out = act(lin, sy) again: the forward's lean form
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float a, b;
                gabor2d_fwd_lean(u[hr][q], v[hr][q], pp[hr][q], qq[hr][q], ep.omega, ep.scale, a, b);
                const bool valid = f0 + q < ep.kvalid;
                pr[hr][q] = valid ? a : 0.f;
                pi[hr][q] = valid ? b : 0.f;
              }
            }
          } else {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            for (int d = 0; d < ep.D; ++d) x[d] = ep.coords[(size_t)row * ep.D + d];
#pragma unroll
            for (int d = 0; d < 4; ++d) xs[hr][d] = x[d];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float uu = bb[q], p2 = bv2[q];
#pragma unroll
              for (int d = 0; d < 4; ++d) { uu = __builtin_fmaf(x[d], w[q][d], uu); p2 = __builtin_fmaf(x[d], wv[q][d], p2); }
              u[hr][q] = uu; pp[hr][q] = p2; v[hr][q] = 0.f; qq[hr][q] = 0.f;
            }
          }
        }
#pragma unroll
        for (int hr = 0; hr < 2; ++hr) {
          const int row = m_w + 16 * rb + 8 * hr + rr;
          f32x4 g0, g1, g2, g3;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float gr = are[hr][q], gi = aim[hr][q];
            const float c_r = __builtin_fmaf(pr[hr][q], gr, pi[hr][q] * gi);
            const float c_i = __builtin_fmaf(pr[hr][q], gi, -(pi[hr][q] * gr));
            const float tt = m2s2 * c_r;
            g0[q] = __builtin_fmaf(tt, u[hr][q], w0 * c_i);      // g_lin = -2 s^2 P lin - j w0 c  (real part / real first layer)
            g1[q] = __builtin_fmaf(tt, v[hr][q], -(w0 * c_r));
            g2[q] = tt * pp[hr][q];                               // g_sy = -2 s^2 P sy
            g3[q] = tt * qq[hr][q];
          }
          if (row < M) {
            if constexpr (EPI == EPI_GABOR2D_BWD) {
              float* Gp = ep.o0 + (size_t)row * ep.ld0 + lc;
              *reinterpret_cast<f32x4*>(Gp) = g0;
              *reinterpret_cast<f32x4*>(Gp + 32) = g1;
              *reinterpret_cast<f32x4*>(Gp + 64) = g2;
              *reinterpret_cast<f32x4*>(Gp + 96) = g3;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (!(f0 + q < ep.kvalid)) { g0[q] = 0.f; g2[q] = 0.f; }
              if (ep.cr_partial) {
                // first-layer gradient sums (both Linears of the layer) instead of a stored g_(u | p) and two passes
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                  for (int d = 0; d < 4; ++d) {
                    cru[q][d] = __builtin_fmaf(g0[q], xs[hr][d], cru[q][d]);
                    crp2[q][d] = __builtin_fmaf(g2[q], xs[hr][d], crp2[q][d]);
                  }
                  cru[q][4] += g0[q];
                  crp2[q][4] += g2[q];
                }
              } else {
                float* Gp = ep.o0 + (size_t)row * (2 * ep.ldu);
                *reinterpret_cast<f32x4*>(Gp + f0) = g0;
                *reinterpret_cast<f32x4*>(Gp + ep.ldu + f0) = g2;
              }
            }
          }
        }
      }
      if constexpr (EPI == EPI_GABOR2D_BWD_FIRST) {
        if (ep.cr_partial) {                                       // 8 row lanes by shuffles; the 4 waves through LDS below
          float* red = reinterpret_cast<float*>(lds);              // [2 sets][4 waves][64 features][5]
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int d = 0; d < 5; ++d) {
              float a = cru[q][d], b = crp2[q][d];
              a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
              b += __shfl_xor(b, 1); b += __shfl_xor(b, 2); b += __shfl_xor(b, 4);
              if (rr == 0) {
                red[(wave * 64 + 32 * G + cq + q) * 5 + d] = a;
                red[4 * 64 * 5 + (wave * 64 + 32 * G + cq + q) * 5 + d] = b;
              }
            }
        }
      }
    }
    if constexpr (EPI == EPI_GABOR2D_BWD_FIRST) {
      if (ep.cr_partial) {
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds);
        const int fbase = (n_w >> 6) << 5;                          // first feature of this 128-column tile
        for (int e = threadIdx.x; e < 2 * 64 * 5; e += 256) {
          const int set = e / (64 * 5), r = e - set * (64 * 5);
          const int f = r / 5, d = r - 5 * f;
          const float* rs = red + set * (4 * 64 * 5);
          const float v = (rs[r] + rs[64 * 5 + r]) + (rs[2 * 64 * 5 + r] + rs[3 * 64 * 5 + r]);
          if (fbase + f < ep.cr_C)
            ep.cr_partial[(size_t)set * ep.cr_set + ((size_t)rt * ep.cr_C + fbase + f) * 5 + d] = v;
        }
      }
    }
  } else {
    // complex epilogues: 64-column groups (re | im of 32 features) = blocks 4 G, 4 G + 1 (re) and 4 G + 2, 4 G + 3 (im)
    const float w0 = ep.omega, w0l2e = ep.omega * 1.44269502f, ns2l2e = -(ep.scale * ep.scale) * 1.44269502f;
    const float m2s2 = -2.f * ep.scale * ep.scale;
#pragma unroll
    for (int G = 0; G < 2; ++G) {
      if (n_w + 64 * G >= Nc) continue;
      const int c0 = n_w + 64 * G + cq;                           // this lane's first re column; im = + 32
      const int f0 = ((c0 >> 6) << 5) + (c0 & 31);                // ... = features f0 .. f0 + 3
      f32x4 b_re = {0.f, 0.f, 0.f, 0.f}, b_im = b_re;
      float w[4][4], bb[4];
      float crs[4][5];                                            // BWD_FIRST with cr_partial: sums of g_u [x | 1]
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 5; ++d) crs[q][d] = 0.f;
      if constexpr (EPI == EPI_GABOR_FWD) {
        b_re = *reinterpret_cast<const f32x4*>(ep.bias + c0);
        b_im = *reinterpret_cast<const f32x4*>(ep.bias + c0 + 32);
      }
      if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool valid = f0 + q < ep.kvalid;
          bb[q] = valid ? ep.b0[f0 + q] : 0.f;
#pragma unroll
          for (int d = 0; d < 4; ++d) w[q][d] = (valid && d < ep.D) ? ep.W0[(f0 + q) * ep.D + d] : 0.f;
        }
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 are[2], aim[2];                                     // [0]: rows 0-7 of the block, [1]: rows 8-15
        h_pair_rows(acc[rb][4 * G], acc[rb][4 * G + 1], are[0], are[1]);
        h_pair_rows(acc[rb][4 * G + 2], acc[rb][4 * G + 3], aim[0], aim[1]);
        if constexpr (EPI == EPI_GABOR_FWD) {
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            const int row = m_w + 16 * rb + 8 * hr + rr;
            f32x4 u, v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              u[q] = are[hr][q] + b_re[q];
              v[q] = aim[hr][q] + b_im[q];
            }
            if (ep.o0 && row < M) {
              float* Lp = ep.o0 + (size_t)row * ep.ld0 + c0;
              *reinterpret_cast<f32x4*>(Lp) = u;
              *reinterpret_cast<f32x4*>(Lp + 32) = v;
            }
            // o1 = null: the last hidden layer of a fused training step -- the final stage recomputes out from lin
            if (ep.o1) {
              f32x4 o_re, o_im;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float a, b;
                gabor_fwd_lean(u[q], v[q], w0, w0l2e, ns2l2e, a, b);
                const bool valid = f0 + q < ep.kvalid;             // pad features are written as 0
                o_re[q] = valid ? a : 0.f;
                o_im[q] = valid ? b : 0.f;
              }
              if (row < M) {
                float* Op = ep.o1 + (size_t)row * ep.ld1 + c0;
                *reinterpret_cast<f32x4*>(Op) = o_re;
                *reinterpret_cast<f32x4*>(Op + 32) = o_im;
              }
            }
          }
        } else if constexpr (EPI == EPI_GABOR_BWD) {
          const bool recompute = ep.recompute_out != 0;
          f32x4 lu[2], lv[2], pr[2], pi[2];
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {                          // all 8 loads of the row block first
            int row = m_w + 16 * rb + 8 * hr + rr;
            row = row < M ? row : M - 1;
            const float* Lp = ep.i0 + (size_t)row * ep.ld0 + c0;
            const float* Op = ep.i1 + (size_t)row * ep.ld1 + c0;
            lu[hr] = *reinterpret_cast<const f32x4*>(Lp);
            lv[hr] = *reinterpret_cast<const f32x4*>(Lp + 32);
            if (!recompute) {
              pr[hr] = *reinterpret_cast<const f32x4*>(Op);
              pi[hr] = *reinterpret_cast<const f32x4*>(Op + 32);
            }
          }
          if (recompute) {
            // out = act(lin) again instead of 8 more bytes per element from HBM: the same lean form as the forward
            // epilogue above, so the value equals the stored one bit for bit when that epilogue produced it
#pragma unroll
            for (int hr = 0; hr < 2; ++hr)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float a, b;
                gabor_fwd_lean(lu[hr][q], lv[hr][q], w0, w0l2e, ns2l2e, a, b);
                const bool valid = f0 + q < ep.kvalid;
                pr[hr][q] = valid ? a : 0.f;
                pi[hr][q] = valid ? b : 0.f;
              }
          }
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            const int row = m_w + 16 * rb + 8 * hr + rr;
            f32x4 gl_re, gl_im;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float gr = are[hr][q], gi = aim[hr][q];
              const float c_r = __builtin_fmaf(pr[hr][q], gr, pi[hr][q] * gi);
              const float c_i = __builtin_fmaf(pr[hr][q], gi, -(pi[hr][q] * gr));
              const float tt = m2s2 * c_r;
              gl_re[q] = __builtin_fmaf(tt, lu[hr][q], w0 * c_i);
              gl_im[q] = __builtin_fmaf(tt, lv[hr][q], -(w0 * c_r));
            }
            if (row < M) {
              float* Gp = ep.o0 + (size_t)row * ep.ld0 + c0;
              *reinterpret_cast<f32x4*>(Gp) = gl_re;
              *reinterpret_cast<f32x4*>(Gp + 32) = gl_im;
            }
          }
        } else {   // EPI_GABOR_BWD_FIRST: u recomputed from the coordinates (D <= 4); g_u [M][ldu] real
#pragma unroll
          for (int hr = 0; hr < 2; ++hr) {
            const int row = m_w + 16 * rb + 8 * hr + rr;
            if (row < M) {
              float x[4] = {0.f, 0.f, 0.f, 0.f};
              for (int d = 0; d < ep.D; ++d) x[d] = ep.coords[(size_t)row * ep.D + d];
              // out_0: read back, or (recompute_out) evaluated again from u with first_fwd_kernel's own form -- same
              // fmaf chain for u, same gabor_fwd_real: the same bits, 8 bytes per element less
              f32x4 pr = {0.f, 0.f, 0.f, 0.f}, pi = pr;
              if (!ep.recompute_out) {
                const float* Op = ep.i1 + (size_t)row * ep.ld1 + c0;
                pr = *reinterpret_cast<const f32x4*>(Op);
                pi = *reinterpret_cast<const f32x4*>(Op + 32);
              }
              f32x4 gu;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float u = bb[q];
#pragma unroll
                for (int d = 0; d < 4; ++d) u = __builtin_fmaf(x[d], w[q][d], u);
                if (ep.recompute_out) {
                  float a, b;
                  gabor_fwd_real(u, w0, ep.scale, a, b);
                  pr[q] = a; pi[q] = b;
                }
                const float v = gabor_bwd_real(are[hr][q], aim[hr][q], u, pr[q], pi[q], w0, m2s2);
                gu[q] = f0 + q < ep.kvalid ? v : 0.f;
              }
              if (ep.cr_partial) {
                // first-layer gradient sums of this lane's rows: g_u [x | 1] (what colreduce_kernel would form from
                // the stored g_u -- which is then never written)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                  for (int d = 0; d < 4; ++d) crs[q][d] = __builtin_fmaf(gu[q], x[d], crs[q][d]);
                  crs[q][4] += gu[q];
                }
              } else {
                *reinterpret_cast<f32x4*>(ep.o0 + (size_t)row * ep.ldu + f0) = gu;
              }
            }
          }
        }
      }
      if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
        if (ep.cr_partial) {
          // sum over the 8 row lanes of each 16-lane row (lanes differing in bits 0-2), then over the 4 waves through LDS
          // (the main loop is over: every wave has passed its last barrier, the stage buffers are free)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int d = 0; d < 5; ++d) {
              float v = crs[q][d];
              v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
              crs[q][d] = v;
            }
          float* red = reinterpret_cast<float*>(lds);            // [4 waves][64 features][5]
          const int fl = 32 * G + cq;                             // feature inside the tile's 64: 32 G + 16 ch + 4 g (+ q)
          if (rr == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int d = 0; d < 5; ++d) red[(wave * 64 + fl + q) * 5 + d] = crs[q][d];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int d = 0; d < 5; ++d) crs[q][d] = 0.f;
        }
      }
    }
    if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
      if (ep.cr_partial) {
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds);
        const int fbase = (n_w >> 6) << 5;                        // first complex feature of this 128-column tile
        for (int e = threadIdx.x; e < 64 * 5; e += 256) {
          const int f = e / 5, d = e - 5 * f;
          const float v = (red[e] + red[64 * 5 + e]) + (red[2 * 64 * 5 + e] + red[3 * 64 * 5 + e]);
          if (fbase + f < ep.cr_C) ep.cr_partial[((size_t)rt * ep.cr_C + fbase + f) * 5 + d] = v;
        }
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemmx3h_nt_kernel(const float* __restrict__ A, int lda,
                                                            const unsigned short* __restrict__ Bu, int M, int Nc,
                                                            int Kd, int tiles_m, int tiles_n, GemmEpiParams ep) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * H_STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int up = lane >> 5, r16 = lane & 15, kc = (lane >> 4) & 1;

  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int ct = idx % tiles_n;
  const int rt = (idx / tiles_n) * 8 + xcd;
  if (rt >= tiles_m) return;
  const int m_base = rt * H_TBM, n_base = ct * H_TBN;
  const int nk = Kd / H_BK;
  // de-phase the two co-resident workgroups of a CU: the second slot of the first generation starts late, so that
  // one workgroup's epilogue traffic meets the other's main loop (later generations inherit the offset)
  if (ep.stagger > 0 && b >= ep.stagger_lo && b < ep.stagger_hi) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ep.stagger) __builtin_amdgcn_s_sleep(32);
  }

  // ---- DMA plan (as wire_gemmx3g.hip): piece q = wave + 4 j
  const char* src[H_PPW];
  int adv[H_PPW], loff[H_PPW];
#pragma unroll
  for (int j = 0; j < H_PPW; ++j) {
    const int q = wave + 4 * j;
    if (q < 16) {
      const int r = 16 * q + (lane >> 2);
      const int c = (lane & 3) ^ h_gray((r >> 2) & 3);
      int row = m_base + r;
      row = row < M ? row : M - 1;
      src[j] = reinterpret_cast<const char*>(A + (size_t)row * lda + c * 4);
      adv[j] = H_BK * 4;
      loff[j] = q * 1024;
    } else {
      const int bq = q - 16;
      const int p = bq >> 2, cb = bq & 3;
      src[j] = reinterpret_cast<const char*>(Bu + ((size_t)ct * nk * 3 + p) * (128 * 16) + cb * (32 * 16) + lane * 8);
      adv[j] = 3 * 128 * 16 * 2;
      loff[j] = H_ABYTES + p * H_BPLANE + cb * 1024;
    }
  }
  auto issue = [&](int kt, int buf) {
    unsigned char* S = smem + buf * H_STAGE;
#pragma unroll
    for (int j = 0; j < H_PPW; ++j) h_dma16(src[j] + (size_t)kt * adv[j], S + loff[j]);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage
  int a_rd[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int rr = wave * 64 + p * 32 + up * 16 + r16;
    const int sw = h_gray((rr >> 2) & 3);
    a_rd[p][0] = rr * 64 + (((2 * kc) ^ sw) << 4);
    a_rd[p][1] = rr * 64 + (((2 * kc + 1) ^ sw) << 4);
  }
  // B: column 16 cb + (lane & 15), half kc; G1 reads plane l (lower lanes) / h (upper lanes), G2 plane m, G3 plane h
  const int b_col = H_ABYTES + r16 * 32 + kc * 16;
  const int b_g1 = b_col + (up ? 0 : 2 * H_BPLANE);
  const int b_g2 = b_col + H_BPLANE;
  const int b_g3 = b_col;

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* S = smem + buf * H_STAGE;
    f32x4 araw[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      araw[p][0] = *reinterpret_cast<const f32x4*>(S + a_rd[p][0]);
      araw[p][1] = *reinterpret_cast<const f32x4*>(S + a_rd[p][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // A operands of the four 16-row blocks: split once per half-wave, planes exchanged between the halves
    hbf16x8 f1[4], f2[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      unsigned H[4], Mi[4], L[4], H2[4];
      h_split2(araw[p][0][0], araw[p][0][1], H[0], Mi[0], L[0]);
      h_split2(araw[p][0][2], araw[p][0][3], H[1], Mi[1], L[1]);
      h_split2(araw[p][1][0], araw[p][1][1], H[2], Mi[2], L[2]);
      h_split2(araw[p][1][2], araw[p][1][3], H[3], Mi[3], L[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        H2[e] = H[e];
        h_swap_halves(H[e], L[e]);       // H = (h0 | l0): F1 of rows 0-15,  L = (h1 | l1): F1 of rows 16-31
        h_swap_halves(Mi[e], H2[e]);     // Mi = (m0 | h0): F2 of rows 0-15, H2 = (m1 | h1): F2 of rows 16-31
      }
      f1[2 * p] = __builtin_bit_cast(hbf16x8, hu32x4{H[0], H[1], H[2], H[3]});
      f1[2 * p + 1] = __builtin_bit_cast(hbf16x8, hu32x4{L[0], L[1], L[2], L[3]});
      f2[2 * p] = __builtin_bit_cast(hbf16x8, hu32x4{Mi[0], Mi[1], Mi[2], Mi[3]});
      f2[2 * p + 1] = __builtin_bit_cast(hbf16x8, hu32x4{H2[0], H2[1], H2[2], H2[3]});
    }
    // the 128 columns in two halves of four 16-column blocks: 12 B fragments (48 registers) live at a time
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      hbf16x8 g1[4], g2[4], g3[4];
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {
        const int cb = 4 * hb + cq;
        g1[cq] = *reinterpret_cast<const hbf16x8*>(S + b_g1 + cb * 512);
        g2[cq] = *reinterpret_cast<const hbf16x8*>(S + b_g2 + cb * 512);
        g3[cq] = *reinterpret_cast<const hbf16x8*>(S + b_g3 + cb * 512);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
          // small terms first: (h l + l h), (m m + h m), (m h + h h)
          // operands swapped (weights first): the block comes out transposed, four consecutive columns per lane
          H_MFMA(g1[cq], f1[rb], acc[rb][4 * hb + cq]);
          H_MFMA(g2[cq], f2[rb], acc[rb][4 * hb + cq]);
          H_MFMA(g3[cq], f2[rb], acc[rb][4 * hb + cq]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
  }
  h_epilogue<EPI>(acc, ep, M, m_base + wave * 64, n_base, Nc, lane, smem, wave, rt);
}

template <int EPI>
static hipError_t launchx3h_t(hipStream_t s, const float* A, int lda, const unsigned short* Bu, int64_t M, int Nc,
                              int Kd, const GemmEpiParams& ep) {
  const int tiles_m = (int)((M + H_TBM - 1) / H_TBM);
  const int tiles_n = (Nc + H_TBN - 1) / H_TBN;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
  hipLaunchKernelGGL((gemmx3h_nt_kernel<EPI>), dim3((unsigned)(tiles_m_pad * tiles_n)), dim3(256), 0, s, A, lda, Bu,
                     (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

static int x3h_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// bit 0: forward / store epilogues, bit 1: data-gradient epilogues.  In bench.py (same box, interleaved runs,
// profiles/r02_bench_h16_ab.txt): forward launches 0.758 -> 0.700 ms, data gradient 0.749 -> 0.739 ms, step 9.93 -> 9.63 ms
// bit 2: siren / gauss / relu epilogues, bit 3: the 2-D Gabor epilogues (sweep A/B, same box: siren 72.4 -> 74.9,
// relu 81.9 -> 85.0, wire2d 44.2 -> 47.6 M samples/s)
static int g_x3_h16 = x3h_env("WIRE_X3_H16", 15);
static int g_x3h_stagger = x3h_env("WIRE_X3H_STAGGER", 0);      // 100 MHz ticks (100 = 1 us)
int gemmx3h_tune_set(const char* key, int value) {
  if (!strcmp(key, "x3_h16") && value >= 0 && value <= 15) { g_x3_h16 = value; return 0; }
  if (!strcmp(key, "x3h_stagger") && value >= 0) { g_x3h_stagger = value; return 0; }
  return -1;
}
int gemmx3h_mode() { return g_x3_h16; }
bool gemmx3h_handles(int epi, int64_t M) {
  if (!g_x3_h16 || M < 4096) return false;
  if (epi == EPI_STORE || epi == EPI_GABOR_FWD) return (g_x3_h16 & 1) != 0;
  if (epi == EPI_GABOR_BWD || epi == EPI_GABOR_BWD_FIRST) return (g_x3_h16 & 2) != 0;
  if (epi >= EPI_SIREN_FWD && epi <= EPI_RELU_BWD) return (g_x3_h16 & 4) != 0;
  if (epi >= EPI_GABOR2D_FWD && epi <= EPI_GABOR2D_BWD_FIRST) return (g_x3_h16 & 8) != 0;
  return false;
}

// Bx3: the split image of launch_x3_split_b; its second half holds the unswizzled planes this kernel copies
hipError_t launch_gemmx3h_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx3, int64_t M, int Nc,
                             int Kd, const GemmEpiParams& ep_in) {
  if (M <= 0) return hipSuccess;
  if ((Nc & 63) || (Kd & 31) || (lda & 3) || M > 0x7fffff00LL) return hipErrorInvalidValue;
  GemmEpiParams ep = ep_in;
  ep.stagger = g_x3h_stagger; ep.stagger_lo = 256; ep.stagger_hi = 512;
  const unsigned short* Bu = (const unsigned short*)Bx3 + (size_t)gemmx3_b_image_floats(Nc, Kd);   // = half the image, in ushorts
  switch (epi) {
    case EPI_STORE: return launchx3h_t<EPI_STORE>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR_FWD: return launchx3h_t<EPI_GABOR_FWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR_BWD: return launchx3h_t<EPI_GABOR_BWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR_BWD_FIRST: return launchx3h_t<EPI_GABOR_BWD_FIRST>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_SIREN_FWD: return launchx3h_t<EPI_SIREN_FWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GAUSS_FWD: return launchx3h_t<EPI_GAUSS_FWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_RELU_FWD: return launchx3h_t<EPI_RELU_FWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_SIREN_BWD: return launchx3h_t<EPI_SIREN_BWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GAUSS_BWD: return launchx3h_t<EPI_GAUSS_BWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_RELU_BWD: return launchx3h_t<EPI_RELU_BWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR2D_FWD:
      if (Nc & 127) return hipErrorInvalidValue;
      return launchx3h_t<EPI_GABOR2D_FWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD: return launchx3h_t<EPI_GABOR2D_BWD>(s, A, lda, Bu, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD_FIRST: return launchx3h_t<EPI_GABOR2D_BWD_FIRST>(s, A, lda, Bu, M, Nc, Kd, ep);
    default: return hipErrorInvalidValue;
  }
}
