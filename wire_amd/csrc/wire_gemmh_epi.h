// wire_gemmh_epi.h -- fused epilogues of the 16 x 16 x 32 NT GEMMs (wire_gemmx3h.hip: 3 x bf16 split, wire_gemmx2h.hip:
// 2 x fp16 split).  Same arithmetic as wire_gemm_epi.h (lean forms of wire_dev.h); what differs is the accumulator
// layout these kernels leave behind and the 16-byte-vector, whole-cache-line global accesses built on it.
// Reference arithmetic: modules/wire.py:88-93, wire2d.py:56-67, siren.py:48-49, gauss.py:27-28, relu.py:28-29 and the
// autograd backward of each.
#pragma once
#include "wire_dev.h"
#include "wire_gemm.h"

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

// X2 (wire_gemmx2h.hip, the 2 x fp16 split GEMM): the accumulators carry the operands' power-of-two scales -- they are
// multiplied by acc_scale = 1 / (s_A s_B) first (exact) -- and the epilogue tracks max |value| of the tensor it writes
// that the NEXT GEMM reads as an operand (out for the forward forms, g_lin for the data-gradient forms) into
// ep.amax_out, from which that GEMM derives ITS operand scale.
WIRE_DEVINL void h_amax4(float& m, const f32x4& v) {
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                         __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
}

// NRB = 16-row blocks of the wave's tile (4: 64 rows per wave, 256-row workgroup tiles; 2: 32 rows per wave, 128-row tiles
// -- not with the first-layer sums (cr_partial), whose per-tile layout is that of the 256-row tile)
template <int EPI, bool X2 = false, int NRB = 4>
WIRE_DEVINL void h_epilogue(f32x4 (&acc)[NRB][8], const GemmEpiParams& ep, const int M, const int m_w, const int n_w,
                            const int Nc, const int lane, unsigned char* lds, const int wave, const int rt,
                            const float acc_scale = 1.f) {
  const int rr = lane & 7, ch = (lane >> 3) & 1, g = lane >> 4;
  float amx = 0.f;
  if constexpr (X2) {
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] *= acc_scale;
  }
  const int cq = 16 * ch + 4 * g;                                 // this lane's first column inside a 32-column span
  if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {                              // 32-column spans of the 128-column wave tile
      const int col = n_w + 32 * sp + cq;
      if (n_w + 32 * sp >= Nc) continue;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
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
      for (int rb = 0; rb < NRB; ++rb) {
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
            if constexpr (X2) h_amax4(amx, o);
            if (row < M) wire_store_out4(ep.o1 + (size_t)row * ep.ld1 + col, o, X2 ? ep.o1_split : 0.f);
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
      for (int rp = 0; rp < NRB; rp += 2) {
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
          if constexpr (X2) h_amax4(amx, gl);
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
    for (int rb = 0; rb < NRB; ++rb) {
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
          if constexpr (X2) { h_amax4(amx, o_re); h_amax4(amx, o_im); }
          if (row < M) {
            float* Op = ep.o1 + (size_t)row * ep.ld1 + oc;
            wire_store_out4(Op, o_re, X2 ? ep.o1_split : 0.f);
            wire_store_out4(Op + 32, o_im, X2 ? ep.o1_split : 0.f);
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
      for (int rb = 0; rb < NRB; ++rb) {
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
          if constexpr (X2 && EPI == EPI_GABOR2D_BWD) { h_amax4(amx, g0); h_amax4(amx, g1); h_amax4(amx, g2); h_amax4(amx, g3); }
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
      for (int rb = 0; rb < NRB; ++rb) {
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
              if constexpr (X2) { h_amax4(amx, o_re); h_amax4(amx, o_im); }
              if (row < M) {
                float* Op = ep.o1 + (size_t)row * ep.ld1 + c0;
                wire_store_out4(Op, o_re, X2 ? ep.o1_split : 0.f);
                wire_store_out4(Op + 32, o_im, X2 ? ep.o1_split : 0.f);
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
            if constexpr (X2) { h_amax4(amx, gl_re); h_amax4(amx, gl_im); }
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
  if constexpr (X2) {
    if (ep.amax_out) wire_amax_publish(ep.amax_out, amx, lane);
  }
}
