// wire_gemm_epi.h -- the fused epilogues of the NT GEMMs, shared by the fp32-MFMA kernel
// (wire_gemm.hip) and the split-bf16 kernel (wire_gemmx3.hip): both leave the same accumulator image.
//
// accumulator element r of tile (i, j): row = m_w + 32 i + (r&3) + 8 (r>>2) + 4 h,
// column = n_w + 32 j + (lane & 31)   (C/D map of v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16).
#pragma once
#include "wire_dev.h"
#include "wire_gemm.h"

// LEAN selects, for the Gabor epilogues of full tiles, the forms of wire_gemm3m.hip: hardware transcendentals
// (gabor_fwd_lean), uniform base pointers with one 32-bit byte offset per accumulator row (the launcher sets
// ep.wide when that could overflow or the leading dimensions differ), loads of 8 rows kept in flight.
template <int EPI, int MT, int WN, bool LEAN = false>
WIRE_DEVINL void gemm_epilogue(f32x16 (&acc)[MT][WN], const GemmEpiParams& ep, const int M, const int m_w,
                               const int n_w, const int l31, const int h) {
  if constexpr (LEAN && (EPI == EPI_GABOR_FWD || EPI == EPI_GABOR_BWD)) {
    if (m_w + MT * 32 <= M && !ep.wide) {
      const unsigned ldb4 = (unsigned)ep.ld0 * 4u;
      bool done = true;
      if constexpr (EPI == EPI_GABOR_FWD) {
        // every feature of the wave's column blocks must be a real one (pad features are written as 0 below)
        done = ((n_w >> 1) + (WN / 2) * 32) <= ep.kvalid;
        if (done) {
          char* __restrict__ lin_b = reinterpret_cast<char*>(ep.o0);
          char* __restrict__ out_b = reinterpret_cast<char*>(ep.o1);
          const float w0 = ep.omega, w0l2e = ep.omega * 1.44269502f;
          const float ns2l2e = -(ep.scale * ep.scale) * 1.44269502f;
#pragma unroll
          for (int jp = 0; jp < WN; jp += 2) {
            const int c_re = n_w + 32 * jp + l31;
            const float b_re = ep.bias[c_re], b_im = ep.bias[c_re + 32];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const unsigned off0 = (unsigned)(m_w + 32 * i + 4 * h) * ldb4 + (unsigned)c_re * 4u;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
                const float u = acc[i][jp][r] + b_re, v = acc[i][jp + 1][r] + b_im;
                float o_re, o_im;
                gabor_fwd_lean(u, v, w0, w0l2e, ns2l2e, o_re, o_im);
                if (lin_b) {
                  *reinterpret_cast<float*>(lin_b + off) = u;
                  *reinterpret_cast<float*>(lin_b + off + 128) = v;
                }
                *reinterpret_cast<float*>(out_b + off) = o_re;
                *reinterpret_cast<float*>(out_b + off + 128) = o_im;
              }
            }
          }
        }
      } else {
        const char* __restrict__ lin_b = reinterpret_cast<const char*>(ep.i0);
        const char* __restrict__ out_b = reinterpret_cast<const char*>(ep.i1);
        char* __restrict__ gl_b = reinterpret_cast<char*>(ep.o0);
        const float m2s2 = -2.f * ep.scale * ep.scale, w0 = ep.omega;
#pragma unroll
        for (int jp = 0; jp < WN; jp += 2) {
          const int c_re = n_w + 32 * jp + l31;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const unsigned off0 = (unsigned)(m_w + 32 * i + 4 * h) * ldb4 + (unsigned)c_re * 4u;
#pragma unroll
            for (int rb0 = 0; rb0 < 16; rb0 += 8) {
              float lu[8], lv[8], pr[8], pi[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int r = rb0 + j;
                const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
                lu[j] = *reinterpret_cast<const float*>(lin_b + off);
                lv[j] = *reinterpret_cast<const float*>(lin_b + off + 128);
                pr[j] = *reinterpret_cast<const float*>(out_b + off);
                pi[j] = *reinterpret_cast<const float*>(out_b + off + 128);
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int r = rb0 + j;
                const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
                const float gr = acc[i][jp][r], gi = acc[i][jp + 1][r];
                const float c_r = __builtin_fmaf(pr[j], gr, pi[j] * gi);
                const float c_i = __builtin_fmaf(pr[j], gi, -(pi[j] * gr));
                const float t = m2s2 * c_r;
                *reinterpret_cast<float*>(gl_b + off) = __builtin_fmaf(t, lu[j], w0 * c_i);
                *reinterpret_cast<float*>(gl_b + off + 128) = __builtin_fmaf(t, lv[j], -(w0 * c_r));
              }
            }
          }
        }
      }
      if (done) return;
    }
  }
  if constexpr (LEAN && EPI == EPI_GABOR_BWD_FIRST) {
    // first layer: u is recomputed from the coordinates (D <= 3), loads of 8 rows in flight
    if (m_w + MT * 32 <= M && !ep.wide && ep.D <= 3) {
      const float m2s2 = -2.f * ep.scale * ep.scale, w0 = ep.omega;
      const char* __restrict__ out_b = reinterpret_cast<const char*>(ep.i1);
      const char* __restrict__ crd_b = reinterpret_cast<const char*>(ep.coords);
      char* __restrict__ gu_b = reinterpret_cast<char*>(ep.o0);
      const unsigned ld1b = (unsigned)ep.ld1 * 4u, ldub = (unsigned)ep.ldu * 4u, ldcb = (unsigned)ep.D * 4u;
#pragma unroll
      for (int jp = 0; jp < WN; jp += 2) {
        const int c_re = n_w + 32 * jp + l31;
        const int feat = ((c_re >> 6) << 5) + l31;
        const bool valid = feat < ep.kvalid;
        float w[3] = {0.f, 0.f, 0.f};
        float bb = 0.f;
        if (valid) {
          bb = ep.b0[feat];
          for (int d = 0; d < ep.D; ++d) w[d] = ep.W0[feat * ep.D + d];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const unsigned row0 = (unsigned)(m_w + 32 * i + 4 * h);
#pragma unroll
          for (int rb0 = 0; rb0 < 16; rb0 += 8) {
            float pr[8], pi[8], x0[8], x1[8], x2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int r = rb0 + j;
              const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
              const unsigned off = row * ld1b + (unsigned)c_re * 4u;
              pr[j] = *reinterpret_cast<const float*>(out_b + off);
              pi[j] = *reinterpret_cast<const float*>(out_b + off + 128);
              x0[j] = *reinterpret_cast<const float*>(crd_b + row * ldcb);
              x1[j] = ep.D > 1 ? *reinterpret_cast<const float*>(crd_b + row * ldcb + 4) : 0.f;
              x2[j] = ep.D > 2 ? *reinterpret_cast<const float*>(crd_b + row * ldcb + 8) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int r = rb0 + j;
              const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
              float u = __builtin_fmaf(x0[j], w[0], bb);
              u = __builtin_fmaf(x1[j], w[1], u);
              u = __builtin_fmaf(x2[j], w[2], u);
              const float gr = acc[i][jp][r], gi = acc[i][jp + 1][r];
              const float c_r = __builtin_fmaf(pr[j], gr, pi[j] * gi);
              const float c_i = __builtin_fmaf(pr[j], gi, -(pi[j] * gr));
              const float gu = __builtin_fmaf(m2s2 * c_r, u, w0 * c_i);
              *reinterpret_cast<float*>(gu_b + row * ldub + (unsigned)feat * 4u) = valid ? gu : 0.f;
            }
          }
        }
      }
      return;
    }
  }
  // ---- lean forms of the real-valued epilogues (siren / gauss / relu) and of the 2-D Gabor: full tiles, uniform
  // base pointers + 32-bit byte offsets (no 64-bit multiply per element, no row test), loads of 8 rows in flight.
  // The short GEMMs of these nets (K = 256 real: 16 stages per tile) spend as long in the epilogue as in the main
  // loop when every element pays a dependent load -> store chain (profiles/r02_siren_kernel_stats.csv).
  if constexpr (LEAN && (EPI == EPI_SIREN_FWD || EPI == EPI_GAUSS_FWD || EPI == EPI_RELU_FWD)) {
    if (m_w + MT * 32 <= M && !ep.wide && n_w + WN * 32 <= ep.kvalid) {
      constexpr int ACT = EPI - EPI_SIREN_FWD;
      char* __restrict__ lin_b = reinterpret_cast<char*>(ep.o0);
      char* __restrict__ out_b = reinterpret_cast<char*>(ep.o1);
      const unsigned ldb4 = (unsigned)ep.ld0 * 4u;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n_w + 32 * j + l31;
        const float bb = ep.bias[col];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const unsigned off0 = (unsigned)(m_w + 32 * i + 4 * h) * ldb4 + (unsigned)col * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
            const float lin = acc[i][j][r] + bb;
            if (lin_b) *reinterpret_cast<float*>(lin_b + off) = lin;
            *reinterpret_cast<float*>(out_b + off) = real_act_fwd_lean<ACT>(lin, ep.omega, ep.scale);
          }
        }
      }
      return;
    }
  }
  if constexpr (LEAN && (EPI == EPI_SIREN_BWD || EPI == EPI_GAUSS_BWD || EPI == EPI_RELU_BWD)) {
    if (m_w + MT * 32 <= M && !ep.wide) {
      constexpr int ACT = EPI - EPI_SIREN_BWD;
      const char* __restrict__ lin_b = reinterpret_cast<const char*>(ep.i0);
      const char* __restrict__ out_b = reinterpret_cast<const char*>(ep.i1);
      char* __restrict__ gl_b = reinterpret_cast<char*>(ep.o0);
      const unsigned ldb4 = (unsigned)ep.ld0 * 4u;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n_w + 32 * j + l31;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const unsigned off0 = (unsigned)(m_w + 32 * i + 4 * h) * ldb4 + (unsigned)col * 4u;
#pragma unroll
          for (int rb0 = 0; rb0 < 16; rb0 += 8) {
            float lv[8], ov[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rb0 + q;
              const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
              // siren needs lin, relu needs out (lin > 0 <=> out > 0; its lin is never stored), gauss both
              lv[q] = (ACT != ACT_RELU) ? *reinterpret_cast<const float*>(lin_b + off) : 0.f;
              ov[q] = (ACT != ACT_SIREN) ? *reinterpret_cast<const float*>(out_b + off) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = rb0 + q;
              const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
              *reinterpret_cast<float*>(gl_b + off) = real_act_bwd_lean<ACT>(acc[i][j][r], lv[q], ov[q], ep.omega, ep.scale);
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (LEAN && EPI == EPI_GABOR2D_FWD) {
    static_assert(EPI != EPI_GABOR2D_FWD || WN == 4, "2-D Gabor needs a 128-column wave tile");
    if (m_w + MT * 32 <= M && !ep.wide && ((n_w >> 7) << 5) + 32 <= ep.kvalid) {
      const int c0 = n_w + l31, grp = n_w >> 7, oc_re = (grp << 6) + l31;
      const float b_u = ep.bias[c0], b_v = ep.bias[c0 + 32], b_p = ep.bias[c0 + 64], b_q = ep.bias[c0 + 96];
      char* __restrict__ L_b = reinterpret_cast<char*>(ep.o0);
      char* __restrict__ O_b = reinterpret_cast<char*>(ep.o1);
      const unsigned ld0b = (unsigned)ep.ld0 * 4u, ld1b = (unsigned)ep.ld1 * 4u;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned row0 = (unsigned)(m_w + 32 * i + 4 * h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
          const float u = acc[i][0][r] + b_u, v = acc[i][1][r] + b_v;
          const float pp = acc[i][2][r] + b_p, qq = acc[i][3][r] + b_q;
          float o_re, o_im;
          gabor2d_fwd_lean(u, v, pp, qq, ep.omega, ep.scale, o_re, o_im);
          if (L_b) {
            float* Lp = reinterpret_cast<float*>(L_b + row * ld0b + (unsigned)c0 * 4u);
            Lp[0] = u; Lp[32] = v; Lp[64] = pp; Lp[96] = qq;
          }
          float* Op = reinterpret_cast<float*>(O_b + row * ld1b + (unsigned)oc_re * 4u);
          Op[0] = o_re; Op[32] = o_im;
        }
      }
      return;
    }
  }
  if constexpr (LEAN && EPI == EPI_GABOR2D_BWD) {
    if (m_w + MT * 32 <= M && !ep.wide) {
      const float m2s2 = -2.f * ep.scale * ep.scale, w0 = ep.omega;
      const char* __restrict__ L_b = reinterpret_cast<const char*>(ep.i0);
      const char* __restrict__ O_b = reinterpret_cast<const char*>(ep.i1);
      char* __restrict__ G_b = reinterpret_cast<char*>(ep.o0);
      const unsigned ld0b = (unsigned)ep.ld0 * 4u, ld1b = (unsigned)ep.ld1 * 4u;
#pragma unroll
      for (int jp = 0; jp < WN; jp += 2) {
        const int c_re = n_w + 32 * jp + l31;
        const int grp = c_re >> 6;
        const unsigned lc4 = (unsigned)((grp << 7) + l31) * 4u, oc4 = (unsigned)c_re * 4u;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const unsigned row0 = (unsigned)(m_w + 32 * i + 4 * h);
#pragma unroll
          for (int rb0 = 0; rb0 < 16; rb0 += 4) {
            float u[4], v[4], pp[4], qq[4], pr[4], pi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = rb0 + q;
              const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
              const float* Lp = reinterpret_cast<const float*>(L_b + row * ld0b + lc4);
              const float* Op = reinterpret_cast<const float*>(O_b + row * ld1b + oc4);
              u[q] = Lp[0]; v[q] = Lp[32]; pp[q] = Lp[64]; qq[q] = Lp[96];
              pr[q] = Op[0]; pi[q] = Op[32];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = rb0 + q;
              const unsigned row = row0 + (unsigned)((r & 3) + 8 * (r >> 2));
              const float gr = acc[i][jp][r], gi = acc[i][jp + 1][r];
              const float c_r = __builtin_fmaf(pr[q], gr, pi[q] * gi);
              const float c_i = __builtin_fmaf(pr[q], gi, -(pi[q] * gr));
              const float t = m2s2 * c_r;
              float* Gp = reinterpret_cast<float*>(G_b + row * ld0b + lc4);
              Gp[0] = __builtin_fmaf(t, u[q], w0 * c_i);
              Gp[32] = __builtin_fmaf(t, v[q], -(w0 * c_r));
              Gp[64] = t * pp[q];
              Gp[96] = t * qq[q];
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = n_w + 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) ep.o0[(size_t)row * ep.ld0 + col] = acc[i][j][r];
        }
      }
  } else if constexpr (EPI == EPI_GABOR_FWD) {
    // tile pair (j, j+1) = (re, im) of 32 complex features
#pragma unroll
    for (int jp = 0; jp < WN; jp += 2) {
      const int c_re = n_w + 32 * jp + l31;
      const int c_im = c_re + 32;
      const int feat = ((c_re >> 6) << 5) + l31;
      const float b_re = ep.bias[c_re];
      const float b_im = ep.bias[c_im];
      const bool valid = feat < ep.kvalid;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float u = acc[i][jp][r] + b_re;
          const float v = acc[i][jp + 1][r] + b_im;
          float o_re, o_im;
          if (LEAN) gabor_fwd_lean(u, v, ep.omega, ep.omega * 1.44269502f, -(ep.scale * ep.scale) * 1.44269502f, o_re, o_im);
          else gabor_fwd(u, v, ep.omega, ep.scale, o_re, o_im);
          if (!valid) { o_re = 0.f; o_im = 0.f; }
          if (row < M) {
            if (ep.o0) {
              ep.o0[(size_t)row * ep.ld0 + c_re] = u;
              ep.o0[(size_t)row * ep.ld0 + c_im] = v;
            }
            ep.o1[(size_t)row * ep.ld1 + c_re] = o_re;
            ep.o1[(size_t)row * ep.ld1 + c_im] = o_im;
          }
        }
    }
  } else if constexpr (EPI == EPI_GABOR_BWD) {
    const float m2s2 = -2.f * ep.scale * ep.scale;
#pragma unroll
    for (int jp = 0; jp < WN; jp += 2) {
      const int c_re = n_w + 32 * jp + l31;
      const int c_im = c_re + 32;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            const size_t o0 = (size_t)row * ep.ld0;
            const size_t o1 = (size_t)row * ep.ld1;
            const float u = ep.i0[o0 + c_re], v = ep.i0[o0 + c_im];
            const float pr = ep.i1[o1 + c_re], pi = ep.i1[o1 + c_im];
            float gl_re, gl_im;
            gabor_bwd(acc[i][jp][r], acc[i][jp + 1][r], u, v, pr, pi, ep.omega, m2s2, gl_re, gl_im);
            ep.o0[o0 + c_re] = gl_re;
            ep.o0[o0 + c_im] = gl_im;
          }
        }
    }
  } else if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
    const float m2s2 = -2.f * ep.scale * ep.scale;
#pragma unroll
    for (int jp = 0; jp < WN; jp += 2) {
      const int c_re = n_w + 32 * jp + l31;
      const int c_im = c_re + 32;
      const int feat = ((c_re >> 6) << 5) + l31;
      const bool valid = feat < ep.kvalid;
      float w[4] = {0.f, 0.f, 0.f, 0.f};
      float bb = 0.f;
      if (valid) {
        bb = ep.b0[feat];
        for (int d = 0; d < ep.D; ++d) w[d] = ep.W0[feat * ep.D + d];
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            float u = bb;
            for (int d = 0; d < ep.D; ++d) u = __builtin_fmaf(ep.coords[(size_t)row * ep.D + d], w[d], u);
            const size_t o1 = (size_t)row * ep.ld1;
            const float pr = ep.i1[o1 + c_re], pi = ep.i1[o1 + c_im];
            float gu = gabor_bwd_real(acc[i][jp][r], acc[i][jp + 1][r], u, pr, pi, ep.omega, m2s2);
            ep.o0[(size_t)row * ep.ldu + feat] = valid ? gu : 0.f;
          }
        }
    }
  } else if constexpr (EPI == EPI_SIREN_FWD || EPI == EPI_GAUSS_FWD || EPI == EPI_RELU_FWD) {
    constexpr int ACT = EPI - EPI_SIREN_FWD;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n_w + 32 * j + l31;
      const float bb = ep.bias[col];
      const bool valid = col < ep.kvalid;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float lin = acc[i][j][r] + bb;
          float o = LEAN ? real_act_fwd_lean<ACT>(lin, ep.omega, ep.scale) : real_act_fwd<ACT>(lin, ep.omega, ep.scale);
          if (!valid) o = 0.f;
          if (row < M) {
            if (ep.o0) ep.o0[(size_t)row * ep.ld0 + col] = lin;
            ep.o1[(size_t)row * ep.ld1 + col] = o;
          }
        }
    }
  } else if constexpr (EPI == EPI_SIREN_BWD || EPI == EPI_GAUSS_BWD || EPI == EPI_RELU_BWD) {
    constexpr int ACT = EPI - EPI_SIREN_BWD;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = n_w + 32 * j + l31;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            const float lin = (ACT != ACT_RELU) ? ep.i0[(size_t)row * ep.ld0 + col] : 0.f;   // relu: lin is not stored
            const float out = ep.i1[(size_t)row * ep.ld1 + col];
            ep.o0[(size_t)row * ep.ld0 + col] =
                LEAN ? real_act_bwd_lean<ACT>(acc[i][j][r], lin, out, ep.omega, ep.scale)
                     : real_act_bwd<ACT>(acc[i][j][r], lin, out, ep.omega, ep.scale);
          }
        }
    }
  } else if constexpr (EPI == EPI_GABOR2D_FWD) {
    // wave tile = 32 rows x 128 columns = (lin_re | lin_im | sy_re | sy_im) of 32 features
    static_assert(EPI != EPI_GABOR2D_FWD || WN == 4, "2-D Gabor needs a 128-column wave tile");
    const int c0 = n_w + l31;                 // GEMM column of lin_re
    const int grp = n_w >> 7;                 // feature group
    const int feat = (grp << 5) + l31;
    const int oc_re = (grp << 6) + l31;       // column in the P-wide output row
    const float b_u = ep.bias[c0], b_v = ep.bias[c0 + 32], b_p = ep.bias[c0 + 64], b_q = ep.bias[c0 + 96];
    const bool valid = feat < ep.kvalid;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float u = acc[i][0][r] + b_u, v = acc[i][1][r] + b_v;
        const float p = acc[i][2][r] + b_p, q = acc[i][3][r] + b_q;
        float o_re, o_im;
        if (LEAN) gabor2d_fwd_lean(u, v, p, q, ep.omega, ep.scale, o_re, o_im);
        else gabor2d_fwd(u, v, p, q, ep.omega, ep.scale, o_re, o_im);
        if (!valid) { o_re = 0.f; o_im = 0.f; }
        if (row < M) {
          if (ep.o0) {
            float* L = ep.o0 + (size_t)row * ep.ld0 + c0;
            L[0] = u; L[32] = v; L[64] = p; L[96] = q;
          }
          ep.o1[(size_t)row * ep.ld1 + oc_re] = o_re;
          ep.o1[(size_t)row * ep.ld1 + oc_re + 32] = o_im;
        }
      }
  } else if constexpr (EPI == EPI_GABOR2D_BWD || EPI == EPI_GABOR2D_BWD_FIRST) {
    // C = g_out (re|im pairs, P-wide); writes g_(lin|sy) into the 2P-wide row
    const float s2 = ep.scale * ep.scale;
    const float m2s2 = -2.f * s2;
#pragma unroll
    for (int jp = 0; jp < WN; jp += 2) {
      const int c_re = n_w + 32 * jp + l31;
      const int grp = c_re >> 6;
      const int feat = (grp << 5) + l31;
      const int lc = (grp << 7) + l31;        // lin_re column in the 2P row
      float w[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
      float bb = 0.f, bv = 0.f;
      const bool valid = feat < ep.kvalid;
      if (EPI == EPI_GABOR2D_BWD_FIRST && valid) {
        bb = ep.b0[feat]; bv = ep.b0b[feat];
        for (int d = 0; d < ep.D; ++d) { w[d] = ep.W0[feat * ep.D + d]; wv[d] = ep.W0b[feat * ep.D + d]; }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_w + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < M) {
            const size_t o1 = (size_t)row * ep.ld1;
            const float pr = ep.i1[o1 + c_re], pi = ep.i1[o1 + c_re + 32];
            const float gr = acc[i][jp][r], gi = acc[i][jp + 1][r];
            const float c_r = __builtin_fmaf(pr, gr, pi * gi);
            const float c_i = __builtin_fmaf(pr, gi, -(pi * gr));
            const float t = m2s2 * c_r;
            if (EPI == EPI_GABOR2D_BWD) {
              const float* L = ep.i0 + (size_t)row * ep.ld0 + lc;
              const float u = L[0], v = L[32], p = L[64], q = L[96];
              float* Gp = ep.o0 + (size_t)row * ep.ld0 + lc;
              // g_lin = -2 s^2 P lin - j w0 c ; g_sy = -2 s^2 P sy
              Gp[0] = __builtin_fmaf(t, u, ep.omega * c_i);
              Gp[32] = __builtin_fmaf(t, v, -(ep.omega * c_r));
              Gp[64] = t * p;
              Gp[96] = t * q;
            } else {
              float u = bb, p = bv;
              for (int d = 0; d < ep.D; ++d) {
                const float x = ep.coords[(size_t)row * ep.D + d];
                u = __builtin_fmaf(x, w[d], u);
                p = __builtin_fmaf(x, wv[d], p);
              }
              // real first layer: g_u = -2 s^2 u P + w0 Im c ; g_p = -2 s^2 p P
              float* Gp = ep.o0 + (size_t)row * (2 * ep.ldu);
              Gp[feat] = valid ? __builtin_fmaf(t, u, ep.omega * c_i) : 0.f;
              Gp[ep.ldu + feat] = valid ? t * p : 0.f;
            }
          }
        }
    }
  }
}
