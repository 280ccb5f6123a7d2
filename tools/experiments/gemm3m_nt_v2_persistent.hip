// wire_gemm3m_v2.hip -- persistent, software-pipelined 3M complex NT GEMM.
//
// Same arithmetic as gemm3m_nt_kernel (wire_gemm3m.hip) -- C = A * B^T with the
// 3-multiplication complex product on v_mfma_f32_32x32x2_f32 -- but organised so
// that the MFMA stream of a wave never waits on memory, LDS or an epilogue:
//
//  * persistent workgroups: 2 per CU walk the 64x64 output tiles (XCD-aware
//    order); the slab stream continues across tile boundaries, so the next
//    tile's first slab is already in flight while the current tile finishes
//    (no per-tile prologue, no workgroup relaunch);
//  * slabs of 32 complex features (one full 256-B (re|im) segment per row),
//    48 MFMAs per wave and ONE barrier per slab, placed after the third of the
//    four 12-MFMA groups: fragments of group g+1 (and of the next slab's group
//    0) are read from LDS while group g runs, the next slab is written to the
//    other LDS buffer before group 2, and its global loads were issued a full
//    slab earlier -- so the wave passes the barrier with its next operands
//    already in registers;
//  * the epilogue of tile t (bias + Gabor / Gabor-gradient + stores) is spread
//    over the slabs of tile t+1: accumulators are double-buffered and each slab
//    retires R = ceil(16/nk) of the 16 accumulator rows of the previous tile,
//    interleaved with that slab's MFMAs (VALU and matrix pipes run side by
//    side).
//
// 96 accumulator + ~90 other VGPRs -> 2 waves per SIMD; LDS 2 x 34 KB per WG.
#include <cstdlib>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"

#define V2T 64          // rows and complex features per tile
#define V2KC 32         // complex reduction indices per slab
#define V2LST 68        // LDS row stride in floats: 32 re + 32 im + 4 pad (conflict-free ds_read_b128)
#define V2BUF (2 * V2T * V2LST)   // floats per LDS buffer (A rows then B rows)

struct V2Tile { int m_base, f_base; };

template <int EPI, int R>
__global__ __launch_bounds__(256, 2) void gemm3m_nt_v2_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int M, int Kp_out,
    int Kp_in, int tiles_m, int tiles_n, int total_vb, GemmEpiParams ep) {
  __shared__ __attribute__((aligned(16))) float smem[2 * V2BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int nk = Kp_in / V2KC;
  const int G = gridDim.x;

  // virtual block -> tile (XCD-aware: vb and vb+8 share an XCD; the tiles_n column tiles of a
  // row tile are adjacent in that order).  G is a multiple of 8, so vb & 7 is constant per WG.
  auto decode = [&](int vb, V2Tile& t) -> bool {
    if (vb >= total_vb) return false;
    const int xcd = vb & 7, idx = vb >> 3;
    const int ct = idx % tiles_n;
    const int rt = (idx / tiles_n) * 8 + xcd;
    t.m_base = rt * V2T;
    t.f_base = ct * V2T;
    return rt < tiles_m;
  };
  // this WG's tiles: vb = blockIdx.x + i * G, skipping the padded (rt >= tiles_m) ones
  int ntiles = 0;
  for (int vb = blockIdx.x; vb < total_vb; vb += G) {
    V2Tile t;
    if (decode(vb, t)) ++ntiles;
  }
  if (ntiles == 0) return;
  const int total_slabs = ntiles * nk;

  // ---- loader state: thread -> (row 0..63, 16-byte chunk 0..3), 4 chunks of A and 4 of B per slab
  const int lrow = tid >> 2;
  const int lc = (tid & 3) * 4;
  const int st_off = lrow * V2LST + lc;
  int load_vb = blockIdx.x;
  int load_slab = 0;            // slab within the load cursor's tile
  const float* a_row = nullptr;
  const float* b_row = nullptr;
  auto load_enter_tile = [&]() {
    V2Tile t;
    while (!decode(load_vb, t)) load_vb += G;     // only padded tiles are skipped; ntiles guarantees termination
    int ra = t.m_base + lrow; ra = ra < M ? ra : M - 1;
    int rb = t.f_base + lrow; rb = rb < Kp_out ? rb : Kp_out - 1;
    a_row = A + (size_t)ra * lda + lc;
    b_row = B + (size_t)rb * ldb + lc;
  };
  f32x4 sa[4], sb[4];           // staging registers: one slab (A and B) in flight
  auto issue_loads = [&]() {
    const int c = load_slab * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sa[j] = *reinterpret_cast<const f32x4*>(a_row + c + 16 * j);
      sb[j] = *reinterpret_cast<const f32x4*>(b_row + c + 16 * j);
    }
    if (++load_slab == nk) { load_slab = 0; load_vb += G; }
  };
  auto write_lds = [&](int buf) {
    float* As = smem + buf * V2BUF;
    float* Bs = As + V2T * V2LST;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(&As[st_off + 16 * j]) = sa[j];
      *reinterpret_cast<f32x4*>(&Bs[st_off + 16 * j]) = sb[j];
    }
  };

  // ---- fragment reads: group g of a slab = k indices 8g + 4h + {0..3}
  const int a_rd = (wave_m * 32 + l31) * V2LST + 4 * h;
  const int b_rd = (wave_n * 32 + l31) * V2LST + 4 * h;
  struct Frag { f32x4 ar, ai, br, bi; };
  auto read_frag = [&](int buf, int g, Frag& f) {
    const float* As = smem + buf * V2BUF;
    const float* Bs = As + V2T * V2LST;
    f.ar = *reinterpret_cast<const f32x4*>(&As[a_rd + 8 * g]);
    f.ai = *reinterpret_cast<const f32x4*>(&As[a_rd + 32 + 8 * g]);
    f.br = *reinterpret_cast<const f32x4*>(&Bs[b_rd + 8 * g]);
    f.bi = *reinterpret_cast<const f32x4*>(&Bs[b_rd + 32 + 8 * g]);
  };

  f32x16 acc[3], accp[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; accp[t][r] = 0.f; }
  auto mfma_group = [&](const Frag& f) {
    const f32x4 as = f.ar + f.ai;
    const f32x4 bd = f.bi - f.br;
    const f32x4 bs = f.br + f.bi;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], f.br[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.ar[s], bd[s], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.ai[s], bs[s], acc[2], 0, 0, 0);
    }
  };

  // ---- epilogue of the PREVIOUS tile, row by row ------------------------------------------
  int p_m = 0, p_f = 0;          // previous tile's origin
  bool has_prev = false;
  const float m2s2 = -2.f * ep.scale * ep.scale;
  float pre[4 * R] = {};         // GABOR_BWD: (u, v, out_re, out_im) of the rows being retired
  auto epi_cols = [&](int& feat, int& c_re) {
    feat = p_f + wave_n * 32 + l31;
    c_re = ((feat >> 5) << 6) + (feat & 31);
  };
  // stage 1 (early in the slab): issue the loads the rows of group `grp` need.  All register-array
  // indices below are compile-time constants (macro-expanded J), or the arrays would land in scratch.
  auto epi_load_row = [&](int grp, int j, float& p0, float& p1, float& p2, float& p3) {
    int feat, c_re;
    epi_cols(feat, c_re);
    const int r = grp * R + j;
    const int row = p_m + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const int rr = row < M ? row : M - 1;
    if constexpr (EPI == EPI_GABOR_BWD) {
      p0 = ep.i0[(size_t)rr * ep.ld0 + c_re];
      p1 = ep.i0[(size_t)rr * ep.ld0 + c_re + 32];
    } else {
      p0 = ep.coords[(size_t)rr * ep.D];
      p1 = ep.D > 1 ? ep.coords[(size_t)rr * ep.D + 1] : 0.f;
    }
    p2 = ep.i1[(size_t)rr * ep.ld1 + c_re];
    p3 = ep.i1[(size_t)rr * ep.ld1 + c_re + 32];
  };
#define V2_LOAD(J)                                                                              \
  if constexpr ((J) < R) {                                                                      \
    if (grp * R + (J) < 16) epi_load_row(grp, J, pre[4 * (J)], pre[4 * (J) + 1], pre[4 * (J) + 2], pre[4 * (J) + 3]); \
  }
  auto epi_load = [&](int grp) {
    if constexpr (EPI == EPI_GABOR_BWD || EPI == EPI_GABOR_BWD_FIRST) {
      if (p_f + wave_n * 32 >= Kp_out) return;
      V2_LOAD(0) V2_LOAD(1) V2_LOAD(2) V2_LOAD(3) V2_LOAD(4) V2_LOAD(5) V2_LOAD(6) V2_LOAD(7)
      V2_LOAD(8) V2_LOAD(9) V2_LOAD(10) V2_LOAD(11) V2_LOAD(12) V2_LOAD(13) V2_LOAD(14) V2_LOAD(15)
    }
  };
#undef V2_LOAD
  // stage 2 (late in the slab): finish and store rows grp*R .. grp*R+R-1 (static register indices
  // come from the fully unrolled switch in epi_dispatch)
  float w0c[4] = {0.f, 0.f, 0.f, 0.f};   // BWD_FIRST: first-layer weights of this lane's feature
  float b0c = 0.f;
  auto epi_refresh_feature = [&]() {
    if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
      int feat, c_re;
      epi_cols(feat, c_re);
      const bool valid = feat < ep.kvalid;
      b0c = valid ? ep.b0[feat] : 0.f;
      for (int d = 0; d < 4; ++d) w0c[d] = (valid && d < ep.D) ? ep.W0[feat * ep.D + d] : 0.f;
    }
  };
  auto epi_row = [&](int r, float p0, float p1, float p2, float p3, float t1, float t2, float t3) {
    int feat, c_re;
    epi_cols(feat, c_re);
    if (p_f + wave_n * 32 >= Kp_out) return;
    const int c_im = c_re + 32;
    const int row = p_m + wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const float cre = t1 - t3, cim = t1 + t2;
    if constexpr (EPI == EPI_STORE) {
      if (row < M) {
        ep.o0[(size_t)row * ep.ld0 + c_re] = cre;
        ep.o0[(size_t)row * ep.ld0 + c_im] = cim;
      }
    } else if constexpr (EPI == EPI_GABOR_FWD) {
      const float u = cre + ep.bias[c_re], v = cim + ep.bias[c_im];
      float o_re, o_im;
      gabor_fwd(u, v, ep.omega, ep.scale, o_re, o_im);
      if (feat >= ep.kvalid) { o_re = 0.f; o_im = 0.f; }
      if (row < M) {
        if (ep.o0) {
          ep.o0[(size_t)row * ep.ld0 + c_re] = u;
          ep.o0[(size_t)row * ep.ld0 + c_im] = v;
        }
        ep.o1[(size_t)row * ep.ld1 + c_re] = o_re;
        ep.o1[(size_t)row * ep.ld1 + c_im] = o_im;
      }
    } else if constexpr (EPI == EPI_GABOR_BWD) {
      float gl_re, gl_im;
      gabor_bwd(cre, cim, p0, p1, p2, p3, ep.omega, m2s2, gl_re, gl_im);
      if (row < M) {
        ep.o0[(size_t)row * ep.ld0 + c_re] = gl_re;
        ep.o0[(size_t)row * ep.ld0 + c_im] = gl_im;
      }
    } else if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
      float u = b0c;
      u = __builtin_fmaf(p0, w0c[0], u);
      u = __builtin_fmaf(p1, w0c[1], u);
      if (ep.D > 2) {
        const int rr = row < M ? row : M - 1;
        for (int d = 2; d < ep.D; ++d) u = __builtin_fmaf(ep.coords[(size_t)rr * ep.D + d], w0c[d], u);
      }
      const float gu = gabor_bwd_real(cre, cim, u, p2, p3, ep.omega, m2s2);
      if (row < M) ep.o0[(size_t)row * ep.ldu + feat] = feat < ep.kvalid ? gu : 0.f;
    }
  };
#define V2_ROW(c, J)                                                                             \
  if constexpr ((J) < R && (c) * R + (J) < 16) {                                                 \
    constexpr int rr_ = ((c) * R + (J)) & 15;                                                    \
    epi_row(rr_, pre[(4 * (J)) % (4 * R)], pre[(4 * (J) + 1) % (4 * R)], pre[(4 * (J) + 2) % (4 * R)],  \
            pre[(4 * (J) + 3) % (4 * R)], accp[0][rr_], accp[1][rr_], accp[2][rr_]);             \
  }
#define V2_EPI_CASE(c)                                                                           \
  case c: {                                                                                      \
    V2_ROW(c, 0) V2_ROW(c, 1) V2_ROW(c, 2) V2_ROW(c, 3) V2_ROW(c, 4) V2_ROW(c, 5) V2_ROW(c, 6) V2_ROW(c, 7)      \
    V2_ROW(c, 8) V2_ROW(c, 9) V2_ROW(c, 10) V2_ROW(c, 11) V2_ROW(c, 12) V2_ROW(c, 13) V2_ROW(c, 14) V2_ROW(c, 15) \
  } break;
  auto epi_store = [&](int grp) {
    switch (grp) {
      V2_EPI_CASE(0) V2_EPI_CASE(1) V2_EPI_CASE(2) V2_EPI_CASE(3) V2_EPI_CASE(4) V2_EPI_CASE(5)
      V2_EPI_CASE(6) V2_EPI_CASE(7) V2_EPI_CASE(8) V2_EPI_CASE(9) V2_EPI_CASE(10) V2_EPI_CASE(11)
      V2_EPI_CASE(12) V2_EPI_CASE(13) V2_EPI_CASE(14) V2_EPI_CASE(15)
      default: break;
    }
  };
  constexpr int NGRP = (16 + R - 1) / R;     // row groups per tile

  // ---- prologue: slab 0 -> LDS buffer 0, slab 1 -> staging registers
  load_enter_tile();
  issue_loads();
  write_lds(0);
  __syncthreads();
  if (total_slabs > 1) {
    if (load_slab == 0) load_enter_tile();
    issue_loads();
  }
  Frag f0, f1;
  read_frag(0, 0, f0);

  int comp_vb = blockIdx.x;
  V2Tile cur;
  while (!decode(comp_vb, cur)) comp_vb += G;
  int buf = 0;
  int c = 0;                                  // flat slab counter
  for (int ti = 0; ti < ntiles; ++ti) {
    for (int s = 0; s < nk; ++s, ++c) {
      const bool next1 = (c + 1) < total_slabs;
      const bool next2 = (c + 2) < total_slabs;
      const bool do_epi = has_prev && s < NGRP;
      // The schedule below is pinned with sched_barrier(0): hipcc otherwise sinks the fragment
      // reads behind the MFMAs they were meant to overlap and waits for them right away.
      // group 0
      read_frag(buf, 1, f1);
      if (do_epi) epi_load(s);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(f0);
      __builtin_amdgcn_sched_barrier(0);
      // group 1
      read_frag(buf, 2, f0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(f1);
      __builtin_amdgcn_sched_barrier(0);
      // group 2: next slab -> other buffer, then the slab's only barrier
      if (next1) write_lds(buf ^ 1);
      read_frag(buf, 3, f1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(f0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // group 3: next slab's first fragments, next-next slab's global loads, epilogue rows
      if (next1) read_frag(buf ^ 1, 0, f0);
      if (next2) {
        if (load_slab == 0) load_enter_tile();
        issue_loads();
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(f1);
      if (do_epi) epi_store(s);
      __builtin_amdgcn_sched_barrier(0);
      buf ^= 1;
    }
    // previous tile's leftover row groups (only when nk < NGRP)
    if (has_prev)
      for (int g = nk; g < NGRP; ++g) { epi_load(g); epi_store(g); }
    // rotate accumulators: the finished tile becomes "previous"
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      accp[t] = acc[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    p_m = cur.m_base; p_f = cur.f_base; has_prev = true;
    epi_refresh_feature();
    if (ti + 1 < ntiles) {
      comp_vb += G;
      while (!decode(comp_vb, cur)) comp_vb += G;
    }
  }
  // last tile: nothing left to overlap with
  for (int g = 0; g < NGRP; ++g) { epi_load(g); epi_store(g); }
#undef V2_EPI_CASE
#undef V2_ROW
}

template <int EPI, int R>
static hipError_t launch_v2_r(hipStream_t s, const float* A, int lda, const float* B, int ldb, int64_t M,
                              int Kp_out, int Kp_in, const GemmEpiParams& ep) {
  const int tiles_m = (int)((M + V2T - 1) / V2T);
  const int tiles_n = (Kp_out + V2T - 1) / V2T;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
  const int total_vb = tiles_m_pad * tiles_n;
  int grid = 512;                                  // 2 persistent workgroups per CU
  if (grid > total_vb) grid = (total_vb + 7) & ~7;
  hipLaunchKernelGGL((gemm3m_nt_v2_kernel<EPI, R>), dim3((unsigned)grid), dim3(256), 0, s, A, lda, B, ldb,
                     (int)M, Kp_out, Kp_in, tiles_m, tiles_n, total_vb, ep);
  return hipGetLastError();
}

template <int EPI>
static hipError_t launch_v2_t(hipStream_t s, const float* A, int lda, const float* B, int ldb, int64_t M,
                              int Kp_out, int Kp_in, const GemmEpiParams& ep) {
  const int nk = Kp_in / V2KC;
  const int R = (16 + nk - 1) / nk;                // accumulator rows retired per slab
  if (R <= 1) return launch_v2_r<EPI, 1>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
  if (R == 2) return launch_v2_r<EPI, 2>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
  if (R == 3) return launch_v2_r<EPI, 3>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
  if (R == 4) return launch_v2_r<EPI, 4>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
  if (R <= 8) return launch_v2_r<EPI, 8>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
  return launch_v2_r<EPI, 16>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
}

hipError_t launch_gemm3m_nt_v2(hipStream_t s, int epi, const float* A, int lda, const float* B, int ldb,
                               int64_t M, int Kp_out, int Kp_in, const GemmEpiParams& ep) {
  if (M <= 0) return hipSuccess;
  if ((Kp_out & 31) || (Kp_in & 31) || (lda & 3) || (ldb & 3) || M > 0x7fffff00LL) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE: return launch_v2_t<EPI_STORE>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
    case EPI_GABOR_FWD: return launch_v2_t<EPI_GABOR_FWD>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
    case EPI_GABOR_BWD: return launch_v2_t<EPI_GABOR_BWD>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
    case EPI_GABOR_BWD_FIRST: return launch_v2_t<EPI_GABOR_BWD_FIRST>(s, A, lda, B, ldb, M, Kp_out, Kp_in, ep);
    default: return hipErrorInvalidValue;
  }
}
