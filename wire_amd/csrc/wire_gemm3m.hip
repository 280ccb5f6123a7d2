// wire_gemm3m.hip -- complex GEMMs of the WIRE layers with the 3-multiplication
// (Gauss / "3M") complex product on fp32 MFMA: 6 real flop per complex MAC on the
// matrix cores instead of 8.
//
//   forward / data gradient (NT):   C = A * B^T      (A, C blocked-planar complex rows)
//       T1 = (a_re + a_im) b_re,  T2 = a_re (b_im - b_re),  T3 = a_im (b_re + b_im)
//       C_re = T1 - T3,  C_im = T1 + T2
//   weight gradient (TN):           g_W = g^T conj(z)
//       P1 = g_re^T z_re,  P2 = g_im^T z_im,  P3 = (g_re + g_im)^T (z_re - z_im)
//       g_W.re = P1 + P2,  g_W.im = P3 - P1 + P2
//
// VALU budget.  On gfx950 v_mfma_f32_32x32x2_f32 and fp32 VALU instructions do not
// overlap (tools/mfma_valu_probe.hip: ~5 matrix-pipe cycles per v_fma, ~9 per
// v_exp/v_sin, for any number of waves per SIMD), so these kernels are bound by
// 64 * #MFMA + 5 * #VALU and every vector instruction is treated as a cost:
//   * the three operand combinations of the 3M product are formed in registers from
//     the LDS fragments (one VALU op per MFMA).  Precomputing the weight-side ones
//     into a third weight plane was measured SLOWER (+6 %): the extra LDS plane
//     costs a workgroup of occupancy (3 instead of 4 per CU);
//   * the epilogue uses the hardware transcendental unit (wire_dev.h
//     gabor_fwd_lean), uniform (SGPR) base pointers with one 32-bit byte offset
//     per accumulator row, and a branch-free fast path for full tiles.
//
// Tiling: 64 rows x 64 complex features per 256-thread workgroup, 4 waves of
// 32 x 32, three 32x32 accumulators (T1,T2,T3) per wave; reduction slabs of 16
// complex indices, double-buffered LDS (36 KB -> 4 workgroups per CU), rows
// padded to 36 floats (conflict-free ds_read_b128).  Replaces the
// ATen complex addmm / mm of modules/wire.py:89 and of its autograd backward.
#include <cstdlib>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"

#define T3M 64          // rows and complex features per workgroup tile
#define KC 16           // complex reduction indices per slab
#define LSTA 36         // LDS row stride of A: 16 re + 16 im + 4 pad
#define LSTB 36         // LDS row stride of B: 16 re + 16 im + 4 pad
#define BUF3M (T3M * LSTA + T3M * LSTB)

template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm3m_nt_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int M, int Kp_out,
    int Kp_in, int tiles_m, int tiles_n, GemmEpiParams ep) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF3M];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int ct = idx % tiles_n;
  const int rt = (idx / tiles_n) * 8 + xcd;
  if (rt >= tiles_m) return;
  const int m_base = rt * T3M;
  const int f_base = ct * T3M;                 // first output feature of the tile

  // loader: thread -> (row 0..63, 16-byte chunk 0..3): A re, A im, B re, B im
  const int lrow = tid >> 2;
  const int lc4 = (tid & 3) * 4;
  int ra = m_base + lrow; ra = ra < M ? ra : M - 1;
  int rb = f_base + lrow; rb = rb < Kp_out ? rb : Kp_out - 1;
  const float* a_src = A + (size_t)ra * lda + lc4;
  const float* b_src = B + (size_t)rb * ldb + lc4;
  const int sta = lrow * LSTA + lc4;
  const int stb = T3M * LSTA + lrow * LSTB + lc4;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = Kp_in / KC;                   // slab s covers reduction indices [16 s, 16 s + 16)
  // column of the slab's re segment in a blocked-planar row (the im segment is 32 floats on)
  auto cola = [](int s) { return ((s >> 1) << 6) + ((s & 1) << 4); };

  f32x4 g_ar, g_ai, g_br, g_bi;
  {
    g_ar = *reinterpret_cast<const f32x4*>(a_src);
    g_ai = *reinterpret_cast<const f32x4*>(a_src + 32);
    g_br = *reinterpret_cast<const f32x4*>(b_src);
    g_bi = *reinterpret_cast<const f32x4*>(b_src + 32);
    *reinterpret_cast<f32x4*>(&smem[sta]) = g_ar;
    *reinterpret_cast<f32x4*>(&smem[sta + 16]) = g_ai;
    *reinterpret_cast<f32x4*>(&smem[stb]) = g_br;
    *reinterpret_cast<f32x4*>(&smem[stb + 16]) = g_bi;
  }
  __syncthreads();

  const int a_rd = (wave_m * 32 + l31) * LSTA + 4 * h;
  const int b_rd = T3M * LSTA + (wave_n * 32 + l31) * LSTB + 4 * h;

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < nk;
#ifdef WIRE_ABLATE
    if (more && !(ep.ablate & 1)) {
#else
    if (more) {
#endif
      const int ca = cola(kt + 1);
      g_ar = *reinterpret_cast<const f32x4*>(a_src + ca);
      g_ai = *reinterpret_cast<const f32x4*>(a_src + ca + 32);
      g_br = *reinterpret_cast<const f32x4*>(b_src + ca);
      g_bi = *reinterpret_cast<const f32x4*>(b_src + ca + 32);
    }
    const float* S = smem + buf * BUF3M;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 ar = *reinterpret_cast<const f32x4*>(&S[a_rd + 8 * q]);
      const f32x4 ai = *reinterpret_cast<const f32x4*>(&S[a_rd + 16 + 8 * q]);
      const f32x4 br = *reinterpret_cast<const f32x4*>(&S[b_rd + 8 * q]);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(&S[b_rd + 16 + 8 * q]);
      const f32x4 as = ar + ai;
      const f32x4 bd = bi - br;
      const f32x4 bs = br + bi;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[s], br[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s], bd[s], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[s], bs[s], acc[2], 0, 0, 0);
      }
    }
#ifdef WIRE_ABLATE
    if (more && !(ep.ablate & 2)) {
#else
    if (more) {
#endif
      float* Wd = smem + (buf ^ 1) * BUF3M;
      *reinterpret_cast<f32x4*>(&Wd[sta]) = g_ar;
      *reinterpret_cast<f32x4*>(&Wd[sta + 16]) = g_ai;
      *reinterpret_cast<f32x4*>(&Wd[stb]) = g_br;
      *reinterpret_cast<f32x4*>(&Wd[stb + 16]) = g_bi;
    }
#ifdef WIRE_ABLATE
    if (!(ep.ablate & 4))
#endif
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  // lane holds, for feature f = f_w + l31 and 16 rows, T1/T2/T3 -> (re, im).
  // Row of accumulator register r: m_w + 4 h + (r & 3) + 8 (r >> 2).
  const int m_w = m_base + wave_m * 32;
  const int f_w = f_base + wave_n * 32;
  if (f_w >= Kp_out) return;
  const int feat = f_w + l31;
  const int c_re = ((feat >> 5) << 6) + (feat & 31);
  const int c_im = c_re + 32;
  // fast path: every row of the tile exists and every feature of the wave is a real one
  const bool full = (m_base + T3M <= M) && (f_w + 32 <= ep.kvalid) && !ep.wide;

  if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < M) {
        ep.o0[(size_t)row * ep.ld0 + c_re] = acc[0][r] - acc[2][r];
        ep.o0[(size_t)row * ep.ld0 + c_im] = acc[0][r] + acc[1][r];
      }
    }
  } else if constexpr (EPI == EPI_GABOR_FWD) {
    const float b_re = ep.bias[c_re], b_im = ep.bias[c_im];
    if (full) {
      // uniform bases + one 32-bit byte offset per row (ld0 == ld1 is checked by the launcher)
      char* __restrict__ lin_b = reinterpret_cast<char*>(ep.o0);
      char* __restrict__ out_b = reinterpret_cast<char*>(ep.o1);
      const unsigned ldb4 = (unsigned)ep.ld1 * 4u;
      const unsigned off0 = (unsigned)(m_w + 4 * h) * ldb4 + (unsigned)c_re * 4u;
      const float w0 = ep.omega;
      const float w0l2e = ep.omega * 1.44269502f;
      const float ns2l2e = -(ep.scale * ep.scale) * 1.44269502f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * ldb4;
        const float u = (acc[0][r] - acc[2][r]) + b_re;
        const float v = (acc[0][r] + acc[1][r]) + b_im;
        float o_re, o_im;
        gabor_fwd_lean(u, v, w0, w0l2e, ns2l2e, o_re, o_im);
        if (lin_b) {
          *reinterpret_cast<float*>(lin_b + off) = u;
          *reinterpret_cast<float*>(lin_b + off + 128) = v;
        }
        *reinterpret_cast<float*>(out_b + off) = o_re;
        *reinterpret_cast<float*>(out_b + off + 128) = o_im;
      }
    } else {
      const bool valid = feat < ep.kvalid;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float u = (acc[0][r] - acc[2][r]) + b_re;
        const float v = (acc[0][r] + acc[1][r]) + b_im;
        float o_re, o_im;
        gabor_fwd(u, v, ep.omega, ep.scale, o_re, o_im);
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
    if (m_base + T3M <= M && !ep.wide) {
      const char* __restrict__ lin_b = reinterpret_cast<const char*>(ep.i0);
      const char* __restrict__ out_b = reinterpret_cast<const char*>(ep.i1);
      char* __restrict__ gl_b = reinterpret_cast<char*>(ep.o0);
      const unsigned ldb4 = (unsigned)ep.ld0 * 4u;
      const unsigned off0 = (unsigned)(m_w + 4 * h) * ldb4 + (unsigned)c_re * 4u;
      const float w0 = ep.omega;
      // two batches of 8 rows: issue the 32 loads of a batch, then consume them
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
          const float gr = acc[0][r] - acc[2][r], gi = acc[0][r] + acc[1][r];
          const float c_r = __builtin_fmaf(pr[j], gr, pi[j] * gi);
          const float c_i = __builtin_fmaf(pr[j], gi, -(pi[j] * gr));
          const float t = m2s2 * c_r;
          *reinterpret_cast<float*>(gl_b + off) = __builtin_fmaf(t, lu[j], w0 * c_i);
          *reinterpret_cast<float*>(gl_b + off + 128) = __builtin_fmaf(t, lv[j], -(w0 * c_r));
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
          const size_t o0 = (size_t)row * ep.ld0, o1 = (size_t)row * ep.ld1;
          const float u = ep.i0[o0 + c_re], v = ep.i0[o0 + c_im];
          const float pr = ep.i1[o1 + c_re], pi = ep.i1[o1 + c_im];
          float gl_re, gl_im;
          gabor_bwd(acc[0][r] - acc[2][r], acc[0][r] + acc[1][r], u, v, pr, pi, ep.omega, m2s2, gl_re, gl_im);
          ep.o0[o0 + c_re] = gl_re;
          ep.o0[o0 + c_im] = gl_im;
        }
      }
    }
  } else if constexpr (EPI == EPI_GABOR_BWD_FIRST) {
    const float m2s2 = -2.f * ep.scale * ep.scale;
    const bool valid = feat < ep.kvalid;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    float bb = 0.f;
    if (valid) {
      bb = ep.b0[feat];
      for (int d = 0; d < ep.D; ++d) w[d] = ep.W0[feat * ep.D + d];
    }
    if (m_base + T3M <= M && !ep.wide && ep.D <= 3) {
      // lean path: uniform bases + 32-bit byte offsets, loads of 8 rows in flight
      const char* __restrict__ out_b = reinterpret_cast<const char*>(ep.i1);
      const char* __restrict__ crd_b = reinterpret_cast<const char*>(ep.coords);
      char* __restrict__ gu_b = reinterpret_cast<char*>(ep.o0);
      const unsigned ld1b = (unsigned)ep.ld1 * 4u, ldub = (unsigned)ep.ldu * 4u, ldcb = (unsigned)ep.D * 4u;
      const unsigned row0 = (unsigned)(m_w + 4 * h);
      const float w0 = ep.omega;
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
          const float gr = acc[0][r] - acc[2][r], gi = acc[0][r] + acc[1][r];
          const float c_r = __builtin_fmaf(pr[j], gr, pi[j] * gi);
          const float c_i = __builtin_fmaf(pr[j], gi, -(pi[j] * gr));
          const float gu = __builtin_fmaf(m2s2 * c_r, u, w0 * c_i);
          *reinterpret_cast<float*>(gu_b + row * ldub + (unsigned)feat * 4u) = valid ? gu : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_w + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
          float u = bb;
          for (int d = 0; d < ep.D; ++d) u = __builtin_fmaf(ep.coords[(size_t)row * ep.D + d], w[d], u);
          const size_t o1 = (size_t)row * ep.ld1;
          const float pr = ep.i1[o1 + c_re], pi = ep.i1[o1 + c_im];
          const float gu = gabor_bwd_real(acc[0][r] - acc[2][r], acc[0][r] + acc[1][r], u, pr, pi, ep.omega, m2s2);
          ep.o0[(size_t)row * ep.ldu + feat] = valid ? gu : 0.f;
        }
      }
    }
  }
}

template <int EPI>
static hipError_t launch3m_t(hipStream_t s, const float* A, int lda, const float* B, int ldb, int64_t M,
                             int Kp_out, int Kp_in, const GemmEpiParams& ep) {
  const int tiles_m = (int)((M + T3M - 1) / T3M);
  const int tiles_n = (Kp_out + T3M - 1) / T3M;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
#ifdef WIRE_ABLATE
  const size_t dyn = (size_t)((ep.ablate >> 8) & 0xff) * 1024;   // occupancy experiments: pad LDS per WG
#else
  const size_t dyn = 0;
#endif
  hipLaunchKernelGGL((gemm3m_nt_kernel<EPI>), dim3((unsigned)(tiles_m_pad * tiles_n)), dim3(256), dyn, s, A,
                     lda, B, ldb, (int)M, Kp_out, Kp_in, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

int gemm3m_tune_set(const char* key, int value) { (void)key; (void)value; return -1; }

// C[M][Kp_out] = A[M][Kp_in] * B[Kp_out][Kp_in]^T, all blocked-planar complex rows (launch_pack3m
// writes W and conj(W)^T in that layout).
hipError_t launch_gemm3m_nt(hipStream_t s, int epi, const float* A, int lda, const float* B, int ldb,
                            int64_t M, int Kp_out, int Kp_in, const GemmEpiParams& ep) {
  if (M <= 0) return hipSuccess;
  if ((Kp_out & 31) || (Kp_in & 31) || (lda & 3) || (ldb & 3) || M > 0x7fffff00LL) return hipErrorInvalidValue;
  // the fast epilogues share one row offset between their buffers and use 32-bit byte offsets;
  // anything else (different leading dims, buffers of 4 GB and more) takes the 64-bit path
  GemmEpiParams e2 = ep;
  if ((epi == EPI_GABOR_FWD || epi == EPI_GABOR_BWD) && ep.ld0 != ep.ld1) e2.wide = 1;
  if ((double)M * (double)(ep.ld1 > ep.ld0 ? ep.ld1 : ep.ld0) * 4.0 >= 4294967296.0) e2.wide = 1;
  switch (epi) {
    case EPI_STORE: return launch3m_t<EPI_STORE>(s, A, lda, B, ldb, M, Kp_out, Kp_in, e2);
    case EPI_GABOR_FWD: return launch3m_t<EPI_GABOR_FWD>(s, A, lda, B, ldb, M, Kp_out, Kp_in, e2);
    case EPI_GABOR_BWD: return launch3m_t<EPI_GABOR_BWD>(s, A, lda, B, ldb, M, Kp_out, Kp_in, e2);
    case EPI_GABOR_BWD_FIRST: return launch3m_t<EPI_GABOR_BWD_FIRST>(s, A, lda, B, ldb, M, Kp_out, Kp_in, e2);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------
// weight gradient: slab[split][3][Kp_o][Kp_i] (planes P1,P2,P3), bslab[split][2][Kp_o]
// ---------------------------------------------------------------------------
#define TKR 16   // rows per LDS stage

__global__ __launch_bounds__(256, 3) void gemm3m_tn_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, long long n, int Kp_o,
    int Kp_i, int tiles_i, int tiles, int nsplit, long long chunk, float* __restrict__ slab,
    float* __restrict__ bslab) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TKR * 128];   // [buf][G|Z][row][re 64 | im 64]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_o = wave >> 1, wave_i = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int bb = blockIdx.x;
  const int xcd = bb & 7, idx = bb >> 3;
  const int tile = idx % tiles;
  const int split = (idx / tiles) * 8 + xcd;
  if (split >= nsplit) return;
  const int to = tile / tiles_i, ti = tile % tiles_i;
  const int o_base = to * 64, i_base = ti * 64;
  const long long row0 = (long long)split * chunk;
  long long row1 = row0 + chunk;
  if (row1 > n) row1 = n;

  // loader: thread -> (row 0..15, chunk 0..15): G re, G im, Z re, Z im float4 each
  const int lrow = tid >> 4;
  const int lc = (tid & 15) * 4;                       // feature offset within the 64-feature tile
  int of = o_base + lc; if (of > Kp_o - 4) of = Kp_o - 4;      // ragged last tile: stay in the row
  int jf = i_base + lc; if (jf > Kp_i - 4) jf = Kp_i - 4;
  const int gcol = ((of >> 5) << 6) + (of & 31);
  const int zcol = ((jf >> 5) << 6) + (jf & 31);
  const int st_off = lrow * 128 + lc;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bs_r = 0.f, bs_i = 0.f;
  const bool do_bias = (bslab != nullptr) && (ti == 0) && (wave_i == 0);

  const long long nrows = row1 > row0 ? row1 - row0 : 0;
  const int nk = (int)((nrows + TKR - 1) / TKR);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 r_gr, r_gi, r_zr, r_zi;
  auto gload = [&](int kt) {
    const long long row = row0 + (long long)kt * TKR + lrow;
    if (row < row1) {
      const float* gp = G + (size_t)row * ldg + gcol;
      const float* zp = Z + (size_t)row * ldz + zcol;
      r_gr = *reinterpret_cast<const f32x4*>(gp);
      r_gi = *reinterpret_cast<const f32x4*>(gp + 32);
      r_zr = *reinterpret_cast<const f32x4*>(zp);
      r_zi = *reinterpret_cast<const f32x4*>(zp + 32);
    } else {
      r_gr = zero4; r_gi = zero4; r_zr = zero4; r_zi = zero4;
    }
  };
  auto lstore = [&](int buf) {
    float* Gs = smem + buf * (2 * TKR * 128);
    float* Zs = Gs + TKR * 128;
    *reinterpret_cast<f32x4*>(&Gs[st_off]) = r_gr;
    *reinterpret_cast<f32x4*>(&Gs[st_off + 64]) = r_gi;
    *reinterpret_cast<f32x4*>(&Zs[st_off]) = r_zr;
    *reinterpret_cast<f32x4*>(&Zs[st_off + 64]) = r_zi;
  };
  if (nk > 0) { gload(0); lstore(0); }
  __syncthreads();

  const int g_rd = h * 128 + wave_o * 32 + l31;
  const int z_rd = h * 128 + wave_i * 32 + l31;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) gload(kt + 1);
    const float* Gs = smem + buf * (2 * TKR * 128);
    const float* Zs = Gs + TKR * 128;
#pragma unroll
    for (int kp = 0; kp < TKR / 2; ++kp) {
      const float gr = Gs[g_rd + kp * 256], gi = Gs[g_rd + kp * 256 + 64];
      const float zr = Zs[z_rd + kp * 256], zi = Zs[z_rd + kp * 256 + 64];
      if (do_bias) { bs_r += gr; bs_i += gi; }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr, zr, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gi, zi, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(gr + gi, zr - zi, acc[2], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  const int o_w = o_base + wave_o * 32, i_w = i_base + wave_i * 32;
  if (do_bias) {
    const float vr = bs_r + __shfl_xor(bs_r, 32);
    const float vi = bs_i + __shfl_xor(bs_i, 32);
    if (h == 0 && o_w + l31 < Kp_o) {
      bslab[((size_t)split * 2 + 0) * Kp_o + o_w + l31] = vr;
      bslab[((size_t)split * 2 + 1) * Kp_o + o_w + l31] = vi;
    }
  }
  if (o_w >= Kp_o || i_w >= Kp_i) return;
  const size_t plane = (size_t)Kp_o * Kp_i;
  float* out = slab + (size_t)split * 3 * plane;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o_w + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (o < Kp_o) out[t * plane + (size_t)o * Kp_i + i_w + l31] = acc[t][r];
    }
}

int gemm3m_tn_splits(int64_t n, int Kp_o, int Kp_i, int max_splits) {
  const int tiles = ((Kp_o + 63) / 64) * ((Kp_i + 63) / 64);
  int s = (1024 + tiles - 1) / tiles;            // ~4 workgroups per CU
  int64_t by_rows = (n + 255) / 256;
  if (by_rows < 1) by_rows = 1;
  if (s > by_rows) s = (int)by_rows;
  if (s > max_splits) s = max_splits;
  if (s < 1) s = 1;
  return s;
}

hipError_t launch_gemm3m_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                            int Kp_o, int Kp_i, int splits, float* slab, float* bslab) {
  if ((Kp_o & 31) || (Kp_i & 31) || (ldg & 3) || (ldz & 3) || splits < 1) return hipErrorInvalidValue;
  const int tiles_o = (Kp_o + 63) / 64, tiles_i = (Kp_i + 63) / 64;
  long long chunk = (n + splits - 1) / splits;
  chunk = (chunk + TKR - 1) / TKR * TKR;
  if (chunk < TKR) chunk = TKR;
  const int splits_pad = (splits + 7) & ~7;
  hipLaunchKernelGGL(gemm3m_tn_kernel, dim3((unsigned)(tiles_o * tiles_i * splits_pad)), dim3(256), 0, s, G,
                     ldg, Z, ldz, (long long)n, Kp_o, Kp_i, tiles_i, tiles_o * tiles_i, splits, chunk, slab,
                     bslab);
  return hipGetLastError();
}
