// wire_gemm.hip -- fp32-MFMA GEMM kernels for gfx950 with fused WIRE epilogues.
//
// Two kernels carry >99 % of the flops of the WIRE hot path:
//
//  gemm_nt_kernel : C[M][Nc] = A[M][Kd] * Bt[Nc][Kd]^T   (both operands
//      K-contiguous).  Used for the layer forward (A = activations, Bt = real-
//      expanded weights) and for the data gradient (A = g_lin, Bt = the
//      transposed image).  The epilogue applies bias + the Gabor / SIREN /
//      Gauss / ReLU nonlinearity (forward) or the closed-form activation
//      gradient (backward) on the accumulators, so lin/out/g_lin are each
//      written exactly once and never re-read by an elementwise kernel.
//      Replaces ATen addmm + ~8 elementwise ops per layer
//      (reference modules/wire.py:88-93).
//
//  gemm_tn_kernel : slab[s] = G[rows_s]^T * Z[rows_s]    (weight gradient,
//      reduction over the sample axis split across workgroups; partial slabs
//      are summed deterministically by reduce kernels in wire_point.hip).
//
// Tiling (both): 128x128 output tile per 256-thread workgroup, 4 waves, each
// wave a 64x64 (or 32x128) block of v_mfma_f32_32x32x2_f32 tiles, BK = 16 (40 KB
// of double-buffered LDS -> 4 workgroups per CU) or 32.
//
// These are the 4-multiplication (real-expanded) kernels: they serve the real
// nets (siren / gauss / relu), wire2d, the per-layer API and complex_3m=0.  The
// default path of `wire` layers is the 3-multiplication form in wire_gemm3m.hip.
#include <cstdlib>
#include <atomic>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"
#include "wire_gemm_epi.h"

#define BM 128
#define BN 128
// BKT (K-slab per LDS stage) is a template parameter: 32 -> 72 KB LDS, 2 workgroups/CU;
// 16 -> 40 KB LDS, 3 workgroups/CU.  LDS rows are padded by 4 floats (strides 36 / 20):
// both make the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots.

template <int EPI, int MT, int WN, int BKT>
__global__ __launch_bounds__(256, (BKT == 32 ? 2 : 3)) void gemm_nt_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Bt, int ldb, int M, int Nc,
    int Kd, int tiles_m, int tiles_n, GemmEpiParams ep) {
  constexpr int LDS_STRIDE = BKT + 4;
  constexpr int BK = BKT;
  constexpr int TPR = BKT / 4;          // loader threads per tile row
  constexpr int RPP = 256 / TPR;        // tile rows per loader pass
  constexpr int NP = BM / RPP;          // loader passes per operand
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BM * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr int WAVES_N = BN / (WN * 32);
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int l31 = lane & 31;
  const int h = lane >> 5;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch),
  // so the tiles_n column tiles of one row tile run back-to-back on one L2.
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int idx = b >> 3;
  const int ct = idx % tiles_n;
  const int rt = (idx / tiles_n) * 8 + xcd;
  if (rt >= tiles_m) return;
  const int m_base = rt * BM;
  const int n_base = ct * BN;

  // ---- global -> register staging map: TPR threads x float4 per BK-float row
  const int lrow = tid / TPR;
  const int lc4 = (tid % TPR) * 4;
  const float* a_src[NP];
  const float* b_src[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    int ra = m_base + lrow + RPP * p;
    ra = ra < M ? ra : M - 1;
    int rb = n_base + lrow + RPP * p;
    rb = rb < Nc ? rb : Nc - 1;
    a_src[p] = A + (size_t)ra * lda + lc4;
    b_src[p] = Bt + (size_t)rb * ldb + lc4;
  }
  const int st_off = lrow * LDS_STRIDE + lc4;

  f32x16 acc[MT][WN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool wave_live = (n_base + wave_n * (WN * 32)) < Nc;
  const int nk = Kd / BK;

#ifdef WIRE_ABLATE
  if (ep.ablate & 8) {
    // distinct static priority per wave slot of the SIMD -> co-resident workgroups drift apart
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11));  // HW_ID[3:0] wave_id
    switch (hwid & 3) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
#endif

  f32x4 ra4[NP], rb4[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    ra4[p] = *reinterpret_cast<const f32x4*>(a_src[p]);
    rb4[p] = *reinterpret_cast<const f32x4*>(b_src[p]);
  }
  {
    float* As = smem;
    float* Bs = smem + BM * LDS_STRIDE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<f32x4*>(&As[st_off + RPP * p * LDS_STRIDE]) = ra4[p];
      *reinterpret_cast<f32x4*>(&Bs[st_off + RPP * p * LDS_STRIDE]) = rb4[p];
    }
  }
  __syncthreads();

  const int a_rd = (wave_m * (MT * 32) + l31) * LDS_STRIDE + 4 * h;
  const int b_rd = (wave_n * (WN * 32) + l31) * LDS_STRIDE + 4 * h;

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < nk;
#ifdef WIRE_ABLATE
    if (more && !(ep.ablate & 1)) {
#else
    if (more) {
#endif
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        ra4[p] = *reinterpret_cast<const f32x4*>(a_src[p] + (kt + 1) * BK);
        rb4[p] = *reinterpret_cast<const f32x4*>(b_src[p] + (kt + 1) * BK);
      }
    }
    const float* As = smem + buf * (2 * BM * LDS_STRIDE);
    const float* Bs = As + BM * LDS_STRIDE;
    if (wave_live) {
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        f32x4 af[MT], bf[WN];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&As[a_rd + i * 32 * LDS_STRIDE + 8 * q]);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          bf[j] = *reinterpret_cast<const f32x4*>(&Bs[b_rd + j * 32 * LDS_STRIDE + 8 * q]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    }
#ifdef WIRE_ABLATE
    if (more && !(ep.ablate & 2)) {
#else
    if (more) {
#endif
      float* Aw = smem + (buf ^ 1) * (2 * BM * LDS_STRIDE);
      float* Bw = Aw + BM * LDS_STRIDE;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        *reinterpret_cast<f32x4*>(&Aw[st_off + RPP * p * LDS_STRIDE]) = ra4[p];
        *reinterpret_cast<f32x4*>(&Bw[st_off + RPP * p * LDS_STRIDE]) = rb4[p];
      }
    }
#ifdef WIRE_ABLATE
    if (!(ep.ablate & 4))
#endif
    __syncthreads();
  }
  if (!wave_live) return;

  // epilogue (wire_gemm_epi.h)
  gemm_epilogue<EPI, MT, WN>(acc, ep, M, m_base + wave_m * (MT * 32), n_base + wave_n * (WN * 32), l31, h);
}

// tuning knobs (wire_tune_set; defaults may be overridden by WIRE_NT_BK in the environment)
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static std::atomic<int> g_nt_bk{env_int("WIRE_NT_BK", 16)};   // K-slab depth of the NT kernel: 16 or 32
static int nt_bk() { return g_nt_bk; }
int gemm_tune_set(const char* key, int value) {
  if (!strcmp(key, "nt_bk") && (value == 16 || value == 32)) { g_nt_bk = value; return 0; }
  return -1;
}

template <int EPI, int MT, int WN>
static hipError_t launch_nt_t(hipStream_t s, const float* A, int lda, const float* Bt, int ldb,
                              int64_t M, int Nc, int Kd, const GemmEpiParams& ep) {
  const int tiles_m = (int)((M + BM - 1) / BM);
  const int tiles_n = (Nc + BN - 1) / BN;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
  dim3 grid((unsigned)(tiles_m_pad * tiles_n));
  if (nt_bk() == 16)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, MT, WN, 16>), grid, dim3(256), 0, s, A, lda, Bt, ldb,
                       (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, MT, WN, 32>), grid, dim3(256), 0, s, A, lda, Bt, ldb,
                       (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

hipError_t launch_gemm_nt(hipStream_t s, int epi, const float* A, int lda, const float* Bt, int ldb,
                          int64_t M, int Nc, int Kd, const GemmEpiParams& ep) {
  if (M <= 0) return hipSuccess;
  if ((Nc & 63) || (Kd & 31) || (lda & 3) || (ldb & 3) || M > 0x7fffff00LL) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE: return launch_nt_t<EPI_STORE, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR_FWD: return launch_nt_t<EPI_GABOR_FWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR_BWD: return launch_nt_t<EPI_GABOR_BWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR_BWD_FIRST: return launch_nt_t<EPI_GABOR_BWD_FIRST, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_SIREN_FWD: return launch_nt_t<EPI_SIREN_FWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GAUSS_FWD: return launch_nt_t<EPI_GAUSS_FWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_RELU_FWD: return launch_nt_t<EPI_RELU_FWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_SIREN_BWD: return launch_nt_t<EPI_SIREN_BWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GAUSS_BWD: return launch_nt_t<EPI_GAUSS_BWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_RELU_BWD: return launch_nt_t<EPI_RELU_BWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR2D_FWD:
      if (Nc & 127) return hipErrorInvalidValue;
      return launch_nt_t<EPI_GABOR2D_FWD, 1, 4>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD: return launch_nt_t<EPI_GABOR2D_BWD, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD_FIRST: return launch_nt_t<EPI_GABOR2D_BWD_FIRST, 2, 2>(s, A, lda, Bt, ldb, M, Nc, Kd, ep);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------
// TN GEMM: weight gradient.  slab[split][Pm][Pn] = sum over the split's rows of
// G[row][m] * Z[row][n];  bslab[split][m] = sum of G[row][m]  (bias gradient).
// ---------------------------------------------------------------------------
#define TK 32   // rows (reduction steps) per LDS tile

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, long long n,
    int Pm, int Pn, int tiles_n, int nsplit, long long chunk, float* __restrict__ slab,
    float* __restrict__ bslab, int tiles) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TK * 128];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  // XCD-aware order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), and the
  // `tiles` output tiles of one row split all read the same G/Z rows -> keep them on one L2.
  const int bb = blockIdx.x;
  const int xcd = bb & 7, idx = bb >> 3;
  const int tile = idx % tiles;
  const int split = (idx / tiles) * 8 + xcd;
  if (split >= nsplit) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m_base = tm * 128, n_base = tn * 128;
  const long long row0 = (long long)split * chunk;
  long long row1 = row0 + chunk;
  if (row1 > n) row1 = n;

  // loader: 32 threads x float4 per 128-float row, 8 rows per pass, 4 passes
  const int lrow = tid >> 5;
  const int lc4 = (tid & 31) * 4;
  int gcol = m_base + lc4; if (gcol > Pm - 4) gcol = Pm - 4;
  int zcol = n_base + lc4; if (zcol > Pn - 4) zcol = Pn - 4;
  const int st_off = lrow * 128 + lc4;

  const bool live_m = (m_base + wave_m * 64) < Pm;
  const bool live_n = (n_base + wave_n * 64) < Pn;
  const bool wave_live = live_m && live_n;
  const bool do_bias = (bslab != nullptr) && (tn == 0) && (wave_n == 0) && live_m;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  const long long nrows = row1 > row0 ? row1 - row0 : 0;
  const int nk = (int)((nrows + TK - 1) / TK);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 rg[4], rz[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long row = row0 + (long long)kt * TK + lrow + 8 * p;
      if (row < row1) {
        rg[p] = *reinterpret_cast<const f32x4*>(G + (size_t)row * ldg + gcol);
        rz[p] = *reinterpret_cast<const f32x4*>(Z + (size_t)row * ldz + zcol);
      } else {
        rg[p] = zero4;
        rz[p] = zero4;
      }
    }
  };
  auto lstore = [&](int buf) {
    float* Gs = smem + buf * (2 * TK * 128);
    float* Zs = Gs + TK * 128;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *reinterpret_cast<f32x4*>(&Gs[st_off + 8 * p * 128]) = rg[p];
      *reinterpret_cast<f32x4*>(&Zs[st_off + 8 * p * 128]) = rz[p];
    }
  };

  if (nk > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();

  const int g_rd = h * 128 + wave_m * 64 + l31;
  const int z_rd = h * 128 + wave_n * 64 + l31;

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) gload(kt + 1);
    const float* Gs = smem + buf * (2 * TK * 128);
    const float* Zs = Gs + TK * 128;
    if (wave_live || do_bias) {
#pragma unroll
      for (int kp = 0; kp < TK / 2; ++kp) {
        const float a0 = Gs[g_rd + kp * 256];
        const float a1 = Gs[g_rd + kp * 256 + 32];
        if (do_bias) { bsum[0] += a0; bsum[1] += a1; }
        if (wave_live) {
          const float b0 = Zs[z_rd + kp * 256];
          const float b1 = Zs[z_rd + kp * 256 + 32];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = bsum[i] + __shfl_xor(bsum[i], 32);
      const int m = m_base + wave_m * 64 + 32 * i + l31;
      if (h == 0 && m < Pm) bslab[(size_t)split * Pm + m] = v;
    }
  }
  if (!wave_live) return;
  float* out = slab + (size_t)split * Pm * Pn;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_base + wave_n * 64 + 32 * j + l31;
      if (col < Pn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m_base + wave_m * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Pm) out[(size_t)m * Pn + col] = acc[i][j][r];
        }
      }
    }
}

int gemm_tn_splits(int64_t n, int Pm, int Pn, int max_splits) {
  const int tiles = ((Pm + 127) / 128) * ((Pn + 127) / 128);
  int s = (512 + tiles - 1) / tiles;          // ~2 workgroups per CU
  int64_t by_rows = (n + 255) / 256;          // at least 256 rows per split
  if (by_rows < 1) by_rows = 1;
  if (s > by_rows) s = (int)by_rows;
  if (s > max_splits) s = max_splits;
  if (s < 1) s = 1;
  return s;
}

hipError_t launch_gemm_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz,
                          int64_t n, int Pm, int Pn, int splits, float* slab, float* bslab) {
  if ((Pm & 3) || (Pn & 3) || (ldg & 3) || (ldz & 3) || splits < 1) return hipErrorInvalidValue;
  const int tiles_m = (Pm + 127) / 128, tiles_n = (Pn + 127) / 128;
  long long chunk = (n + splits - 1) / splits;
  chunk = (chunk + TK - 1) / TK * TK;
  if (chunk < TK) chunk = TK;
  const int splits_pad = (splits + 7) & ~7;
  dim3 grid((unsigned)(tiles_m * tiles_n * splits_pad), 1);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, s, G, ldg, Z, ldz, (long long)n, Pm, Pn,
                     tiles_n, splits, chunk, slab, bslab, tiles_m * tiles_n);
  return hipGetLastError();
}
