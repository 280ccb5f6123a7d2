// wire_gemmx3.hip -- the GEMMs of the WIRE hot path on the bf16 matrix cores at fp32 accuracy.
//
// Every fp32 operand x is split exactly into three bf16 terms, x = h + m + l (round-to-nearest at each
// level; 3 x 8 significand bits cover the 24 of fp32), and a product is accumulated from the six partial
// products whose weight is >= 2^-16 of the leading one:
//     a b ~= a_h b_l + a_l b_h + a_m b_m + a_h b_m + a_m b_h + a_h b_h          (dropped: <= 2^-26 |a b|)
// Each partial product of two bf16 numbers is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in
// fp32, so the result carries fewer roundings than the k-ordered fmaf chain of v_mfma_f32_32x32x2_f32:
// measured against fp64 (tools/bf16x3_numerics.hip, K = 512): rms error 0.8x, max 0.76x of the fp32 MFMA.
// Cost: 6 bf16 MFMAs at 16x the fp32-MFMA rate = 2.67x the throughput, and -- unlike the fp32 MFMA, which
// shares the SIMD's vector ALU (tools/mfma_valu_probe.hip) -- the bf16 matrix pipe runs beside the VALU, so
// the splitting and the fused activation epilogues overlap with it.
//
// Memory formats do not change: activations stay fp32 blocked-planar rows in HBM (wire_dev.h); they are
// split by the loader on their way into LDS (11 VALU ops per 2 elements, once per workgroup).  Weights are
// split once per optimizer step into a stage-major bf16 image (x3_split_b_kernel).
//
//  gemmx3_nt_kernel : C[M][Nc] = A[M][Kd] * Bt[Nc][Kd]^T + fused epilogue (wire_gemm_epi.h), forward and
//      data gradient.  128 x 128 tile, 4 waves of 64 x 64 (or 32 x 128), stages of 16 reduction indices,
//      double-buffered LDS of 24 KB per stage (3 planes x (A 128 x 32 B + B 128 x 32 B)) -> 3 workgroups/CU.
//      Operand fragments are 16-byte rows of one plane: ds_read_b128, conflict-free through an XOR of the
//      16-byte half with bit 3 of the row.
//  gemmx3_tn_kernel : slab[s] = G[rows_s]^T * Z[rows_s] (weight gradient; reduction over samples).  The MFMA
//      wants 8 consecutive samples per lane while rows are feature-contiguous: the loader stores row-major
//      bf16 planes and the fragments are read with the gfx950 transposing LDS read ds_read_b64_tr_b16.
//
// Replaces the ATen complex addmm / mm of modules/wire.py:89 and of its autograd backward (and the real
// addmm of modules/siren.py:49, gauss.py:28, relu.py:29, wire2d.py:57-58).
#include <cstdlib>
#include <atomic>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"
#include "wire_gemm_epi.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define X3_BN 128
#define X3_BK 16
#ifndef X3_TN_PF
#define X3_TN_PF 1                   // the same for the weight-gradient kernel (register budget)
#endif
#ifndef X3_LOAD_POS
#define X3_LOAD_POS 2                 // where a stage issues its global loads: 0 top, 1 after the LDS reads, 2 end
                                     // (2: 5 % faster on 128-row tiles than 0 in a same-box A/B, equal on 256-row tiles)
#endif
#ifndef X3_PF
#define X3_PF 2                      // stages of global prefetch ahead of the MFMAs (1 or 2)
#endif
#define X3_PLANE 4096                 // bytes of one operand plane of a stage: 128 rows x 16 bf16

// ---- exact three-way split of two floats into packed bf16 pairs (low half = first element)
WIRE_DEVINL unsigned cvt_pk_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// low half of a packed pair as a float.  Written as an opaque shift: from (H << 16) the compiler would
// re-derive a second, single-element v_cvt_pk_bf16_f32 (one more VALU op per pair and level).
WIRE_DEVINL float bf16_lo(unsigned packed) {
  unsigned r;
  asm("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(packed));
  return __uint_as_float(r);
}
struct Split2 { unsigned h, m, l; };
WIRE_DEVINL Split2 split2(float x0, float x1) {
  Split2 s;
  s.h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - bf16_lo(s.h), r1 = x1 - __uint_as_float(s.h & 0xffff0000u);
  s.m = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - bf16_lo(s.m), q1 = r1 - __uint_as_float(s.m & 0xffff0000u);
  // q has at most 8 significant bits left: the truncating pack is exact
  s.l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
  return s;
}

// byte offset of (row, 16-byte half) inside a 128-row x 32-byte operand plane
WIRE_DEVINL int x3_off(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

#ifdef WIRE_PROBE16
// probe only (results wrong; tools/build_glds_tune.sh): the same flops as two 16x16x32 MFMAs on quarters of the accumulator
typedef float x3f32x4 __attribute__((ext_vector_type(4)));
#define X3_MFMA(a, b, c)                                                                                  \
  do {                                                                                                    \
    x3f32x4 q0_ = {c[0], c[1], c[2], c[3]}, q1_ = {c[4], c[5], c[6], c[7]};                               \
    q0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0_, 0, 0, 0);                                    \
    q1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q1_, 0, 0, 0);                                    \
    c[0] = q0_[0]; c[1] = q0_[1]; c[2] = q0_[2]; c[3] = q0_[3];                                           \
    c[4] = q1_[0]; c[5] = q1_[1]; c[6] = q1_[2]; c[7] = q1_[3];                                           \
  } while (0)
#else
#define X3_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// ---------------------------------------------------------------------------
// weights: fp32 image Bt[Nc][ldb] -> stage-major split image
//   Bx3[col tile ct][stage kt][plane p][col c (128)][16 k]   (bf16; the two 8-k halves of a column swapped when
//   bit 3 of c is set: the LDS bank swizzle)
// so that the 4 KB of one (tile, stage, plane) are contiguous: thread t of the GEMM loads bytes [16 t, 16 t + 16).
// ---------------------------------------------------------------------------
WIRE_DEVINL void x3_split_b_body(const float* __restrict__ Bt, int ldb, int Nc, int Kd, int nk,
                                 unsigned short* __restrict__ Bx3) {
  const int k2 = (blockIdx.x * blockDim.x + threadIdx.x) * 2;   // pair of reduction indices
  const int j = blockIdx.y;                                     // column, < tiles_n * 128
  if (k2 >= Kd) return;
  float x0 = 0.f, x1 = 0.f;
  if (j < Nc) { x0 = Bt[(size_t)j * ldb + k2]; x1 = Bt[(size_t)j * ldb + k2 + 1]; }
  const Split2 sp = split2(x0, x1);
  const unsigned H = sp.h, Mi = sp.m, L = sp.l;
  // the bank swizzle of the LDS image (x3_off: 16-byte half XOR bit 3 of the column) is baked in here, so that a
  // plane of a stage is copied to LDS as it lies -- by 16-byte register stores at byte 16 t (gemmx3_nt_kernel) or
  // by LDS-DMA (wire_gemmx3g.hip)
  const int ct = j >> 7, c = j & 127, kt = k2 >> 4, kk = k2 & 15;
  const int kk_sw = (((kk >> 3) ^ ((c >> 3) & 1)) << 3) | (kk & 7);
  const size_t base = ((size_t)(ct * nk + kt) * 3 * 128 + c) * 16 + kk_sw;
  *reinterpret_cast<unsigned*>(Bx3 + base) = H;
  *reinterpret_cast<unsigned*>(Bx3 + base + 128 * 16) = Mi;
  *reinterpret_cast<unsigned*>(Bx3 + base + 2 * 128 * 16) = L;
  // second image, same order WITHOUT the swizzle: the 16 x 16 x 32 fragment reads of wire_gemmx3h.hip (16 columns x
  // one 8-k half per 16-lane group) are conflict-free on the plain [column][16 k] rows
  unsigned short* Bu = Bx3 + (size_t)gridDim.y * nk * 3 * 16;
  const size_t ubase = ((size_t)(ct * nk + kt) * 3 * 128 + c) * 16 + kk;
  *reinterpret_cast<unsigned*>(Bu + ubase) = H;
  *reinterpret_cast<unsigned*>(Bu + ubase + 128 * 16) = Mi;
  *reinterpret_cast<unsigned*>(Bu + ubase + 2 * 128 * 16) = L;
}

__global__ void x3_split_b_kernel(const float* __restrict__ Bt, int ldb, int Nc, int Kd, int nk,
                                  unsigned short* __restrict__ Bx3) {
  x3_split_b_body(Bt, ldb, Nc, Kd, nk, Bx3);
}
// several images of one shape per launch (blockIdx.z): the per-step repacking of all layers in one go
__global__ void x3_split_b_batch_kernel(X3SplitBatch sb, int ldb, int Nc, int Kd, int nk) {
  x3_split_b_body(sb.src[blockIdx.z], ldb, Nc, Kd, nk, (unsigned short*)sb.dst[blockIdx.z]);
}

int64_t gemmx3_b_image_floats(int Nc, int Kd) {
  const int64_t tiles_n = (Nc + X3_BN - 1) / X3_BN;
  // 3 bf16 planes = 1.5 floats per element; two images (bank-swizzled for the 32 x 32 x 16 kernels, plain for the
  // 16 x 16 x 32 kernel): the second starts gemmx3_b_image_floats() USHORTS into the buffer
  return tiles_n * 128 * (int64_t)Kd * 3;
}

hipError_t launch_x3_split_b(hipStream_t s, const float* Bt, int ldb, int Nc, int Kd, void* Bx3) {
  if ((Kd & 15) || (ldb & 1)) return hipErrorInvalidValue;
  const int tiles_n = (Nc + X3_BN - 1) / X3_BN;
  dim3 grid((unsigned)((Kd / 2 + 127) / 128), (unsigned)(tiles_n * 128));
  hipLaunchKernelGGL(x3_split_b_kernel, grid, dim3(128), 0, s, Bt, ldb, Nc, Kd, Kd / X3_BK,
                     (unsigned short*)Bx3);
  return hipGetLastError();
}

hipError_t launch_x3_split_b_batch(hipStream_t s, const X3SplitBatch& sb, int nb, int ldb, int Nc, int Kd) {
  if ((Kd & 15) || (ldb & 1) || nb < 1 || nb > X3_SPLIT_MAXB) return hipErrorInvalidValue;
  const int tiles_n = (Nc + X3_BN - 1) / X3_BN;
  dim3 grid((unsigned)((Kd / 2 + 127) / 128), (unsigned)(tiles_n * 128), (unsigned)nb);
  hipLaunchKernelGGL(x3_split_b_batch_kernel, grid, dim3(128), 0, s, sb, ldb, Nc, Kd, Kd / X3_BK);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// NT GEMM
// ---------------------------------------------------------------------------
template <int EPI, int MT, int WN>
__global__ __launch_bounds__(256, (MT * WN > 4 ? 2 : 3)) void gemmx3_nt_kernel(
    const float* __restrict__ A, int lda, const unsigned short* __restrict__ Bx3, int M, int Nc, int Kd,
    int tiles_m, int tiles_n, GemmEpiParams ep) {
  constexpr int WAVES_N = X3_BN / (WN * 32);
  constexpr int TBM = (4 / WAVES_N) * MT * 32;     // tile rows: 128 (MT = 2 or 1 x 4) or 256 (MT = 4)
  constexpr int APLANE = TBM * 32;                 // bytes of one A plane of a stage
  constexpr int STAGE = 3 * APLANE + 3 * X3_PLANE;
  constexpr int NPA = TBM / 128;                   // loader passes over the A rows
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the column tiles of
  // one row tile run back-to-back on one L2.
  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int ct = idx % tiles_n;
  const int rt = (idx / tiles_n) * 8 + xcd;
  if (rt >= tiles_m) return;
  const int m_base = rt * TBM, n_base = ct * X3_BN;

  // loader: thread -> (row t >> 1 (+128 per pass), half t & 1): 8 floats of A, one 16-byte piece of each B plane
  const int lrow = tid >> 1, lhalf = tid & 1;
  const float* a_src[NPA];
#pragma unroll
  for (int p = 0; p < NPA; ++p) {
    int ra = m_base + lrow + 128 * p;
    ra = ra < M ? ra : M - 1;
    a_src[p] = A + (size_t)ra * lda + lhalf * 8;
  }
  const int nk = Kd / X3_BK;
  const unsigned short* b_src = Bx3 + (size_t)ct * nk * (3 * 128 * 16) + tid * 8;
  const int st_off = x3_off(lrow, lhalf);

  f32x16 acc[MT][WN];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool wave_live = (n_base + wave_n * (WN * 32)) < Nc;

  // global -> register staging, PF stages ahead of the MFMAs (register sets alternate)
  constexpr int PF = X3_PF;
  struct Staged { f32x4 a0[NPA], a1[NPA]; u32x4 b[3]; };
  Staged st[PF];
  auto gload = [&](Staged& R, int kt) {
#ifdef WIRE_ABLATE
    if (!(ep.ablate & 128))
#endif
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      R.a0[p] = *reinterpret_cast<const f32x4*>(a_src[p] + kt * X3_BK);
      R.a1[p] = *reinterpret_cast<const f32x4*>(a_src[p] + kt * X3_BK + 4);
    }
#ifdef WIRE_ABLATE
    if (!(ep.ablate & 256))
#endif
#pragma unroll
    for (int p = 0; p < 3; ++p)
      R.b[p] = *reinterpret_cast<const u32x4*>(b_src + (size_t)kt * (3 * 128 * 16) + p * (128 * 16));
  };
  auto lstore = [&](const Staged& R, int buf) {
    unsigned char* S = smem + buf * STAGE;
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
      const Split2 s0 = split2(R.a0[p][0], R.a0[p][1]), s1 = split2(R.a0[p][2], R.a0[p][3]);
      const Split2 s2 = split2(R.a1[p][0], R.a1[p][1]), s3 = split2(R.a1[p][2], R.a1[p][3]);
      unsigned char* d = S + st_off + p * 4096;      // 128 rows on: x3_off keeps its XOR (row bit 3)
      *reinterpret_cast<u32x4*>(d) = u32x4{s0.h, s1.h, s2.h, s3.h};
      *reinterpret_cast<u32x4*>(d + APLANE) = u32x4{s0.m, s1.m, s2.m, s3.m};
      *reinterpret_cast<u32x4*>(d + 2 * APLANE) = u32x4{s0.l, s1.l, s2.l, s3.l};
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(S + 3 * APLANE + p * X3_PLANE + tid * 16) = R.b[p];
  };
#ifdef WIRE_ABLATE
  // diagnostic build only (tools/gemm_tune): stamp the shader clock and the 100 MHz reference around the main loop
  unsigned long long t0c = 0, t0r = 0;
  if (ep.ablate & 64) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
#endif
  gload(st[0], 0);
  lstore(st[0], 0);
  if (PF == 2) gload(st[PF - 1], 1);               // nk is even (Kd % 32 == 0)
  __syncthreads();

  int a_rd[MT], b_rd[WN];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_rd[i] = x3_off(wave_m * (MT * 32) + i * 32 + l31, h);
#pragma unroll
  for (int j = 0; j < WN; ++j) b_rd[j] = 3 * APLANE + x3_off(wave_n * (WN * 32) + j * 32 + l31, h);

  // one stage: (optionally) start the global loads of stage kt + PF, MFMAs of stage kt from LDS buffer kt & 1,
  // (optionally) split + store stage kt + 1 into the other buffer.  Branch-free inside: the main loop always
  // loads and stores, the tail never loads (clean s_waitcnt vmcnt(N) placement: the loads of the NEXT-next
  // stage stay in flight across the barrier).  Waves of a ragged last column tile compute on zero padding.
  auto stage = [&](const int kt, const int u, const bool do_load, const bool do_store) {
    const int buf = kt & 1;
    // the set that held stage kt was stored to LDS one iteration ago: it receives stage kt + PF
#ifdef WIRE_ABLATE
    const bool ld = do_load && !(ep.ablate & 1);
#else
    const bool ld = do_load;
#endif
    if (X3_LOAD_POS == 0) {
      if (ld) gload(st[u], kt + PF);
      __builtin_amdgcn_sched_barrier(0);     // all loads of the stage are issued before its MFMAs
    }
    if (X3_LOAD_POS == 3 && ld) gload(st[u], kt + PF);   // placed by the group barriers below: spread over the MFMAs
    {
      const unsigned char* S = smem + buf * STAGE;
      bf16x8 af[MT][3], bf[WN][3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(S + p * APLANE + a_rd[i]);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(S + p * X3_PLANE + b_rd[j]);
      if (X3_LOAD_POS == 1) {
        __builtin_amdgcn_sched_barrier(0);   // fragment reads first, then the global loads, then the MFMAs
        if (ld) gload(st[u], kt + PF);
        __builtin_amdgcn_sched_barrier(0);
      }
      // small terms first: h*l, l*h, m*m, h*m, m*h, h*h
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          X3_MFMA(af[i][0], bf[j][2], acc[i][j]);
          X3_MFMA(af[i][2], bf[j][0], acc[i][j]);
          X3_MFMA(af[i][1], bf[j][1], acc[i][j]);
          X3_MFMA(af[i][0], bf[j][1], acc[i][j]);
          X3_MFMA(af[i][1], bf[j][0], acc[i][j]);
          X3_MFMA(af[i][0], bf[j][0], acc[i][j]);
        }
    }
#ifdef WIRE_ABLATE
    if (do_store && !(ep.ablate & 2)) lstore(st[(u + 1) % PF], buf ^ 1);
#else
    if (do_store) lstore(st[(u + 1) % PF], buf ^ 1);
#endif
    // issue order: the split of the next stage (VALU) and its LDS stores ride in the shadow of this stage's
    // MFMAs -- 2 vector ops after every MFMA, one ds_write after every fourth (MI355X guide: <= 5 single-issue
    // instructions hide per v_mfma_f32_32x32x16_bf16)
    if (do_store) {
#pragma unroll
      for (int g = 0; g < MT * WN * 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (NPA * 44 + MT * WN * 6 - 1) / (MT * WN * 6) + 1, 0);
        if ((g & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, (NPA * 3 + 3 + 5) / 6, 0);
        if (X3_LOAD_POS == 3 && do_load && (g % 4) == 1 && g / 4 < 2 * NPA + 3)
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    if (X3_LOAD_POS == 2) {
      __builtin_amdgcn_sched_barrier(0);     // loads last: behind the MFMAs and the split of the stage
      if (ld) gload(st[u], kt + PF);
    }
#ifdef WIRE_ABLATE
    if (!(ep.ablate & 4))
#endif
    __syncthreads();
  };
  int kt0 = 0;
  for (; kt0 + PF < nk; kt0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) stage(kt0 + u, u, true, true);
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) stage(kt0 + u, u, false, u + 1 < PF);
#ifdef WIRE_ABLATE
  if ((ep.ablate & 64) && tid == 0) {
    unsigned long long* st_out = reinterpret_cast<unsigned long long*>(const_cast<float*>(ep.i0)) + (size_t)blockIdx.x * 2;
    st_out[0] = __builtin_amdgcn_s_memtime() - t0c;
    st_out[1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
#endif
  if (!wave_live) return;
  gemm_epilogue<EPI, MT, WN, true>(acc, ep, M, m_base + wave_m * (MT * 32), n_base + wave_n * (WN * 32), l31, h);
}

template <int EPI, int MT, int WN>
static hipError_t launchx3_nt_t(hipStream_t s, const float* A, int lda, const void* Bx3, int64_t M, int Nc,
                                int Kd, const GemmEpiParams& ep) {
  constexpr int TBM = (4 / (X3_BN / (WN * 32))) * MT * 32;
  const int tiles_m = (int)((M + TBM - 1) / TBM);
  const int tiles_n = (Nc + X3_BN - 1) / X3_BN;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
  hipLaunchKernelGGL((gemmx3_nt_kernel<EPI, MT, WN>), dim3((unsigned)(tiles_m_pad * tiles_n)), dim3(256), 0, s,
                     A, lda, (const unsigned short*)Bx3, (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

static int x3_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static std::atomic<int> g_x3_tn_tall{x3_env("WIRE_X3_TN_TALL", 0)};   // 256 x 128 tiles in the weight-gradient kernel
static std::atomic<int> g_x3_tn16{x3_env("WIRE_X3_TN16", 1)};         // 256 x 256 tiles on v_mfma_f32_16x16x32_bf16
// 256-row tiles (4 x 2 MFMA tiles per wave) for the Gabor epilogues of large batches: fewer weight bytes
// per MFMA through the 64 B/clk L1 path (tools/mfma_bf16_probe.hip), 7-9 % faster at N = 262144
static std::atomic<int> g_x3_tall{x3_env("WIRE_X3_TALL", 1)};
static std::atomic<int> g_x3_tall_real{x3_env("WIRE_X3_TALL_REAL", 0)};   // the same for siren / gauss / relu (A/B switch)
int gemmx3_tn16_mode() { return g_x3_tn16; }
// The LDS-DMA 32 x 32 x 16 edition (tools/wire_gemmx3g.hip: the staging A/B of round 2) lives outside the product library;
// a harness that links it registers it here.
static bool (*g_glds_handles)(int, int64_t) = nullptr;
static hipError_t (*g_glds_launch)(hipStream_t, int, const float*, int, const void*, int64_t, int, int,
                                   const GemmEpiParams&) = nullptr;
static int (*g_glds_tune)(const char*, int) = nullptr;
void gemmx3_register_glds(bool (*handles)(int, int64_t),
                          hipError_t (*launch)(hipStream_t, int, const float*, int, const void*, int64_t, int, int,
                                               const GemmEpiParams&),
                          int (*tune)(const char*, int)) {
  g_glds_handles = handles; g_glds_launch = launch; g_glds_tune = tune;
}
int gemmx3_tune_set(const char* key, int value) {
  if (!strcmp(key, "x3_tall") && (value == 0 || value == 1)) { g_x3_tall = value; return 0; }
  if (!strcmp(key, "x3_tn_tall") && (value == 0 || value == 1)) { g_x3_tn_tall = value; return 0; }
  if (!strcmp(key, "x3_tn16") && (value == 0 || value == 1)) { g_x3_tn16 = value; return 0; }
  if (!strcmp(key, "x3_tall_real") && (value == 0 || value == 1)) { g_x3_tall_real = value; return 0; }
  if (gemmx3h_tune_set(key, value) == 0) return 0;
  return g_glds_tune ? g_glds_tune(key, value) : -1;
}

// true when launch_gemmx3_nt runs this epilogue on the 16 x 16 x 32 edition (lean epilogues, optional out store,
// recompute_out) at M rows under the current knobs
bool gemmx3_nt_is_h16(int epi, int64_t M) {
  return !(g_glds_handles && g_glds_handles(epi, M)) && gemmx3h_handles(epi, M);
}

hipError_t launch_gemmx3_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx3, int64_t M,
                            int Nc, int Kd, const GemmEpiParams& ep_in) {
  if (M <= 0) return hipSuccess;
  if (g_glds_handles && g_glds_handles(epi, M)) return g_glds_launch(s, epi, A, lda, Bx3, M, Nc, Kd, ep_in);   // harness only
  if (gemmx3h_handles(epi, M)) return launch_gemmx3h_nt(s, epi, A, lda, Bx3, M, Nc, Kd, ep_in);   // 16 x 16 x 32 edition
  if ((Nc & 63) || (Kd & 31) || (lda & 3) || M > 0x7fffff00LL) return hipErrorInvalidValue;
  // the lean Gabor epilogues share one 32-bit byte offset between their buffers
  GemmEpiParams ep = ep_in;
  if ((epi == EPI_GABOR_FWD || epi == EPI_GABOR_BWD || (epi >= EPI_SIREN_FWD && epi <= EPI_RELU_BWD)) &&
      ep.ld0 != ep.ld1)
    ep.wide = 1;
  if ((double)M * (double)(ep.ld1 > ep.ld0 ? ep.ld1 : ep.ld0) * 4.0 >= 4294967296.0) ep.wide = 1;
  // (256-row tiles for wire2d and the real nets were measured 5 % SLOWER on their steps: they stay on 128 rows)
  if (g_x3_tall_real && M >= 4096) {
    switch (epi) {
      case EPI_SIREN_FWD: return launchx3_nt_t<EPI_SIREN_FWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_GAUSS_FWD: return launchx3_nt_t<EPI_GAUSS_FWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_RELU_FWD: return launchx3_nt_t<EPI_RELU_FWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_SIREN_BWD: return launchx3_nt_t<EPI_SIREN_BWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_GAUSS_BWD: return launchx3_nt_t<EPI_GAUSS_BWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_RELU_BWD: return launchx3_nt_t<EPI_RELU_BWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      default: break;
    }
  }
  if (g_x3_tall && M >= 4096) {
    switch (epi) {
      case EPI_STORE: return launchx3_nt_t<EPI_STORE, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_GABOR_FWD: return launchx3_nt_t<EPI_GABOR_FWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_GABOR_BWD: return launchx3_nt_t<EPI_GABOR_BWD, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      case EPI_GABOR_BWD_FIRST: return launchx3_nt_t<EPI_GABOR_BWD_FIRST, 4, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
      default: break;
    }
  }
  switch (epi) {
    case EPI_STORE: return launchx3_nt_t<EPI_STORE, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR_FWD: return launchx3_nt_t<EPI_GABOR_FWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR_BWD: return launchx3_nt_t<EPI_GABOR_BWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR_BWD_FIRST: return launchx3_nt_t<EPI_GABOR_BWD_FIRST, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_SIREN_FWD: return launchx3_nt_t<EPI_SIREN_FWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GAUSS_FWD: return launchx3_nt_t<EPI_GAUSS_FWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_RELU_FWD: return launchx3_nt_t<EPI_RELU_FWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_SIREN_BWD: return launchx3_nt_t<EPI_SIREN_BWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GAUSS_BWD: return launchx3_nt_t<EPI_GAUSS_BWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_RELU_BWD: return launchx3_nt_t<EPI_RELU_BWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR2D_FWD:
      if (Nc & 127) return hipErrorInvalidValue;
      return launchx3_nt_t<EPI_GABOR2D_FWD, 1, 4>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD: return launchx3_nt_t<EPI_GABOR2D_BWD, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD_FIRST: return launchx3_nt_t<EPI_GABOR2D_BWD_FIRST, 2, 2>(s, A, lda, Bx3, M, Nc, Kd, ep);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------
// TN GEMM: slab[split][Pm][Pn] = sum over the split's rows of G[row][m] Z[row][n];
// bslab[split][m] = sum of G[row][m] (bias gradient; plain fp32 adds in the loader).
// LDS plane of a stage: [16 rows][128 features] bf16, 256-byte rows whose four 64-byte chunks are
// XOR-ed with (row & 3): the 4 rows a transposed read gathers fall on different banks.
// ---------------------------------------------------------------------------
#define X3_TK 16

WIRE_DEVINL s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p)));
}

// MT = 32-feature blocks of G per wave (2 or 4): the tile is (64 MT) features of G x 128 features of Z
template <int MT>
__global__ __launch_bounds__(256, (MT > 2 ? 2 : 3)) void gemmx3_tn_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, long long n, int Pm, int Pn,
    int tiles_n, int nsplit, long long chunk, float* __restrict__ slab, float* __restrict__ bslab, int tiles) {
  constexpr int TM = 64 * MT;                  // G features per tile
  constexpr int GROW = TM * 2;                 // bytes of a G row in LDS (bf16)
  constexpr int GPLANE = X3_TK * GROW, ZPLANE = X3_TK * 256;
  constexpr int TSTAGE = 3 * GPLANE + 3 * ZPLANE;
  constexpr int G_TPR = TM / 4;                // loader threads per G row
  constexpr int G_RPP = 256 / G_TPR;           // G rows per loader pass (8 or 4)
  constexpr int G_NP = X3_TK / G_RPP;          // passes (2 or 4)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TSTAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  // XCD-aware order: the `tiles` output tiles of one row split read the same G/Z rows -> one L2
  const int bb = blockIdx.x;
  const int xcd = bb & 7, idx = bb >> 3;
  const int tile = idx % tiles;
  const int split = (idx / tiles) * 8 + xcd;
  if (split >= nsplit) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m_base = tm * TM, n_base = tn * 128;
  const long long row0 = (long long)split * chunk;
  long long row1 = row0 + chunk;
  if (row1 > n) row1 = n;
  if (row1 <= row0) return;                    // (cannot happen: chunk * nsplit covers n with no empty split)

  // loaders: a float4 of one row per thread and pass; (row & 3) is the same in every pass
  const int g_lrow = tid / G_TPR, g_lc4 = (tid % G_TPR) * 4;
  const int z_lrow = tid >> 5, z_lc4 = (tid & 31) * 4;
  int gcol = m_base + g_lc4; if (gcol > Pm - 4) gcol = Pm - 4;      // ragged last tile: stay in the row
  int zcol = n_base + z_lc4; if (zcol > Pn - 4) zcol = Pn - 4;
  const int g_st = g_lrow * GROW + ((((g_lc4 >> 5) ^ (g_lrow & 3))) << 6) + ((g_lc4 & 31) << 1);
  const int z_st = 3 * GPLANE + z_lrow * 256 + ((((z_lc4 >> 5) ^ (z_lrow & 3))) << 6) + ((z_lc4 & 31) << 1);

  const bool live_m = (m_base + wave_m * (32 * MT)) < Pm;
  const bool live_n = (n_base + wave_n * 64) < Pn;
  const bool wave_live = live_m && live_n;
  const bool do_bias = (bslab != nullptr) && (tn == 0);

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  const int nrows = (int)(row1 - row0);
  const int nk = (nrows + X3_TK - 1) / X3_TK;
  const int nk_full = nrows / X3_TK;           // stages whose 16 rows all exist
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  constexpr int PF = X3_TN_PF;
  struct Staged { f32x4 g[G_NP], z[2]; };
  Staged st[PF];
  // addressing: one wave-uniform row base per stage (scalar registers) + per-thread 32-bit offsets that never
  // change -> no vector ALU work per stage.  The offsets of the passes are clamped once to the split's last
  // row, so a stage past the end (prefetch) or the ragged last stage re-reads valid rows.
  unsigned g_off[G_NP], z_off[2];
#pragma unroll
  for (int p = 0; p < G_NP; ++p) g_off[p] = (unsigned)(g_lrow + G_RPP * p) * (unsigned)ldg + (unsigned)gcol;
#pragma unroll
  for (int p = 0; p < 2; ++p) z_off[p] = (unsigned)(z_lrow + 8 * p) * (unsigned)ldz + (unsigned)zcol;
  const float* const g_base = G + (size_t)row0 * ldg;
  const float* const z_base = Z + (size_t)row0 * ldz;
  auto gload = [&](Staged& R, int kt) {
    // full stages only; anything else points at the last full stage or, when the split has none, at row 0
    int ks = kt < nk_full ? kt : nk_full - 1;
    ks = ks < 0 ? 0 : ks;
    const float* gb = g_base + (size_t)ks * (X3_TK * (size_t)ldg);
    const float* zb = z_base + (size_t)ks * (X3_TK * (size_t)ldz);
#pragma unroll
    for (int p = 0; p < G_NP; ++p) R.g[p] = *reinterpret_cast<const f32x4*>(gb + g_off[p]);
#pragma unroll
    for (int p = 0; p < 2; ++p) R.z[p] = *reinterpret_cast<const f32x4*>(zb + z_off[p]);
  };
  // the ragged last stage (only the last split of a row count that is not a multiple of 16 has one)
  auto gload_tail = [&](Staged& R) {
    const int rb = nk_full * X3_TK;
#pragma unroll
    for (int p = 0; p < G_NP; ++p) {
      const int row = rb + g_lrow + G_RPP * p;
      R.g[p] = row < nrows ? *reinterpret_cast<const f32x4*>(g_base + (size_t)row * ldg + gcol) : zero4;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rb + z_lrow + 8 * p;
      R.z[p] = row < nrows ? *reinterpret_cast<const f32x4*>(z_base + (size_t)row * ldz + zcol) : zero4;
    }
  };
  auto lstore = [&](const Staged& R, int buf) {
    unsigned char* S = smem + buf * TSTAGE;
#pragma unroll
    for (int p = 0; p < G_NP; ++p) {
      const f32x4 gv = R.g[p];
      const Split2 s0 = split2(gv[0], gv[1]), s1 = split2(gv[2], gv[3]);
      unsigned char* d = S + g_st + p * (G_RPP * GROW);
      *reinterpret_cast<u32x2*>(d) = u32x2{s0.h, s1.h};
      *reinterpret_cast<u32x2*>(d + GPLANE) = u32x2{s0.m, s1.m};
      *reinterpret_cast<u32x2*>(d + 2 * GPLANE) = u32x2{s0.l, s1.l};
      if (do_bias) {   // scalar adds: a packed v_pk_add_f32 costs the MFMA stream more than two v_add_f32
        bsum[0] += gv[0]; bsum[1] += gv[1]; bsum[2] += gv[2]; bsum[3] += gv[3];
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 zv = R.z[p];
      const Split2 s0 = split2(zv[0], zv[1]), s1 = split2(zv[2], zv[3]);
      unsigned char* d = S + z_st + p * (8 * 256);
      *reinterpret_cast<u32x2*>(d) = u32x2{s0.h, s1.h};
      *reinterpret_cast<u32x2*>(d + ZPLANE) = u32x2{s0.m, s1.m};
      *reinterpret_cast<u32x2*>(d + 2 * ZPLANE) = u32x2{s0.l, s1.l};
    }
  };

  if (nk_full > 0) gload(st[0], 0); else gload_tail(st[0]);
  lstore(st[0], 0);
  __syncthreads();

  // transposed fragment reads: lane -> group g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3:
  // row 8 (g >> 1) + q (+4 for the second read), features 16 (g & 1) + 4 p of a 32-feature block
  const int q = (lane >> 2) & 3;
  const int sub = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  int g_rd[MT], z_rd[2];
#pragma unroll
  for (int i = 0; i < MT; ++i) g_rd[i] = (8 * h + q) * GROW + sub + ((((wave_m * MT + i) ^ q)) << 6);
#pragma unroll
  for (int i = 0; i < 2; ++i) z_rd[i] = 3 * GPLANE + (8 * h + q) * 256 + sub + ((((wave_n * 2 + i) ^ q)) << 6);

  // stage kt: start the loads of stage kt + 1 (mode 1: a full stage, 2: the ragged tail, 0: nothing),
  // MFMAs of stage kt from LDS buffer kt & 1, then split + store stage kt + 1 into the other buffer
  auto stage = [&](const int kt, const int mode) {
    const int buf = kt & 1;
    if (mode == 1) gload(st[0], kt + 1);
    if (mode == 2) gload_tail(st[0]);
    __builtin_amdgcn_sched_barrier(0);          // keep the loads at the top: the compiler would sink them to their use
    {
      const unsigned char* S = smem + buf * TSTAGE;
      bf16x8 gf[MT][3], zf[2][3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const s16x4 g0 = lds_tr16(S + p * GPLANE + g_rd[i]);
          const s16x4 g1 = lds_tr16(S + p * GPLANE + g_rd[i] + 4 * GROW);
          gf[i][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const s16x4 z0 = lds_tr16(S + p * ZPLANE + z_rd[i]);
          const s16x4 z1 = lds_tr16(S + p * ZPLANE + z_rd[i] + 4 * 256);
          zf[i][p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(z0, z1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          X3_MFMA(gf[i][0], zf[j][2], acc[i][j]);
          X3_MFMA(gf[i][2], zf[j][0], acc[i][j]);
          X3_MFMA(gf[i][1], zf[j][1], acc[i][j]);
          X3_MFMA(gf[i][0], zf[j][1], acc[i][j]);
          X3_MFMA(gf[i][1], zf[j][0], acc[i][j]);
          X3_MFMA(gf[i][0], zf[j][0], acc[i][j]);
        }
    }
    if (mode != 0) lstore(st[0], buf ^ 1);
    __syncthreads();
  };
  // full stages 0 .. nk_full - 2 each prefetch the next full stage; then the last full stage (which prefetches
  // the ragged tail if there is one) and the tail itself
  int kt = 0;
  for (; kt + 1 < nk_full; ++kt) stage(kt, 1);
  if (nk_full > 0) { stage(kt, nk > nk_full ? 2 : 0); ++kt; }
  if (nk > nk_full) stage(kt, 0);

  if (do_bias) {
    // G_RPP loader threads share a feature quad: sum them through LDS (all fragment reads are done)
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<f32x4*>(&red[g_lrow * TM + g_lc4]) = bsum;
    __syncthreads();
    if (tid < TM) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < G_RPP; ++r) v += red[r * TM + tid];
      const int m = m_base + tid;
      if (m < Pm) bslab[(size_t)split * Pm + m] = v;
    }
  }
  if (!wave_live) return;
  float* out = slab + (size_t)split * Pm * Pn;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_base + wave_n * 64 + 32 * j + l31;
      if (col < Pn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m_base + wave_m * (32 * MT) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Pm) out[(size_t)m * Pn + col] = acc[i][j][r];
        }
      }
    }
}

// ---------------------------------------------------------------------------
// TN GEMM, 16 x 16 x 32 edition (both padded widths multiples of 256; knob "x3_tn16").
//
// The 128 x 128 kernel above spends 88 split operations per 48 MFMAs of a wave and stage; with the 16-cycle
// instruction shape that mix is bound by vector issue, not by the matrix cores (probe: -3 %).  Here a workgroup of
// 8 waves owns 256 features of G x 256 features of Z: each of the 16 x (256 + 256) fp32 values of a stage is still
// split once, but feeds twice the MFMAs -- 88 vector operations beside 96 MFMAs per wave and stage.
//
//  * plane pairing as in wire_gemmx3h.hip: lanes 0-31 of a fragment carry the 16 rows (k) of the stage for one
//    plane, lanes 32-63 the same 16 rows for another, so one instruction sums two of the six partial products:
//        Z operand (h | l) x G operand (l | h)      ->  h l + l h
//        Z operand (m | h) x G operand (m | m)      ->  m m + h m
//        Z operand (m | h) x G operand (h | h)      ->  m h + h h
//    With the planes pre-split in LDS the pairing costs nothing: the two halves of a wave read different planes.
//  * LDS plane: [16-feature block][slot(row)][32 bytes], slot = (row & 3) | row bit 3 << 2 | row bit 2 << 3, so the
//    (rows q, 8 + q; 4 features each) gather of a transposed read is 256 contiguous bytes per half wave, the second
//    read of a fragment sits +256 bytes, a feature block +512 bytes: every fragment address is one per-lane base
//    plus an immediate.  The loader writes the same 256-byte runs (lane = block half, row slot, feature quad).
//  * Z is the first MFMA operand: a 16 x 16 accumulator block holds, per lane, G feature (lane & 15) and four
//    consecutive Z features -> 16-byte slab stores, re-paired to full 128-byte lines by two DPP moves per register.
//  * 2 x 48 KB stage buffers, one workgroup per CU; tiles x splits = 256 workgroups for 512 x 512.
// ---------------------------------------------------------------------------
typedef float x3f32x4 __attribute__((ext_vector_type(4)));
#define X3_MFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define X3T_PLANE (X3_TK * 512)                 // bytes of one 256-feature bf16 plane of a stage
#define X3T_STAGE (6 * X3T_PLANE)               // G h, m, l then Z h, m, l

// harness-only ablation switches (tools/build_tn_tune.sh; results wrong): 1 no global loads, 2 no split / LDS
// stores, 4 no fragment reads, 8 no slab stores
#ifdef WIRE_ABLATE_TN
#define TN_ABL(x) ((abl & (x)) != 0)
#define TN_ABL_PARAM , int abl
#else
#define TN_ABL(x) false
#define TN_ABL_PARAM
#endif

__global__ __launch_bounds__(512, 2) void gemmx3_tn16_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, long long n, int Pm, int Pn,
    int tiles_n, int nsplit, long long chunk, float* __restrict__ slab, float* __restrict__ bslab, int tiles
    TN_ABL_PARAM) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;     // 64 features of G x 128 features of Z per wave
  const int bb = blockIdx.x;
  const int xcd = bb & 7, idx = bb >> 3;
  const int tile = idx % tiles;
  const int split = (idx / tiles) * 8 + xcd;
  if (split >= nsplit) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m_base = tm * 256, n_base = tn * 256;
  const long long row0 = (long long)split * chunk;
  long long row1 = row0 + chunk;
  if (row1 > n) row1 = n;
  if (row1 <= row0) return;

  // loader: wave w owns feature blocks 2w, 2w + 1 of both operands; a lane = (block half, row slot 0-7, quad p);
  // pass t adds 4 to the row: rows {0-3, 8-11} then {4-7, 12-15}
  const int l_fb = 2 * wave + (lane >> 5), l_rs = (lane >> 2) & 7, l_p = lane & 3;
  const int l_row = (l_rs & 3) + 8 * (l_rs >> 2);
  const int l_feat = 16 * l_fb + 4 * l_p;
  const int l_st = l_fb * 512 + l_rs * 32 + l_p * 8;   // + 256 for pass 1
  const bool do_bias = (bslab != nullptr) && (tn == 0);

  x3f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = x3f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  const int nrows = (int)(row1 - row0);
  const int nk = (nrows + X3_TK - 1) / X3_TK;
  const int nk_full = nrows / X3_TK;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  struct Staged { f32x4 g[2], z[2]; };
  Staged st;
  unsigned g_off[2], z_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    g_off[t] = (unsigned)(l_row + 4 * t) * (unsigned)ldg + (unsigned)(m_base + l_feat);
    z_off[t] = (unsigned)(l_row + 4 * t) * (unsigned)ldz + (unsigned)(n_base + l_feat);
  }
  const float* const g_base = G + (size_t)row0 * ldg;
  const float* const z_base = Z + (size_t)row0 * ldz;
  auto gload = [&](Staged& R, int kt) {
    int ks = kt < nk_full ? kt : nk_full - 1;
    ks = ks < 0 ? 0 : ks;
    const float* gb = g_base + (size_t)ks * (X3_TK * (size_t)ldg);
    const float* zb = z_base + (size_t)ks * (X3_TK * (size_t)ldz);
#pragma unroll
    for (int t = 0; t < 2; ++t) R.g[t] = *reinterpret_cast<const f32x4*>(gb + g_off[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) R.z[t] = *reinterpret_cast<const f32x4*>(zb + z_off[t]);
  };
  auto gload_tail = [&](Staged& R) {
    const int rb = nk_full * X3_TK;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = rb + l_row + 4 * t;
      R.g[t] = row < nrows ? *reinterpret_cast<const f32x4*>(g_base + (size_t)row * ldg + m_base + l_feat) : zero4;
      R.z[t] = row < nrows ? *reinterpret_cast<const f32x4*>(z_base + (size_t)row * ldz + n_base + l_feat) : zero4;
    }
  };
  auto lstore = [&](const Staged& R, int buf) {
    unsigned char* S = smem_t + buf * X3T_STAGE + l_st;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 gv = R.g[t];
      const Split2 s0 = split2(gv[0], gv[1]), s1 = split2(gv[2], gv[3]);
      unsigned char* d = S + t * 256;
      *reinterpret_cast<u32x2*>(d) = u32x2{s0.h, s1.h};
      *reinterpret_cast<u32x2*>(d + X3T_PLANE) = u32x2{s0.m, s1.m};
      *reinterpret_cast<u32x2*>(d + 2 * X3T_PLANE) = u32x2{s0.l, s1.l};
      if (do_bias) { bsum[0] += gv[0]; bsum[1] += gv[1]; bsum[2] += gv[2]; bsum[3] += gv[3]; }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 zv = R.z[t];
      const Split2 s0 = split2(zv[0], zv[1]), s1 = split2(zv[2], zv[3]);
      unsigned char* d = S + 3 * X3T_PLANE + t * 256;
      *reinterpret_cast<u32x2*>(d) = u32x2{s0.h, s1.h};
      *reinterpret_cast<u32x2*>(d + X3T_PLANE) = u32x2{s0.m, s1.m};
      *reinterpret_cast<u32x2*>(d + 2 * X3T_PLANE) = u32x2{s0.l, s1.l};
    }
  };

  // fragment reads: 16-lane group g = lane >> 4 -> rows 8 (g & 1) + q, plane by g >> 1; the byte offset inside a
  // feature block is (lane & 31) * 8 for the first read, + 256 for the second (rows + 4)
  const int hi = lane >> 5;
  const int r_lane = (lane & 31) * 8;
  // plane order in LDS: 0 = h, 1 = m, 2 = l
  const int g_lh = r_lane + (hi ? 0 : 2) * X3T_PLANE + wave_m * (4 * 512);          // G (l | h)
  const int g_c = r_lane + wave_m * (4 * 512);                                       // G m / G h: + plane
  const int z_hl = 3 * X3T_PLANE + r_lane + (hi ? 2 : 0) * X3T_PLANE + wave_n * (8 * 512);   // Z (h | l)
  const int z_mh = 3 * X3T_PLANE + r_lane + (hi ? 0 : 1) * X3T_PLANE + wave_n * (8 * 512);   // Z (m | h)

  auto frag = [&](const unsigned char* p) {
    const s16x4 a = lds_tr16(p), b = lds_tr16(p + 256);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  auto mfma_block = [&](const int buf) {
    const unsigned char* S = smem_t + buf * X3T_STAGE;
    bf16x8 g1[4], g2[4], g3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g1[i] = frag(S + g_lh + i * 512);
      g2[i] = frag(S + g_c + X3T_PLANE + i * 512);
      g3[i] = frag(S + g_c + i * 512);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16x8 f1 = frag(S + z_hl + j * 512);
      const bf16x8 f2 = frag(S + z_mh + j * 512);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        X3_MFMA16(f1, g1[i], acc[i][j]);
        X3_MFMA16(f2, g2[i], acc[i][j]);
        X3_MFMA16(f2, g3[i], acc[i][j]);
      }
    }
  };
  // stage s of the split into the staging registers: a full stage, the ragged tail, or nothing
  auto load_any = [&](const int sidx) {
    if (sidx < nk_full) gload(st, sidx);
    else if (sidx < nk) gload_tail(st);
  };
  // Two wave kinds, so that the two waves of a SIMD (w and w + 4) are out of phase between the barriers -- one runs
  // its MFMAs while the other splits and stores (all waves in phase: 0.69 ms; all early: 0.66; odd waves early, i.e.
  // whole SIMDs in phase: 0.65; this: 0.60 -- profiles/r02b_gemm_tn16_ab.txt):
  //   late  (waves 0-3): loads of stage kt + 1 | MFMAs of stage kt | split + store stage kt + 1
  //   early (waves 4-7): split + store stage kt + 1 (loaded a stage ago) | loads of stage kt + 2 | MFMAs of stage kt
  // Both write buffer (kt + 1) & 1 during stage kt and read buffer kt & 1: the buffer protocol is the same.
  const bool early = wave >= 4;
  load_any(0);
  lstore(st, 0);
  if (early) load_any(1);
  __syncthreads();

  auto stage = [&](const int kt, const bool fast) {
    const int buf = kt & 1;
    if (!early) {
      if (fast) { if (!TN_ABL(1)) gload(st, kt + 1); } else load_any(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(buf);
      if ((fast || kt + 1 < nk) && !TN_ABL(2)) lstore(st, buf ^ 1);
    } else {
      if ((fast || kt + 1 < nk) && !TN_ABL(2)) lstore(st, buf ^ 1);
      if (fast) { if (!TN_ABL(1)) gload(st, kt + 2); } else load_any(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(buf);
    }
    __syncthreads();
  };
  int kt = 0;
#ifdef WIRE_ABLATE_TN
  if (TN_ABL(4)) {
    // fragments read once, MFMAs on loop-invariant registers
    const unsigned char* S = smem_t;
    bf16x8 g1[4], g2[4], g3[4], f1[2], f2[2];
    for (int i = 0; i < 4; ++i) { g1[i] = frag(S + g_lh + i * 512); g2[i] = frag(S + g_c + X3T_PLANE + i * 512); g3[i] = frag(S + g_c + i * 512); }
    for (int j = 0; j < 2; ++j) { f1[j] = frag(S + z_hl + j * 512); f2[j] = frag(S + z_mh + j * 512); }
    for (; kt + 2 < nk_full; ++kt) {
      if (!TN_ABL(1)) gload(st, kt + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          X3_MFMA16(f1[j & 1], g1[i], acc[i][j]);
          X3_MFMA16(f2[j & 1], g2[i], acc[i][j]);
          X3_MFMA16(f2[j & 1], g3[i], acc[i][j]);
        }
      if (!TN_ABL(2)) lstore(st, (kt & 1) ^ 1);
      __syncthreads();
    }
  }
#endif
  for (; kt + 2 < nk_full; ++kt) stage(kt, true);      // stages kt + 1 and kt + 2 are full ones
  for (; kt < nk; ++kt) stage(kt, false);
  if (TN_ABL(8)) return;

  if (do_bias) {
    // the 8 row-slot lanes of a feature quad: sum through LDS (all fragment reads are done)
    float* red = reinterpret_cast<float*>(smem_t);
    *reinterpret_cast<f32x4*>(&red[l_rs * 256 + l_feat]) = bsum;
    __syncthreads();
    if (tid < 256) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r * 256 + tid];
      bslab[(size_t)split * Pm + m_base + tid] = v;
    }
  }
  // accumulator block (i, j): lane -> G feature 16 i + (lane & 15), Z features 16 j + 4 (lane >> 4) + q.  Pair the
  // blocks j, j + 1 inside each 16-lane row: lane (rr = lane & 7, ch = (lane >> 3) & 1, g = lane >> 4) then holds
  // G feature rr (first result) / rr + 8 (second), Z features 32 (j / 2) + 16 ch + 4 g + q: whole 128-byte lines
  float* out = slab + (size_t)split * Pm * Pn;
  const int rr = lane & 7, cq = 16 * ((lane >> 3) & 1) + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_base + wave_m * 64 + 16 * i + rr;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      x3f32x4 xp, yp;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = __float_as_int(acc[i][2 * jp][q]), y = __float_as_int(acc[i][2 * jp + 1][q]);
        xp[q] = __int_as_float(__builtin_amdgcn_update_dpp(x, y, 0x128, 0xf, 0xc, false));
        yp[q] = __int_as_float(__builtin_amdgcn_update_dpp(y, x, 0x128, 0xf, 0x3, false));
      }
      const int col = n_base + wave_n * 128 + 32 * jp + cq;
      *reinterpret_cast<x3f32x4*>(out + (size_t)m * Pn + col) = xp;
      *reinterpret_cast<x3f32x4*>(out + (size_t)(m + 8) * Pn + col) = yp;
    }
  }
}

// Row splits of the weight gradient: tiles x splits workgroups should fill the 256 CUs of an MI355X with the
// SAME number of resident workgroups each (3, else 2, else 1) -- e.g. K = 181 (9 tiles): 56 splits = 504
// workgroups = 2 per CU, where 64 splits (576 workgroups) would leave a quarter of the CUs with 3 and the
// rest with 2, and the launch waits for the loaded ones.
// (a single 256 x 256 tile -- siren / gauss / relu at 256 features -- needs 256 row splits to fill the chip: callers
// size their slabs for up to 256)
static bool tn16_applies(int Pm, int Pn, int mode) { return mode != 0 && Pm % 256 == 0 && Pn % 256 == 0; }

static int tn_splits_for(int64_t n, int Pm, int Pn, int max_splits, bool tn16) {
  int s = 0;
  if (tn16) {
    // 256 x 256 tiles, one 8-wave workgroup per CU
    const int tiles = (Pm / 256) * (Pn / 256);
    s = tiles >= 256 ? 1 : 256 / tiles;
  } else {
    const int tiles = ((Pm + 127) / 128) * ((Pn + 127) / 128);
    for (int w = 3; w >= 1 && s == 0; --w) {
      const int c = (256 * w) / tiles;
      if (c >= 1 && c <= max_splits) s = c;
    }
    if (s == 0) s = tiles > 768 ? 1 : max_splits;   // the tiles alone fill the chip / fewer tiles than the cap allows
  }
  int64_t by_rows = (n + 255) / 256;          // at least 256 rows per split
  if (by_rows < 1) by_rows = 1;
  if (s > by_rows) s = (int)by_rows;
  if (s > max_splits) s = max_splits;
  if (s < 1) s = 1;
  // what the launcher will actually use once the split length is rounded to whole 32-row double stages: callers that
  // pass this count get no empty split (and no memset of unused slabs)
  if (n < 1) return s;
  long long chunk = (n + s - 1) / s;
  chunk = (chunk + 2 * X3_TK - 1) / (2 * X3_TK) * (2 * X3_TK);
  s = (int)((n + chunk - 1) / chunk);
  return s < 1 ? 1 : s;
}
int gemmx3_tn_splits(int64_t n, int Pm, int Pn, int max_splits) {
  return tn_splits_for(n, Pm, Pn, max_splits, tn16_applies(Pm, Pn, g_x3_tn16));
}
// the larger of the two kernels' counts: scratch sized with it stays valid when "x3_tn16" is switched
int gemmx3_tn_splits_max(int64_t n, int Pm, int Pn, int max_splits) {
  const int a = tn_splits_for(n, Pm, Pn, max_splits, false);
  const int b = tn16_applies(Pm, Pn, 1) ? tn_splits_for(n, Pm, Pn, max_splits, true) : a;
  return a > b ? a : b;
}

hipError_t launch_gemmx3_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                            int Pm, int Pn, int splits, float* slab, float* bslab) {
  // all padded widths are multiples of 64; a ragged last tile clamps its loader column inside the row
  if ((Pm & 63) || (Pn & 63) || (ldg & 3) || (ldz & 3) || splits < 1 || n < 1) return hipErrorInvalidValue;
  const bool tn16 = tn16_applies(Pm, Pn, g_x3_tn16);
  const bool tall = !tn16 && g_x3_tn_tall && (Pm % 256 == 0);
  const int tiles_m = tn16 || tall ? Pm / 256 : (Pm + 127) / 128;
  const int tiles_n = tn16 ? Pn / 256 : (Pn + 127) / 128;
  long long chunk = (n + splits - 1) / splits;
  chunk = (chunk + 2 * X3_TK - 1) / (2 * X3_TK) * (2 * X3_TK);
  // no empty split: shrink the count to what the rounded chunk needs
  const int used = (int)((n + chunk - 1) / chunk);
  const int splits_pad = (used + 7) & ~7;
  dim3 grid((unsigned)(tiles_m * tiles_n * splits_pad));
  // slabs of unused splits must still be defined for the reduce kernels
  if (used < splits) {
    hipError_t e = hipMemsetAsync(slab + (size_t)used * Pm * Pn, 0, (size_t)(splits - used) * Pm * Pn * 4, s);
    if (e != hipSuccess) return e;
    if (bslab) {
      e = hipMemsetAsync(bslab + (size_t)used * Pm, 0, (size_t)(splits - used) * Pm * 4, s);
      if (e != hipSuccess) return e;
    }
  }
  if (tn16) {
    // 96 KB of dynamic LDS needs the opt-in; per launch (a host-side call of about a microsecond), because the attribute
    // belongs to the current device's copy of the function and a process may drive more than one
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemmx3_tn16_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X3T_STAGE);
    if (attr != hipSuccess) return attr;
#ifdef WIRE_ABLATE_TN
    static const int abl = x3_env("WIRE_TN_ABL", 0);
    hipLaunchKernelGGL(gemmx3_tn16_kernel, grid, dim3(512), 2 * X3T_STAGE, s, G, ldg, Z, ldz, (long long)n, Pm, Pn,
                       tiles_n, used, chunk, slab, bslab, tiles_m * tiles_n, abl);
#else
    hipLaunchKernelGGL(gemmx3_tn16_kernel, grid, dim3(512), 2 * X3T_STAGE, s, G, ldg, Z, ldz, (long long)n, Pm, Pn,
                       tiles_n, used, chunk, slab, bslab, tiles_m * tiles_n);
#endif
  } else if (tall)
    hipLaunchKernelGGL(gemmx3_tn_kernel<4>, grid, dim3(256), 0, s, G, ldg, Z, ldz, (long long)n, Pm, Pn, tiles_n,
                       used, chunk, slab, bslab, tiles_m * tiles_n);
  else
    hipLaunchKernelGGL(gemmx3_tn_kernel<2>, grid, dim3(256), 0, s, G, ldg, Z, ldz, (long long)n, Pm, Pn, tiles_n,
                       used, chunk, slab, bslab, tiles_m * tiles_n);
  return hipGetLastError();
}
