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
#include <atomic>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"
#include "wire_gemmh_epi.h"

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
  // (X2 form with scale 1: the epilogue also tracks max |out| / |g_lin| when ep.amax_out is set -- a net whose first layer is
  // a GEMM on this kernel hands out_0 to a 2 x fp16 split GEMM, wire_gemmx2h.hip)
  h_epilogue<EPI, true>(acc, ep, M, m_base + wave * 64, n_base, Nc, lane, smem, wave, rt, 1.f);
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
static std::atomic<int> g_x3_h16{x3h_env("WIRE_X3_H16", 15)};
static std::atomic<int> g_x3h_stagger{x3h_env("WIRE_X3H_STAGGER", 0)};      // 100 MHz ticks (100 = 1 us)
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
