// wire_gemmx2h.hip -- the layer GEMMs of the WIRE hot path as a TWO-way fp16 split on v_mfma_f32_16x16x32_f16.
//
// Arithmetic.  Every fp32 operand x is scaled by a power of two s (exact) and split into two fp16 terms,
//     x s = h + l,   h = fp16(x s),  l = fp16(x s - h)        (round-to-nearest at both levels: |x s - h - l| <= 2^-23 |x s|
//                                                               worst case -- 23 of fp32's 24 bits)
// and a product is accumulated from the three partial products of weight >= 2^-12 of the leading one:
//     a b  ~=  a_h b_l + a_l b_h + a_h b_h                     (dropped: a_l b_l <= 2^-22 |a b|)
// Each partial product of two fp16 numbers is exact in fp32 and the MFMA accumulates in fp32.  Measured against fp64
// (tools/f16x2_numerics.hip, profiles/r03_f16x2_numerics.txt; K = 512, five data distributions including gradient-like
// 1e-7-magnitude operands): rms error 0.64 - 0.78 x that of the exact-fp32 MFMA chain (v_mfma_f32_32x32x2_f32) and
// 0.76 - 0.80 x that of the six-product 3 x bf16 split of wire_gemmx3.hip -- fp32 arithmetic in the sense of the parity
// protocol, at HALF the matrix-core work of the 3 x bf16 split (3 instead of 6 MFMAs per fp32 product).
//
// Range.  fp16 has 5 exponent bits, so the scale matters: s maps the operand tensor's max |value| into [2^14, 2^15)
// (wire_dev.h: wire_x2_scales).  Then h is normal for every element within 2^-28 of the maximum and the error of an
// element is max(2^-24 |x|, 2^-39 max|x|) (l subnormal: gfx950's f16 MFMA honours subnormal inputs -- probed; were they
// flushed the floor would be 2^-28 max|x|, still below fp32's own epsilon relative to the maximum).  The maximum of each
// operand tensor is tracked by its PRODUCER (wire_gemmh_epi.h epilogues, first_fwd_kernel, final_fused_kernel: one
// atomicMax per wave into 64 sharded slots) and read at the consumer's start: no host round trip.  Results are
// multiplied by 1 / (s_A s_B) in the epilogue (exact).
//
// NT kernel (forward and data gradient), C[M][Nc] = A[M][Kd] Bt[Nc][Kd]^T + fused epilogue:
//  * 256 x 128 tile, 4 waves stacked in M (64 rows x 128 columns each), stages of 32 reduction indices: one MFMA
//    covers the 32 k of a stage for ONE partial product -- three 16-cycle instructions per 16 x 16 block and stage,
//    against six (as three paired ones per 16 k) in wire_gemmx3h.hip;
//  * A (AMODE 2, default): the wave's 64 rows x 128 bytes of a stage go by LDS-DMA into a WAVE-PRIVATE, single-buffered
//    8 KB region in WHOLE cache lines (an instruction fetches 8 rows x 128 B; the bank swizzle sits on the source address),
//    which the wave reads into registers at the top of a stage and refills right away -- no barrier, no second buffer -- and
//    splits in registers: 24 vector ops per 16-row block and stage (v_cvt_pk_f16_f32, v_fma_mix_f32), every lane productive --
//    96 per wave and 32 k where the 3 x bf16 kernel spends 224.  (AMODE 1: every lane loads ITS fragment -- row lane & 15, k
//    slot lane >> 4 -- straight from global memory into registers one stage ahead; AMODE 0: half-line LDS-DMA pieces: the
//    editions of the first half of round 3, 6 % slower per step: twice the line requests on the vector-memory path.)
//  * B: the pre-split weight image (x2_split_b_kernel), two 8 KB planes per stage, double-buffered by LDS-DMA; the
//    image is stored in FRAGMENT order [16-column block][k slot][column][8 k], so a fragment read is lane * 16 bytes
//    (conflict-free, one address register) and a DMA piece is 1 KB as it lies;
//  * LDS 64 KB (32 KB of weight stages + 32 KB the epilogue reductions use) -> 2 workgroups per CU; one s_barrier per stage;
//  * epilogues: wire_gemmh_epi.h (shared with wire_gemmx3h.hip), with the scale and the maximum tracking.
//
// Replaces the ATen complex addmm / mm of modules/wire.py:89 and of its autograd backward (and the real addmm of
// modules/siren.py:49, gauss.py:28, relu.py:29, wire2d.py:57-58) at M >= 4096.
#include <cstdlib>
#include <atomic>
#include <cstring>

#include "wire_dev.h"
#include "wire_gemm.h"
#include "wire_gemmh_epi.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned x2u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x2u32x2 __attribute__((ext_vector_type(2)));

#define X2_BK 32
#define X2_TBM 256
#define X2_TBN 128
#define X2_AWAVE (64 * 128)            // a wave's private A region: 64 rows x 32 k fp32
#define X2_ABYTES (4 * X2_AWAVE)
#define X2_BPLANE (X2_TBN * 64)        // 128 columns x 32 k fp16
#define X2_BSTAGE (2 * X2_BPLANE)      // planes h, l

#define X2_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// LDS-DMA of 16 bytes per lane (1 KB per wave instruction), as inline assembly rather than
// __builtin_amdgcn_global_load_lds (round 4): with the builtin in a loop the compiler's wait-count pass degrades every
// LDS-read wait of that loop to lgkmcnt(0) -- each fragment read then waits for all the reads issued after it (found with
// the fused forward kernel, wire_fused.hip: fx_dma16; a 20-line kernel shows 14 x lgkmcnt(2) without the builtin and
// 7 x lgkmcnt(0) with it).  The DMA is synchronised by hand in these kernels anyway (vmcnt + barrier).  "x2_dma_asm" = 0
// (WIRE_X2_DMA_BUILTIN at build time) keeps the builtin for the A/B.
WIRE_DEVINL void x2_dma16(const void* gsrc, unsigned char* lds_piece) {
#ifdef WIRE_X2_DMA_BUILTIN
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
#else
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_piece;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(la) : "memory");
#endif
}

// ---------------------------------------------------------------------------
// max |value| of a tensor into its sharded slots (weights once per optimizer step; activations of the per-layer API)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_batch_kernel(X2AmaxBatch ab, long long count) {
  const float* __restrict__ src = ab.src[blockIdx.y];
  float m = 0.f;
  const long long n4 = count >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    h_amax4(m, *reinterpret_cast<const f32x4*>(src + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) m = __builtin_fmaxf(m, __builtin_fabsf(src[4 * n4 + threadIdx.x]));
  wire_amax_publish(ab.slots[blockIdx.y], m, threadIdx.x & 63);
}
hipError_t launch_amax_batch(hipStream_t s, const X2AmaxBatch& ab, int nb, int64_t count) {
  if (nb < 1 || nb > X2_AMAX_MAXB || count < 0) return hipErrorInvalidValue;
  if (count == 0) return hipSuccess;
  long long blocks = (count / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(amax_batch_kernel, dim3((unsigned)blocks, nb), dim3(256), 0, s, ab, (long long)count);
  return hipGetLastError();
}
hipError_t launch_amax(hipStream_t s, const float* src, int64_t count, unsigned* slots) {
  X2AmaxBatch ab{};
  ab.src[0] = src; ab.slots[0] = slots;
  return launch_amax_batch(s, ab, 1, count);
}

// ---------------------------------------------------------------------------
// weights: fp32 image Bt[Nc][ldb] -> scaled, split, fragment-ordered fp16 image
//   Bx2[col tile ct][stage kt (32 k)][plane (h, l)][16-col block cb (8)][k slot ks (4)][col (16)][8 k]
// so that the 16 KB of one (tile, stage) are contiguous and lie as the GEMM's LDS stage does.  Columns >= Nc of the last
// tile are zero.
// ---------------------------------------------------------------------------
__global__ void x2_split_b_batch_kernel(X2SplitBatch sb, int ldb, int Nc, int Kd, int nk) {
  const float* __restrict__ Bt = sb.src[blockIdx.z];
  unsigned short* __restrict__ Bx2 = (unsigned short*)sb.dst[blockIdx.z];
  float s, inv;
  wire_x2_scales(wire_amax_read(sb.slots[blockIdx.z], threadIdx.x & 63), s, inv);
  const int k2 = (blockIdx.x * blockDim.x + threadIdx.x) * 2;   // pair of reduction indices
  const int j = blockIdx.y;                                     // column, < tiles_n * 128
  if (k2 >= Kd) return;
  float x0 = 0.f, x1 = 0.f;
  if (j < Nc) { x0 = Bt[(size_t)j * ldb + k2]; x1 = Bt[(size_t)j * ldb + k2 + 1]; }
  unsigned H, L;
  x2_split2(x0, x1, s, H, L);
  const int ct = j >> 7, c = j & 127, cb = c >> 4, col = c & 15;
  const int kt = k2 >> 5, kk = k2 & 31, ks = kk >> 3, k8 = kk & 7;
  const size_t base = (((size_t)(ct * nk + kt) * 2) * 8 + cb) * 512 + (ks * 16 + col) * 8 + k8;
  *reinterpret_cast<unsigned*>(Bx2 + base) = H;
  *reinterpret_cast<unsigned*>(Bx2 + base + 8 * 512) = L;
}
int64_t gemmx2_b_image_floats(int Nc, int Kd) { return (int64_t)((Nc + 127) / 128) * 128 * Kd; }
hipError_t launch_x2_split_b_batch(hipStream_t s, const X2SplitBatch& sb, int nb, int ldb, int Nc, int Kd) {
  if ((Kd & 31) || nb < 1 || nb > X2_SPLIT_MAXB) return hipErrorInvalidValue;
  const int tiles_n = (Nc + 127) / 128;
  dim3 grid((unsigned)((Kd / 2 + 127) / 128), (unsigned)(tiles_n * 128), (unsigned)nb);
  hipLaunchKernelGGL(x2_split_b_batch_kernel, grid, dim3(128), 0, s, sb, ldb, Nc, Kd, Kd / 32);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// NT GEMM
// ---------------------------------------------------------------------------
// AMODE 0: A through LDS as described above.  AMODE 1: A straight from global memory into the fragment registers of the
// NEXT stage (8 global_load_dwordx4 per lane and stage, in flight under the MFMAs of the current one): no LDS-DMA pieces,
// no LDS reads and no wait for them at the top of a stage for A; LDS holds the weight stages only (32 KB).
// NRB = 16-row blocks per wave: 4 -> 256 x 128 tile, 128 accumulator registers, 2 workgroups per CU; 2 -> 128 x 128 tile, 64
// accumulator registers, <= 168 registers in all: THREE workgroups per CU (AMODE 1 only: LDS holds the weight stages alone).
// PFD = stages of prefetch distance: 1 -> the rows and weight pieces of stage kt + 1 go out during stage kt; 2 (NRB 2, AMODE 1
// only: it needs a second set of raw-fragment registers and a third weight buffer) -> those of stage kt + 2.
// APRE: A is a pre-split activation (wire_dev.h: wire_store_out4; scale fixed on the host, ep.a_split_inv): the 32 bytes a
// lane loads per 16-row block ARE its two fragments up to a register shuffle -- no vector arithmetic between the loads and
// the MFMAs (measured with the loaded bytes taken as the fragments, profiles/r03_gemm_x2_presplit_probe.txt: - 7 % per launch).
template <int EPI, int AMODE, int NRB, int PFD = 1, bool APRE = false>
__global__ __launch_bounds__(256, (NRB == 2 ? 3 : 2)) void gemmx2h_nt_kernel(const float* __restrict__ A, int lda,
                                                            const unsigned short* __restrict__ Bx2, int M, int Nc,
                                                            int Kd, int tiles_m, int tiles_n, GemmEpiParams ep) {
  static_assert(NRB == 4 || AMODE == 1, "the 128-row tile loads A straight into registers");
  // PFD = 3 (NRB 4, AMODE 1): the WEIGHT pieces two stages ahead in a ring of three stage buffers (80 KB of LDS, still two
  // workgroups per CU), the rows one stage ahead as with PFD = 1: the stage-end barrier then never waits for a weight piece
  static_assert(PFD == 1 || (PFD == 2 && NRB == 2 && AMODE == 1) || (PFD == 3 && NRB == 4 && AMODE == 1),
                "two stages of prefetch: 128-row tile, A in registers; weight ring: the default tile");
  constexpr bool BRING = PFD == 3;
  static_assert(!APRE || ((AMODE == 1 || AMODE == 2) && NRB == 4 && PFD == 1), "pre-split A: the 256-row editions only");
  constexpr int A_LDS = NRB == 4 ? X2_ABYTES : 0;      // (AMODE 1 at NRB 4: this part only serves the epilogues' reductions)
  constexpr int TBM = 64 * NRB;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[A_LDS + (PFD == 1 ? 2 : 3) * X2_BSTAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int b = blockIdx.x;
  const int xcd = b & 7, idx = b >> 3;
  const int ct = idx % tiles_n;
  const int rt = (idx / tiles_n) * 8 + xcd;
  if (rt >= tiles_m) return;
  const int m_base = rt * TBM, n_base = ct * X2_TBN;
  const int nk = Kd / X2_BK;

  // operand scales from the producers' maxima (uniform; a kernel boundary lies between the producers and this read)
  float s_a, inv_a, s_b, inv_b;
  if constexpr (APRE) { s_a = 1.f; inv_a = ep.a_split_inv; }
  else wire_x2_scales(wire_amax_read(ep.amax_a, lane), s_a, inv_a);
  wire_x2_scales(wire_amax_read(ep.amax_b, lane), s_b, inv_b);
  (void)s_b;

  // ---- DMA plan.  A: this wave's own 64 rows, 8 pieces of (16 rows x 64 bytes) per stage -- block rb, k half p: lane =
  // (k slot inside the half, 16-byte half of the slot's 32 bytes, row) lands at [k slot][half][row][16 B] of the block,
  // the order the fragment reads below walk.  Offsets are relative to the tile's first row (32-bit).
  unsigned a_off[NRB];
  {
    const int r = lane & 15, hq = (lane >> 4) & 1, ksl = lane >> 5;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      int row = m_base + wave * (16 * NRB) + rb * 16 + r;
      row = row < M ? row : M - 1;
      a_off[rb] = (unsigned)(row - m_base) * (unsigned)lda * 4u + (unsigned)(ksl * 32 + hq * 16);
    }
  }
  constexpr unsigned a_second = 64u;
  // AMODE 2: the same wave-private region filled in WHOLE CACHE LINES -- a DMA instruction fetches 8 rows x 128 bytes (lane =
  // (row l >> 3, 16-byte chunk), each line touched by exactly one instruction) instead of 16 rows x 64 bytes (each line
  // touched by two): half the line requests on the vector-memory path for the same bytes (profiles/r03_gemm_x2_whole_line.txt:
  // - 14 % per launch).  The image is lane-linear ([row][8 slots of 16 B]), so the bank swizzle sits on the SOURCE: slot p of
  // row r holds chunk p ^ sw(r), sw(r) = bit 1 of r | bit 3 of r << 2.  A fragment read (lane = row r, k slot ks: chunks 2 ks
  // and 2 ks + 1 at slots (2 ks) ^ sw, (2 ks + 1) ^ sw) is conflict-free: a ds_read_b128 lane group holds the 16 rows with
  // k slots that differ by one between rows {0-3, 12-15} and {4-11} (chunks differ by XOR 2), the 16-byte bank slot of a
  // lane is 8 (r & 1) + slot, and within either parity of r the four rows of each set get four different sw with bit 1 clear.
  unsigned a_off2[NRB][2];
  if constexpr (AMODE == 2) {
    const int rr = lane >> 3, pp = lane & 7;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = m_base + wave * (16 * NRB) + rb * 16 + 8 * j + rr;
        row = row < M ? row : M - 1;
        const int sw = ((rr >> 1) & 1) | (j << 2);
        a_off2[rb][j] = (unsigned)(row - m_base) * (unsigned)lda * 4u + (unsigned)((pp ^ sw) * 16);
      }
  }
  const char* const a_tile = reinterpret_cast<const char*>(A + (size_t)m_base * lda);
  // B: the 16 KB of a stage lie in the image as they do in LDS: piece q = wave + 4 j, 1 KB each
  const char* const b_tile = reinterpret_cast<const char*>(Bx2) + (size_t)ct * nk * X2_BSTAGE;
  const unsigned b_off = (unsigned)(wave * 1024 + lane * 16);
  unsigned char* const a_lds = smem + wave * X2_AWAVE;
  // AMODE 1: lane (row r = lane & 15, k slot ks = lane >> 4) loads its own 32 bytes of every block
  unsigned g_off[NRB];
  {
    const int r = lane & 15, ks = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      int row = m_base + wave * (16 * NRB) + rb * 16 + r;
      row = row < M ? row : M - 1;
      g_off[rb] = (unsigned)(row - m_base) * (unsigned)lda * 4u + (unsigned)(ks * 32);
    }
  }
  constexpr size_t a_step = X2_BK * 4;
  f32x4 araw[NRB][2];
  auto aload = [&](int kt) {
    const char* ab = a_tile + (size_t)kt * a_step;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      araw[rb][0] = *reinterpret_cast<const f32x4*>(ab + g_off[rb]);
      araw[rb][1] = *reinterpret_cast<const f32x4*>(ab + g_off[rb] + 16);
    }
  };
  auto issue = [&](int kt, int buf) {
    const char* ab = a_tile + (size_t)kt * (X2_BK * 4);
    if constexpr (AMODE == 0) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        x2_dma16(ab + a_off[rb], a_lds + rb * 2048);
        x2_dma16(ab + a_off[rb] + a_second, a_lds + rb * 2048 + 1024);
      }
    }
    if constexpr (AMODE == 2) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        x2_dma16(ab + a_off2[rb][0], a_lds + rb * 2048);
        x2_dma16(ab + a_off2[rb][1], a_lds + rb * 2048 + 1024);
      }
    }
    const char* bb = b_tile + (size_t)kt * X2_BSTAGE;
    unsigned char* S = smem + A_LDS + buf * X2_BSTAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) x2_dma16(bb + b_off + j * 4096, S + j * 4096);
  };

  f32x4 acc[NRB][8];
#pragma unroll
  for (int i = 0; i < NRB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: A block rb: lane (row r = lane & 15, k slot ks = lane >> 4) reads its 32 bytes as two 16-byte
  // pieces at [ks][0][r], [ks][1][r]; B block cb of a plane: lane * 16
  const int a_rd = wave * X2_AWAVE + (lane >> 4) * 512 + (lane & 15) * 16;
  // AMODE 2: row r of the block at (r >> 3) * 1024 + (r & 7) * 128, chunks 2 ks and 2 ks + 1 at the swizzled slots
  const int a2_sw = (((lane & 15) >> 1) & 1) | ((((lane & 15) >> 3) & 1) << 2);
  const int a2_row = wave * X2_AWAVE + ((lane & 15) >> 3) * 1024 + (lane & 7) * 128;
  const int a2_rd0 = a2_row + (((lane >> 4) * 2) ^ a2_sw) * 16;
  const int a2_rd1 = a2_row + (((lane >> 4) * 2 + 1) ^ a2_sw) * 16;
  const int b_rd = A_LDS + lane * 16;
  // the last column tile of a width that is no multiple of 128 (K = 212: 448) has 64 columns: skip the idle half
  const bool half_tile = Nc - n_base <= 64;

  if constexpr (PFD == 2) {
    // stages come in pairs (nk is even: Kd is a multiple of 64): the even stage of a pair lives in ar0, the odd one in ar1;
    // weight stage kt in buffer kt % 3.  During stage kt the loads of stage kt + 2 go out; the wait at its end leaves exactly
    // those (2 NRB + 4 = 8 memory operations) in flight and guarantees stage kt + 1's.
    f32x4 ar1[NRB][2];
    auto aload_to = [&](f32x4 (&dst)[NRB][2], int kt) {
      const char* ab = a_tile + (size_t)kt * (X2_BK * 4);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        dst[rb][0] = *reinterpret_cast<const f32x4*>(ab + g_off[rb]);
        dst[rb][1] = *reinterpret_cast<const f32x4*>(ab + g_off[rb] + 16);
      }
    };
    auto half_stage = [&](f32x4 (&ar)[NRB][2], const int kt) {
      f16x8 ah[NRB], al[NRB];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        unsigned H[4], L[4];
        x2_split2(ar[rb][0][0], ar[rb][0][1], s_a, H[0], L[0]);
        x2_split2(ar[rb][0][2], ar[rb][0][3], s_a, H[1], L[1]);
        x2_split2(ar[rb][1][0], ar[rb][1][1], s_a, H[2], L[2]);
        x2_split2(ar[rb][1][2], ar[rb][1][3], s_a, H[3], L[3]);
        ah[rb] = __builtin_bit_cast(f16x8, x2u32x4{H[0], H[1], H[2], H[3]});
        al[rb] = __builtin_bit_cast(f16x8, x2u32x4{L[0], L[1], L[2], L[3]});
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool more = kt + 2 < nk;
      if (more) { issue(kt + 2, (kt + 2) % 3); aload_to(ar, kt + 2); }
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* S = smem + b_rd + (kt % 3) * X2_BSTAGE;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if (hb == 1 && half_tile) break;
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
          bh[cq] = *reinterpret_cast<const f16x8*>(S + (4 * hb + cq) * 1024);
          bl[cq] = *reinterpret_cast<const f16x8*>(S + X2_BPLANE + (4 * hb + cq) * 1024);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int cq = 0; cq < 4; ++cq) {
            X2_MFMA(bl[cq], ah[rb], acc[rb][4 * hb + cq]);
            X2_MFMA(bh[cq], al[rb], acc[rb][4 * hb + cq]);
            X2_MFMA(bh[cq], ah[rb], acc[rb][4 * hb + cq]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(2 * NRB + 4 == 8, "the in-flight count of the end-of-stage wait");
    issue(0, 0); aload_to(araw, 0);
    issue(1, 1); aload_to(ar1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt += 2) {
      half_stage(araw, kt);
      half_stage(ar1, kt + 1);
    }
    h_epilogue<EPI, true, NRB>(acc, ep, M, m_base + wave * (16 * NRB), n_base, Nc, lane, smem, wave, rt, inv_a * inv_b);
    return;
  }

  issue(0, 0);
  if constexpr (BRING) { if (nk > 1) issue(1, 1); }
  if constexpr (AMODE == 1) aload(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (AMODE == 0 || AMODE == 2) {
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        if constexpr (AMODE == 2) {
          araw[rb][0] = *reinterpret_cast<const f32x4*>(smem + a2_rd0 + rb * 2048);
          araw[rb][1] = *reinterpret_cast<const f32x4*>(smem + a2_rd1 + rb * 2048);
        } else {
          araw[rb][0] = *reinterpret_cast<const f32x4*>(smem + a_rd + rb * 2048);
          araw[rb][1] = *reinterpret_cast<const f32x4*>(smem + a_rd + rb * 2048 + 256);
        }
      }
      // the region is refilled right away: every read of it must have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    f16x8 ah[NRB], al[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      if constexpr (APRE) {
        // two 16-byte units [h h h h | l l l l] of 4 columns each -> the 8 h and the 8 l of the lane's k slot
        const x2u32x4 u0 = __builtin_bit_cast(x2u32x4, araw[rb][0]), u1 = __builtin_bit_cast(x2u32x4, araw[rb][1]);
        ah[rb] = __builtin_bit_cast(f16x8, x2u32x4{u0[0], u0[1], u1[0], u1[1]});
        al[rb] = __builtin_bit_cast(f16x8, x2u32x4{u0[2], u0[3], u1[2], u1[3]});
      } else {
        unsigned H[4], L[4];
        x2_split2(araw[rb][0][0], araw[rb][0][1], s_a, H[0], L[0]);
        x2_split2(araw[rb][0][2], araw[rb][0][3], s_a, H[1], L[1]);
        x2_split2(araw[rb][1][0], araw[rb][1][1], s_a, H[2], L[2]);
        x2_split2(araw[rb][1][2], araw[rb][1][3], s_a, H[3], L[3]);
        ah[rb] = __builtin_bit_cast(f16x8, x2u32x4{H[0], H[1], H[2], H[3]});
        al[rb] = __builtin_bit_cast(f16x8, x2u32x4{L[0], L[1], L[2], L[3]});
      }
    }
    if constexpr (AMODE == 1) {
      // the raw registers are free again: next stage's rows and weight pieces go out under this stage's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BRING) {
        // rows of the next stage first, then the weight pieces of the one after: the stage-end wait leaves those 4 in flight
        if (kt + 1 < nk) aload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);
      } else
      if (kt + 1 < nk) { issue(kt + 1, buf ^ 1); aload(kt + 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned char* S = smem + b_rd + buf * X2_BSTAGE;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      if (hb == 1 && half_tile) break;
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {
        bh[cq] = *reinterpret_cast<const f16x8*>(S + (4 * hb + cq) * 1024);
        bl[cq] = *reinterpret_cast<const f16x8*>(S + X2_BPLANE + (4 * hb + cq) * 1024);
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
          // small terms first; operands swapped (weights first): the block comes out transposed, four consecutive
          // columns per lane (wire_gemmh_epi.h)
          X2_MFMA(bl[cq], ah[rb], acc[rb][4 * hb + cq]);
          X2_MFMA(bh[cq], al[rb], acc[rb][4 * hb + cq]);
          X2_MFMA(bh[cq], ah[rb], acc[rb][4 * hb + cq]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
    if (BRING && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BRING) buf = buf == 2 ? 0 : buf + 1; else buf ^= 1;
  }
  h_epilogue<EPI, true, NRB>(acc, ep, M, m_base + wave * (16 * NRB), n_base, Nc, lane, smem, wave, rt, inv_a * inv_b);
}

static int x2_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// A/B in one process and on one box (profiles/r03_gemm_x2_nt_tn_ab.txt, N = 262 144, K = 256 complex): A through registers
// 0.452 / 0.539 / 0.559 ms (store / Gabor forward / data gradient) against 0.470 / 0.556 / 0.563 through LDS.  Measured
// and dropped (profiles/r03_gemm_x2_prefetch_ablation.txt): an L2 prefetch of the rows three stages ahead (+ 0.02 ms: the
// cost of the HBM reads is not their latency -- with every A row served from cache the store form takes 0.347 ms, i.e. the
// 0.54 GB of A cost 0.105 ms, what they cost at 5 TB/s beside a matrix pipe that shares the chip's power budget).
// Round 3, late: A through the wave-private LDS region in WHOLE cache lines (AMODE 2, default): 0.382 / 0.496 / 0.519 ms -- the
// half-line pieces of AMODE 0 and the fragment-shaped register loads of AMODE 1 (16 rows x 64 bytes per instruction) put
// twice the line requests on the vector-memory path for the same bytes (profiles/r03_gemm_x2_whole_line.txt).
static std::atomic<int> g_x2_amode{x2_env("WIRE_X2_AMODE", 2)};
// "x2_tn_rows" / WIRE_X2_TN_ROWS: upper bound on the rows ONE weight-gradient workgroup accumulates sequentially in its fp32
// accumulators (0 = the fill-the-chip policy of gemmx2_tn_splits alone).  A shorter chain means more row splits, i.e.
// more slabs for wgrad_reduce_kernel: the knob of the summation-order measurement (tools/wgrad_order_probe.py).
static std::atomic<int> g_x2_tn_rows{x2_env("WIRE_X2_TN_ROWS", 0)};
// "x2_tn_p384" / WIRE_X2_TN_P384: waves of the weight-gradient workgroup at P = 384 (K = 181) -- 8 (default, round 4): 48
// features of G per wave, every SIMD carries two waves; 6: 64 features per wave (round 3)
static std::atomic<int> g_x2_tn_p384{x2_env("WIRE_X2_TN_P384", 8)};
int gemmx2h_tune_get(const char* key) {
  if (!strcmp(key, "x2_amode")) return g_x2_amode;
  if (!strcmp(key, "x2_tn_rows")) return g_x2_tn_rows;
  if (!strcmp(key, "x2_tn_p384")) return g_x2_tn_p384;
  return -1;
}
int gemmx2h_tune_set(const char* key, int value) {
  if (!strcmp(key, "x2_amode") && value >= 0 && value <= 2) { g_x2_amode = value; return 0; }
  if (!strcmp(key, "x2_tn_rows") && value >= 0 && (value == 0 || value >= 256)) { g_x2_tn_rows = value; return 0; }
  if (!strcmp(key, "x2_tn_p384") && (value == 6 || value == 8)) { g_x2_tn_p384 = value; return 0; }
  return -1;
}


template <int EPI>
static hipError_t launchx2h_t(hipStream_t s, const float* A, int lda, const unsigned short* Bx2, int64_t M, int Nc,
                              int Kd, const GemmEpiParams& ep) {
  // (128-row tiles at three workgroups per CU, two stages of prefetch, a ring of three weight stages: bit-identical and measured
  //  no faster -- profiles/r03_gemm_x2_rows128.txt, r03_gemm_x2_prefetch2.txt; they live on in tools/wire_gemmx2h_probe.hip)
  const int tbm = X2_TBM;
  const int tiles_m = (int)((M + tbm - 1) / tbm);
  const int tiles_n = (Nc + X2_TBN - 1) / X2_TBN;
  const int tiles_m_pad = (tiles_m + 7) & ~7;
  const dim3 grid((unsigned)(tiles_m_pad * tiles_n));
  if (ep.a_split_inv != 0.f) {
    // (only the forward forms read an activation; the data gradients read g_lin, which has no a-priori bound)
    if constexpr (EPI == EPI_STORE || EPI == EPI_GABOR_FWD || EPI == EPI_GABOR2D_FWD || EPI == EPI_SIREN_FWD ||
                  EPI == EPI_GAUSS_FWD) {
      if (g_x2_amode == 2)
        hipLaunchKernelGGL((gemmx2h_nt_kernel<EPI, 2, 4, 1, true>), grid, dim3(256), 0, s, A, lda, Bx2, (int)M, Nc, Kd, tiles_m,
                           tiles_n, ep);
      else
        hipLaunchKernelGGL((gemmx2h_nt_kernel<EPI, 1, 4, 1, true>), grid, dim3(256), 0, s, A, lda, Bx2, (int)M, Nc, Kd, tiles_m,
                           tiles_n, ep);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (g_x2_amode == 2)
    hipLaunchKernelGGL((gemmx2h_nt_kernel<EPI, 2, 4>), grid, dim3(256), 0, s, A, lda, Bx2, (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  else if (g_x2_amode == 1)
    hipLaunchKernelGGL((gemmx2h_nt_kernel<EPI, 1, 4>), grid, dim3(256), 0, s, A, lda, Bx2, (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  else
    hipLaunchKernelGGL((gemmx2h_nt_kernel<EPI, 0, 4>), grid, dim3(256), 0, s, A, lda, Bx2, (int)M, Nc, Kd, tiles_m, tiles_n, ep);
  return hipGetLastError();
}

// Bx2: the image of launch_x2_split_b_batch; ep.amax_a / ep.amax_b: the operands' maximum slots
hipError_t launch_gemmx2h_nt(hipStream_t s, int epi, const float* A, int lda, const void* Bx2v, int64_t M, int Nc,
                             int Kd, const GemmEpiParams& ep) {
  if (M <= 0) return hipSuccess;
  if ((Nc & 63) || (Kd & 31) || (lda & 3) || M > 0x7fffff00LL || (!ep.amax_a && ep.a_split_inv == 0.f) || !ep.amax_b)
    return hipErrorInvalidValue;
  // 32-bit row offsets inside a 256-row tile
  if ((int64_t)lda * 4 * X2_TBM > 0x7fffffffLL) return hipErrorInvalidValue;
  const unsigned short* Bx2 = (const unsigned short*)Bx2v;
  switch (epi) {
    case EPI_STORE: return launchx2h_t<EPI_STORE>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR_FWD: return launchx2h_t<EPI_GABOR_FWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR_BWD: return launchx2h_t<EPI_GABOR_BWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR_BWD_FIRST: return launchx2h_t<EPI_GABOR_BWD_FIRST>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_SIREN_FWD: return launchx2h_t<EPI_SIREN_FWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GAUSS_FWD: return launchx2h_t<EPI_GAUSS_FWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_RELU_FWD: return launchx2h_t<EPI_RELU_FWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_SIREN_BWD: return launchx2h_t<EPI_SIREN_BWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GAUSS_BWD: return launchx2h_t<EPI_GAUSS_BWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_RELU_BWD: return launchx2h_t<EPI_RELU_BWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR2D_FWD:
      if (Nc & 127) return hipErrorInvalidValue;
      return launchx2h_t<EPI_GABOR2D_FWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD: return launchx2h_t<EPI_GABOR2D_BWD>(s, A, lda, Bx2, M, Nc, Kd, ep);
    case EPI_GABOR2D_BWD_FIRST: return launchx2h_t<EPI_GABOR2D_BWD_FIRST>(s, A, lda, Bx2, M, Nc, Kd, ep);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------
// TN GEMM (weight gradient): slab[s] = G[rows_s]^T Z[rows_s] -- the 2 x fp16 edition of gemmx3_tn16_kernel
// (wire_gemmx3.hip), same structure, with the workgroup shape a template parameter so that the reference's own widths
// tile without waste:
//     WM x WN waves, each 64 features of G x 128 features of Z  ->  tile (64 WM) x (128 WN)
//     (4, 2): 256 x 256, 8 waves  -- padded widths that are multiples of 256 (K = 256: 512; the 256-feature real nets)
//     (8, 1) x 48: 384 x 128, 8 waves of 48 features of G (RB = 3) -- hidden_features = 256 through the reference's API:
//             K = 181, P = 384 (round 4; round 3's (6, 1): 6 waves x 64 stays behind "x2_tn_p384" = 6)
//     (7, 1): 448 x 128, 7 waves  -- the occupancy net as written (3 x 300 -> K = 212, P = 448; its last Z tile is half)
//   (other widths -- config 1's K = 90, P = 192 -- keep the 128 x 128 3 x bf16 kernel of wire_gemmx3.hip)
//  * the loader splits every fp32 value of a stage once (6 vector ops per 2 values) and stores two fp16 planes per
//    operand in the [16-feature block][row slot][32 bytes] layout whose transposed fragment reads (ds_read_b64_tr_b16)
//    and loader stores are 256-byte runs per half wave; its work units (a pair of feature blocks x 8 row slots) are dealt
//    round-robin to the waves, the unit -> (block pair, pass) map is wave-uniform and folds into scalar address parts;
//  * stages of 32 rows: the h h product takes the 32 rows of a stage in ONE 16 x 16 x 32 MFMA (rows 8 g .. 8 g + 7 in
//    lane group g); the two small products ride together as plane PAIRS in the two k halves of an instruction, once per
//    16 rows:  Z (h | l) x G (l | h) -> h l + l h.  Three MFMAs per 16 x 16 block and 32 rows (six in the 3 x bf16 kernel).
//    Row slot of row r: (r & 3) | ((r >> 3) & 3) << 2 | ((r >> 2) & 1) << 4 -- every fragment address is the lane's base
//    plus an immediate (+ 512: rows + 4; + 256: rows + 16; + 1024: next feature block);
//  * two stage buffers of (64 WM + 128 WN) x 128 bytes (64 KB at (4, 2), (6, 1) and (8, 1) x 48), one workgroup per CU; waves
//    >= 4 (SIMD partners of waves 0-3) run half a stage apart ((4, 2) and (8, 1) x 48);
//  * blockIdx.y = member of a batch of equal-shaped weight gradients (the hidden layers behind wire_fused.hip's chain);
//  * G and Z carry their own power-of-two scales (maximum slots of their producers); the slabs are unscaled on the way out.
// ---------------------------------------------------------------------------
typedef short x2s16x4 __attribute__((ext_vector_type(4)));
#define X2T_TK 32

WIRE_DEVINL x2s16x4 x2_lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (x2s16x4 __attribute__((address_space(3)))*)(const_cast<unsigned char*>(p)));
}

// ZMODE: what the Z rows hold -- 0: fp32 values (scale from their maximum slots); 1: pre-split fp16 pairs (wire_dev.h:
// wire_store_out4; 1 / scale from the host); 2 / 3 (round 4, the fused training forward of siren / gauss nets stores no out_l at
// all): the pre-activation r = c lin the forward's activation was evaluated on -- the loader evaluates sin(2 pi r) (2) or
// exp2(-r^2) (3) again, the very instructions of wire_fused.hip's producer, and splits with the fixed scale 1 / z_pre_inv.
// RB = 16-feature blocks of G per wave (4: a wave owns 64 x 128 of the tile).  RB = 3 with (WM, WN) = (8, 1) is the P = 384
// shape of round 4 (hidden_features = 256 through the reference's API, K = 181): 8 waves x 48 features = 384 x 128 -- the
// (6, 1) shape's 6 waves leave two SIMDs of a CU with one wave and two with two, so the workgroup runs at the pace of 8 waves'
// work on 6 waves' results; with 8 waves of 3 x 8 blocks every SIMD carries two waves of 72 MFMAs per stage (96 before).  Its 12
// block pairs of G deal out to the 8 waves as 6 loader units each: pairs w (passes 0, 2), w + 4 (passes 1, 3) and, for waves
// 0-3, w + 8 (passes 0, 2), for waves 4-7, w - 4 + 8 ... i.e. pair (w + 8) mod 12 with passes (1, 3) -- every pair gets its four
// passes from two waves, whose bias sums meet in the LDS reduction.
template <int WM, int WN, int ZMODE, int RB = 4>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemmx2_tn16_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, long long n, int Pm, int Pn,
    int tiles_n, int nsplit, long long chunk, float* __restrict__ slab, float* __restrict__ bslab, int tiles,
    const unsigned* __restrict__ amax_g, const unsigned* __restrict__ amax_z, const float z_pre_inv,
    const long long g_step, const long long z_step, const int amax_step) {
  // blockIdx.y = member of a BATCH of weight gradients of equal shape (the hidden layers behind the data-gradient chain of
  // wire_fused.hip, whose g_lin_l / activations / maximum slots lie a fixed step apart): one launch fills the chip with
  // nsplit x gridDim.y workgroups, each accumulating gridDim.y times the rows -- a third of the slabs at three layers
  {
    const int by = blockIdx.y;
    G += (size_t)by * g_step; Z += (size_t)by * z_step;
    slab += (size_t)by * nsplit * Pm * Pn;
    if (bslab) bslab += (size_t)by * nsplit * Pm;
    amax_g += by * amax_step;
    if (amax_z) amax_z += by * amax_step;
  }
  constexpr int NW = WM * WN;                       // waves
  constexpr int TM = 16 * RB * WM, TN = 128 * WN;   // tile: features of G x features of Z
  constexpr bool G3 = RB == 3;                      // the (8, 1) x 48-feature shape: explicit loader dealing (above)
  static_assert(RB == 4 || (RB == 3 && WM == 8 && WN == 1), "wave tiles: 64 x 128, or 48 x 128 in the 8-wave P = 384 shape");
  constexpr int GPLANE = X2T_TK * TM * 2, ZPLANE = X2T_TK * TN * 2;   // bytes of one fp16 plane of a stage
  constexpr int STAGE = 2 * GPLANE + 2 * ZPLANE;    // G h, G l, Z h, Z l
  constexpr int UG = TM / 8, UZ = 16 * WN;          // loader unit pairs (2 feature blocks x 8 row slots x 4 quads) per stage
  constexpr int IG = (UG + NW - 1) / NW, IZ = (UZ + NW - 1) / NW;     // ... per wave
  constexpr int PG = TM / 32, PZ = TN / 32;         // feature-block pairs per operand
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;   // (kept a vector value: as a scalar it drives the allocator over its budget)
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int bb = blockIdx.x;
  const int xcd = bb & 7, idx = bb >> 3;
  const int tile = idx % tiles;
  const int split = (idx / tiles) * 8 + xcd;
  if (split >= nsplit) return;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m_base = tm * TM, n_base = tn * TN;
  const long long row0 = (long long)split * chunk;
  long long row1 = row0 + chunk;
  if (row1 > n) row1 = n;
  if (row1 <= row0) return;

  float s_g, inv_g, s_z, inv_z;
  wire_x2_scales(wire_amax_read(amax_g, lane), s_g, inv_g);
  // ZPRE: Z is a pre-split activation (wire_dev.h: wire_store_out4) -- the 16 bytes a loader lane reads are the 8 + 8 bytes
  // it stores into the h and l planes; 1 / its scale comes from the host.  (A template parameter: as a run-time branch of
  // the loader it costs the (WM, 1) shapes 199 spilled registers.)
  constexpr bool ZPRE = ZMODE == 1;
  if constexpr (ZMODE >= 1) { s_z = 1.f / z_pre_inv; inv_z = z_pre_inv; }
  else wire_x2_scales(wire_amax_read(amax_z, lane), s_z, inv_z);

  // loader lane = (block of the pair, row slot 0-7, feature quad); unit pair u = (block pair u % P, pass u / P); pass t
  // writes slots 8 t + (0-7), i.e. rows l_row0 + {0, 16, 4, 20}
  const int l_hb = lane >> 5, l_rs = (lane >> 2) & 7, l_p = lane & 3;
  const int l_row0 = (l_rs & 3) + 8 * (l_rs >> 2);
  auto pass_row = [](const int t) { return 16 * (t & 1) + 4 * (t >> 1); };   // 0, 16, 4, 20
  const int l_feat = 16 * l_hb + 4 * l_p;                 // + 32 (block pair)
  const int l_st = l_hb * 1024 + l_rs * 32 + l_p * 8;     // + 2048 (block pair) + 256 (pass)
  const bool do_bias = (bslab != nullptr) && (tn == 0);
  // a Z tile that sticks out of the row (P = 448: the fourth 128-feature tile has 64): its loads stay inside the row
  const bool z_edge = WN == 1 && n_base + TN > Pn;      // ((4, 2) only runs on widths that are multiples of 256)

  f32x4 acc[RB][8];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // G units of wave w: block pairs w + NW b, b < NB = PG / NW (1 at (4, 2), 2 at (WM, 1)), each in all 4 passes: unit i =
  // (block pair w + NW (i % NB), pass i / NB) -- the pass is a compile-time constant, the block pair one scalar.  Z units:
  // (4, 2): block pair w, pass i, likewise; (WM, 1): 16 units dealt round-robin, u = w + NW i < 16, pair u & 3, pass u >> 2.
  // Bias sums: one running sum per block pair of this wave, all passes added up.
  static_assert(G3 || (PG % NW == 0 && IG == 4 * (PG / NW)), "the block pairs of G must deal out evenly to the waves");
  static_assert(!G3 || (PG == 12 && IG == 6), "8 waves x 6 units = 12 block pairs x 4 passes");
  static_assert(PZ == NW || PZ == 4, "Z units: one block pair per wave, or 4 block pairs dealt round-robin");
  constexpr int NB = G3 ? 3 : PG / NW;              // block pairs (= bias sums) of a wave
  // G3: unit i of wave w = (pair, pass): 0 (w, 0), 1 (pB, hiw), 2 (w + 4, 1), 3 (w, 2), 4 (pB, 2 + hiw), 5 (w + 4, 3) with
  // hiw = w >= 4, pB = (w + 8) mod 12
  const int hiw = wave >= 4 ? 1 : 0;
  const int g3_pb = hiw ? wave - 4 : wave + 8;
  auto g3_pair = [&](const int i) { return (i % 3) == 0 ? wave : ((i % 3) == 1 ? g3_pb : wave + 4); };
  // row of the pass inside the 32-row stage (pass_row below) and its LDS offset (pass x 256)
  auto g3_row = [&](const int i) { return (i % 3) == 1 ? 16 * hiw + (i >= 3 ? 4 : 0) : (i == 0 ? 0 : i == 2 ? 16 : i == 3 ? 4 : 20); };
  auto g3_pass = [&](const int i) { return (i % 3) == 1 ? hiw + (i >= 3 ? 2 : 0) : (i == 0 ? 0 : i == 2 ? 1 : i == 3 ? 2 : 3); };
  constexpr bool ZFIX = (PZ == NW);
  f32x4 bsum[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bsum[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nrows = (int)(row1 - row0);
  const int nk = (nrows + X2T_TK - 1) / X2T_TK;
  const int nk_full = nrows / X2T_TK;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  struct Staged { f32x4 g[IG], z[IZ]; };
  Staged st;
  const unsigned g_off = (unsigned)l_row0 * (unsigned)ldg + (unsigned)(m_base + l_feat);
  const unsigned z_off = (unsigned)l_row0 * (unsigned)ldz + (unsigned)(n_base + l_feat);
  const float* const g_base = G + (size_t)row0 * ldg;
  const float* const z_base = Z + (size_t)row0 * ldz;
  // tail = true: the ragged last stage, rows >= nrows read as zero
  auto gload = [&](Staged& R, int kt, const bool tail) {
    int ks = tail ? nk_full : (kt < nk_full ? kt : nk_full - 1);
    ks = ks < 0 ? 0 : ks;
    const float* gb = g_base + (size_t)ks * (X2T_TK * (size_t)ldg);
    const float* zb = z_base + (size_t)ks * (X2T_TK * (size_t)ldz);
    const int rbase = ks * X2T_TK + l_row0;
#pragma unroll
    for (int i = 0; i < IG; ++i) {
      if constexpr (G3) {
        const int fp = g3_pair(i), pr = g3_row(i);
        const bool ok = !tail || rbase + pr < nrows;
        R.g[i] = ok ? *reinterpret_cast<const f32x4*>(gb + (size_t)pr * ldg + 32 * fp + g_off) : zero4;
      } else {
        const int fp = wave + NW * (i % NB), t = i / NB;
        const bool ok = !tail || rbase + pass_row(t) < nrows;
        R.g[i] = ok ? *reinterpret_cast<const f32x4*>(gb + (size_t)pass_row(t) * ldg + 32 * fp + g_off) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < IZ; ++i) {
      const int u = wave + NW * i;
      if (ZFIX || u < UZ) {
        const int fp = ZFIX ? wave : (u & 3), t = ZFIX ? i : (u >> 2);
        bool ok = !tail || rbase + pass_row(t) < nrows;
        if (z_edge) ok = ok && (n_base + 32 * fp + l_feat < Pn);
        R.z[i] = ok ? *reinterpret_cast<const f32x4*>(zb + (size_t)pass_row(t) * ldz + 32 * fp + z_off) : zero4;
      }
    }
  };
  auto lstore = [&](const Staged& R, int buf) {
    unsigned char* S = smem_t + buf * STAGE + l_st;
#pragma unroll
    for (int i = 0; i < IG; ++i) {
      const int fp = G3 ? g3_pair(i) : wave + NW * (i % NB), t = G3 ? g3_pass(i) : i / NB;
      const f32x4 gv = R.g[i];
      unsigned h0, l0, h1, l1;
      x2_split2(gv[0], gv[1], s_g, h0, l0);
      x2_split2(gv[2], gv[3], s_g, h1, l1);
      unsigned char* d = S + fp * 2048 + t * 256;
      *reinterpret_cast<x2u32x2*>(d) = x2u32x2{h0, h1};
      *reinterpret_cast<x2u32x2*>(d + GPLANE) = x2u32x2{l0, l1};
      if (do_bias) { bsum[i % NB][0] += gv[0]; bsum[i % NB][1] += gv[1]; bsum[i % NB][2] += gv[2]; bsum[i % NB][3] += gv[3]; }
    }
#pragma unroll
    for (int i = 0; i < IZ; ++i) {
      const int u = wave + NW * i;
      if (ZFIX || u < UZ) {
        const int fp = ZFIX ? wave : (u & 3), t = ZFIX ? i : (u >> 2);
        const f32x4 zv = R.z[i];
        unsigned h0, l0, h1, l1;
        if constexpr (ZPRE) {
          const x2u32x4 zu = __builtin_bit_cast(x2u32x4, zv);
          h0 = zu[0]; h1 = zu[1]; l0 = zu[2]; l1 = zu[3];
        } else {
          f32x4 zz = zv;
          if constexpr (ZMODE == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) zz[e] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(zv[e]));
          }
          if constexpr (ZMODE == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) zz[e] = __builtin_amdgcn_exp2f(-(zv[e] * zv[e]));
          }
          x2_split2(zz[0], zz[1], s_z, h0, l0);
          x2_split2(zz[2], zz[3], s_z, h1, l1);
        }
        unsigned char* d = S + 2 * GPLANE + fp * 2048 + t * 256;
        *reinterpret_cast<x2u32x2*>(d) = x2u32x2{h0, h1};
        *reinterpret_cast<x2u32x2*>(d + ZPLANE) = x2u32x2{l0, l1};
      }
    }
  };

  // fragment reads (plane order in LDS: G h, G l, Z h, Z l)
  const int hi = lane >> 5;
  const int g_hh = lane * 8 + wave_m * (RB * 1024);
  const int g_pr = (lane & 31) * 8 + (hi ? 0 : 1) * GPLANE + wave_m * (RB * 1024);                   // G (l | h)
  const int z_hh = 2 * GPLANE + lane * 8 + wave_n * (8 * 1024);
  const int z_pr = 2 * GPLANE + (lane & 31) * 8 + (hi ? 1 : 0) * ZPLANE + wave_n * (8 * 1024);      // Z (h | l)

  auto frag = [&](const unsigned char* p) {
    const x2s16x4 a = x2_lds_tr16(p), b = x2_lds_tr16(p + 512);
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  // three passes over the 8 Z blocks -- pair products of rows 0-15, pair products of rows 16-31, h h -- each holding the
  // G fragments of the wave's 4 blocks for ONE product kind: 16 + 4 fragment registers live beside the 128 accumulator and
  // the 32 - 44 staging registers.  (Both pair products in one pass, 32 + 8 fragment registers: 11 spilled registers at
  // (4, 2) in this templated form -- the dedicated (4, 2) kernel it grew out of held them and ran 0.417 instead of 0.437 ms
  // per launch at K = 256; all three kinds at once, 48 + 12, spill everywhere.  A spill reload waits in vmcnt order for the
  // prefetched global loads, so no spill is tolerated here.)
  auto mfma_block = [&](const int buf) {
    const unsigned char* S = smem_t + buf * STAGE;
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
      const int go = ps == 2 ? g_hh : g_pr + 256 * ps;
      const int zo = ps == 2 ? z_hh : z_pr + 256 * ps;
      f16x8 gf[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) gf[i] = frag(S + go + i * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f16x8 zf = frag(S + zo + j * 1024);
#pragma unroll
        for (int i = 0; i < RB; ++i) X2_MFMA(zf, gf[i], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);               // (or the next pass's fragment reads are hoisted up here)
    }
  };
  // stage s of the split into the staging registers: a full stage, the ragged tail, or nothing
  auto load_any = [&](const int sidx) {
    if (sidx < nk_full) gload(st, sidx, false);
    else if (sidx < nk) gload(st, sidx, true);
  };
  // late (waves 0-3): loads of stage kt + 1 | MFMAs of stage kt | split + store stage kt + 1
  // early (waves >= 4): split + store stage kt + 1 (loaded a stage ago) | loads of stage kt + 2 | MFMAs of stage kt
  // Both write buffer (kt + 1) & 1 during stage kt and read buffer kt & 1: the buffer protocol is the same.
  // (the (WM, 1) shapes stage 11 instead of 8 loads per lane: with both wave kinds in the code the allocator spills ~25
  // registers, and a spill reload waits, in vmcnt order, for the prefetched global loads -- measured 2 x the stage time at
  // P = 384; all their waves run the late schedule)
  // (the 48-feature shape has the registers for both wave kinds: 216 without the stagger)
  const bool early = (WN == 2 || G3) && wave >= 4;
  load_any(0);
  lstore(st, 0);
  if (early) load_any(1);
  __syncthreads();

  auto stage = [&](const int kt, const bool fast) {
    const int buf = kt & 1;
    if (!early) {
      if (fast) gload(st, kt + 1, false); else load_any(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(buf);
      if (fast || kt + 1 < nk) lstore(st, buf ^ 1);
    } else {
      if (fast || kt + 1 < nk) lstore(st, buf ^ 1);
      if (fast) gload(st, kt + 2, false); else load_any(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(buf);
    }
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 2 < nk_full; ++kt) stage(kt, true);      // stages kt + 1 and kt + 2 are full ones
  for (; kt < nk; ++kt) stage(kt, false);

  if (do_bias) {
    // the 8 row-slot lanes of a feature quad and the 4 passes of a block pair: sum through LDS (all fragment reads are done)
    // (G3: a block pair's passes come from two waves -- the even passes from one, the odd ones from another: 16 partial rows)
    constexpr int RR = G3 ? 16 : 8;
    float* red = reinterpret_cast<float*>(smem_t);       // [8 row slots (x 2 pass parities)][TM features]
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int fp = G3 ? g3_pair(i) : wave + NW * i;
      const int hf = G3 ? (g3_pass(i) & 1) : 0;
      *reinterpret_cast<f32x4*>(&red[(l_rs + 8 * hf) * TM + 32 * fp + l_feat]) = bsum[i];
    }
    __syncthreads();
    for (int f = tid; f < TM; f += 64 * NW) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RR; ++r) v += red[r * TM + f];
      bslab[(size_t)split * Pm + m_base + f] = v;
    }
  }
  // accumulator block (i, j): lane -> G feature 16 i + (lane & 15), Z features 16 j + 4 (lane >> 4) + q; blocks j, j + 1
  // re-paired inside each 16-lane row (wire_gemmh_epi.h: h_pair_rows) -> whole 128-byte lines
  float* out = slab + (size_t)split * Pm * Pn;
  const float inv = inv_g * inv_z;
  const int rr = lane & 7, cq = 16 * ((lane >> 3) & 1) + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int m = m_base + wave_m * (16 * RB) + 16 * i + rr;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      f32x4 xp, yp;
      h_pair_rows(acc[i][2 * jp], acc[i][2 * jp + 1], xp, yp);
#pragma unroll
      for (int q = 0; q < 4; ++q) { xp[q] *= inv; yp[q] *= inv; }
      const int col = n_base + wave_n * 128 + 32 * jp + cq;
      if (col < Pn) {                                    // (an edge Z tile: its upper 32-column spans do not exist)
        *reinterpret_cast<f32x4*>(out + (size_t)m * Pn + col) = xp;
        *reinterpret_cast<f32x4*>(out + (size_t)(m + 8) * Pn + col) = yp;
      }
    }
  }
}

// workgroup shape for a weight gradient of Pm x Pn padded features (multiples of 64): 0 = none (3 x bf16 kernels)
static int x2_tn_shape(int Pm, int Pn) {
  if ((Pm & 63) || (Pn & 63) || Pm < 64 || Pn < 64) return 0;
  if (Pm % 256 == 0 && Pn % 256 == 0) return 42;
  if (Pm == 384) return g_x2_tn_p384 == 8 ? 381 : 61;   // 381: 8 waves x 48 features (round 4); 61: 6 waves x 64
  if (Pm == 448) return 71;
  return 0;
}
// tile of a shape code: G features x Z features
static int x2_tn_tm(int shp) { return shp == 381 ? 384 : 64 * (shp / 10); }
static int x2_tn_tn(int shp) { return shp == 381 ? 128 : 128 * (shp % 10); }
bool gemmx2_tn_applies(int Pm, int Pn) { return x2_tn_shape(Pm, Pn) != 0; }
// Row splits that fill the chip with ONE workgroup per CU.  Blocks are dealt round-robin over the 8 XCDs and split s runs
// its tiles on XCD s % 8, so an XCD receives tiles * ceil(splits / 8) workgroups and has 32 CUs: one more than 32 and a CU
// runs two workgroups back to back -- the launch takes twice as long.  (Measured: 3 tiles x 84 splits = 33 workgroups on four
// XCDs: 0.609 ms, waves alive 0.29 ms, profiles/r03_tn61_pmc_summary.txt; 3 x 80: see profiles/r03_gemm_x2_tn_shapes.txt.)
int gemmx2_tn_splits(int64_t n, int Pm, int Pn, int max_splits) {
  const int shp = x2_tn_shape(Pm, Pn);
  if (!shp) return 0;
  const int TMf = x2_tn_tm(shp), TNf = x2_tn_tn(shp);
  const int tiles = (Pm / TMf) * ((Pn + TNf - 1) / TNf);
  int s = tiles >= 32 ? 8 : 8 * (32 / tiles);
  int64_t by_rows = (n + 255) / 256;                     // at least 256 rows per split
  if (by_rows < 1) by_rows = 1;
  if (s > by_rows) s = (int)by_rows;
  if (s > max_splits) s = max_splits;
  if (s < 1) s = 1;
  if (n < 1) return s;
  const int cap_rows = g_x2_tn_rows;                     // bounded accumulation chains: more splits (up to 4096), whatever max_splits
  if (cap_rows > 0) {
    int64_t need = (n + cap_rows - 1) / cap_rows;
    need = (need + 7) / 8 * 8;
    if (need > 4096) need = 4096;
    if (need > s) s = (int)need;
  }
  long long chunk = (n + s - 1) / s;
  chunk = (chunk + 2 * X2T_TK - 1) / (2 * X2T_TK) * (2 * X2T_TK);
  s = (int)((n + chunk - 1) / chunk);
  return s < 1 ? 1 : s;
}

// row splits per member of a batch of `batch` weight gradients (launch_gemmx2_tn): the chip filled once by all members
// together, and exactly as many splits as the launcher's 64-row rounding of the chunk will use
int gemmx2_tn_batch_splits(int64_t n, int Pm, int Pn, int max_splits, int batch) {
  int s = gemmx2_tn_splits(n, Pm, Pn, max_splits);
  if (batch <= 1 || s < batch) return s;
  // an XCD (32 CUs, one workgroup each) receives tiles x ceil(splits / 8) workgroups of EVERY member: 3 members x 11 = 33 ran
  // one workgroup behind the others on four XCDs and the launch took twice as long (the first edition: + 0.1 ms per step)
  const int shp = x2_tn_shape(Pm, Pn);
  const int tiles = (Pm / x2_tn_tm(shp)) * ((Pn + x2_tn_tn(shp) - 1) / x2_tn_tn(shp));
  const int groups = 32 / (tiles * batch);
  if (groups < 1) return 0;
  if (s > 8 * groups) s = 8 * groups;
  long long chunk = (n + s - 1) / s;
  chunk = (chunk + 2 * X2T_TK - 1) / (2 * X2T_TK) * (2 * X2T_TK);
  return (int)((n + chunk - 1) / chunk);
}

template <int WM, int WN, int ZMODE, int RB = 4>
static hipError_t launch_x2_tn_z(hipStream_t s, dim3 grid, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                                 int Pm, int Pn, int tiles_n, int used, long long chunk, float* slab, float* bslab,
                                 int tiles, const unsigned* amax_g, const unsigned* amax_z, float z_pre_inv,
                                 long long g_step, long long z_step, int amax_step) {
  constexpr int STAGE = X2T_TK * (16 * RB * WM + 128 * WN) * 4;
  // > 64 KB of dynamic LDS needs the opt-in; per launch (a host-side call of about a microsecond), because the attribute
  // belongs to the current device's copy of the function and a process may drive more than one
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemmx2_tn16_kernel<WM, WN, ZMODE, RB>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL((gemmx2_tn16_kernel<WM, WN, ZMODE, RB>), grid, dim3(64 * WM * WN), 2 * STAGE, s, G, ldg, Z, ldz,
                     (long long)n, Pm, Pn, tiles_n, used, chunk, slab, bslab, tiles, amax_g, amax_z, z_pre_inv, g_step, z_step,
                     amax_step);
  return hipGetLastError();
}
template <int WM, int WN, int RB = 4>
static hipError_t launch_x2_tn_t(hipStream_t s, dim3 grid, const float* G, int ldg, const float* Z, int ldz, int64_t n,
                                 int Pm, int Pn, int tiles_n, int used, long long chunk, float* slab, float* bslab,
                                 int tiles, const unsigned* amax_g, const unsigned* amax_z, float z_pre_inv, int z_act,
                                 long long g_step, long long z_step, int amax_step) {
#define X2_TN_Z(M) launch_x2_tn_z<WM, WN, M, RB>(s, grid, G, ldg, Z, ldz, n, Pm, Pn, tiles_n, used, chunk, slab, bslab, tiles, amax_g, \
                                            amax_z, z_pre_inv, g_step, z_step, amax_step)
  if constexpr (WM == 4 && WN == 2) {                      // the re-evaluating loaders: the 256-feature real nets only
    if (z_act == 2) return X2_TN_Z(2);
    if (z_act == 3) return X2_TN_Z(3);
  }
  if (z_act >= 2) return hipErrorInvalidValue;
  return z_pre_inv != 0.f ? X2_TN_Z(1) : X2_TN_Z(0);
#undef X2_TN_Z
}

// `splits` from gemmx2_tn_splits; slabs [splits][Pm][Pn] (+ bslab [splits][Pm]) as launch_gemmx3_tn writes them.
// z_pre_inv != 0: Z is a pre-split activation (wire_dev.h: wire_store_out4), 1 / its scale; amax_z is not read.
// z_act = 2 / 3: Z holds the pre-activation r of a sine / Gaussian layer as wire_fused.hip stored it; the loader evaluates the
// activation again and splits with scale 1 / z_pre_inv (which must be given).
// batch > 1: `batch` weight gradients of the same shape in one launch -- member j reads G + j g_step, Z + j z_step and the
// maximum slots amax_g + j amax_step (amax_z likewise) and writes the slabs [j][splits][Pm][Pn] (+ bslab [j][splits][Pm]).
hipError_t launch_gemmx2_tn(hipStream_t s, const float* G, int ldg, const float* Z, int ldz, int64_t n, int Pm,
                            int Pn, int splits, float* slab, float* bslab, const unsigned* amax_g,
                            const unsigned* amax_z, float z_pre_inv, int z_act, int batch, long long g_step,
                            long long z_step, int amax_step) {
  const int shp = x2_tn_shape(Pm, Pn);
  if (!shp || (ldg & 3) || (ldz & 3) || splits < 1 || n < 1 || !amax_g || (!amax_z && z_pre_inv == 0.f) ||
      (z_act != 0 && (z_act < 2 || z_act > 3 || z_pre_inv == 0.f)) || batch < 1 || batch > 8)
    return hipErrorInvalidValue;
  const int TMf = x2_tn_tm(shp), TNf = x2_tn_tn(shp);
  const int tiles_m = Pm / TMf, tiles_n = (Pn + TNf - 1) / TNf;
  long long chunk = (n + splits - 1) / splits;
  chunk = (chunk + 2 * X2T_TK - 1) / (2 * X2T_TK) * (2 * X2T_TK);
  const int used = (int)((n + chunk - 1) / chunk);
  const int splits_pad = (used + 7) & ~7;
  dim3 grid((unsigned)(tiles_m * tiles_n * splits_pad), (unsigned)batch);
  if (batch > 1 && used != splits) return hipErrorInvalidValue;   // (a batch's slab sets lie `used` slabs apart: ask for what
                                                                  //  gemmx2_tn_batch_splits returns)
  if (used < splits) {   // slabs of unused splits must still be defined for the reduce kernels
    hipError_t e = hipMemsetAsync(slab + (size_t)used * Pm * Pn, 0, (size_t)(splits - used) * Pm * Pn * 4, s);
    if (e != hipSuccess) return e;
    if (bslab) {
      e = hipMemsetAsync(bslab + (size_t)used * Pm, 0, (size_t)(splits - used) * Pm * 4, s);
      if (e != hipSuccess) return e;
    }
  }
#define X2_TN_ARGS s, grid, G, ldg, Z, ldz, n, Pm, Pn, tiles_n, used, chunk, slab, bslab, tiles_m * tiles_n, amax_g, amax_z, z_pre_inv, z_act, g_step, z_step, amax_step
  switch (shp) {
    case 42: return launch_x2_tn_t<4, 2>(X2_TN_ARGS);
    case 61: return launch_x2_tn_t<6, 1>(X2_TN_ARGS);
    case 381: return launch_x2_tn_t<8, 1, 3>(X2_TN_ARGS);
    case 71: return launch_x2_tn_t<7, 1>(X2_TN_ARGS);
    default: return hipErrorInvalidValue;
  }
#undef X2_TN_ARGS
}
