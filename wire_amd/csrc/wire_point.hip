// wire_point.hip -- the bandwidth/VALU-bound kernels around the MFMA GEMMs:
// weight packing, first layer (D <= 4 inputs), final linear (O <= 8 outputs)
// forward / backward, deterministic slab reductions, layout conversion and the
// training-step glue (coordinate generation, MSE gradient, flat Adam).
//
// All of them move each activation byte at most once and keep 128-byte
// (32 lanes x 4 B) or 16-byte-per-lane accesses; none needs MFMA.
#include "wire_dev.h"
#include "wire_point.h"

#define MAXO 8

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// ===========================================================================
// packing
// ===========================================================================
// Hidden layer.  GEMM column j (output) and reduction index k (input) are in
// blocked-planar order.  Complex layer (modules/wire.py:89, F.linear without
// conjugation):  lin = z W^T  <=>  real image
//     [ (o,re),(i,re) ] =  W_re   [ (o,re),(i,im) ] = -W_im
//     [ (o,im),(i,re) ] =  W_im   [ (o,im),(i,im) ] =  W_re
// The data-gradient GEMM g_z = g_lin conj(W) uses exactly the transposed image.
WIRE_DEVINL void pack_hidden_body(int kind, const float* __restrict__ W, const float* __restrict__ b,
                                  const float* __restrict__ V, const float* __restrict__ c, int K,
                                  int Kin, int P, int Pin, int Nc, float* __restrict__ Bt_fwd,
                                  float* __restrict__ Bt_dgrad, float* __restrict__ bias) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;   // input (reduction) index
  const int j = blockIdx.y;                              // GEMM output column
  if (k >= Pin) return;
  float val = 0.f, bv = 0.f;
  if (kind == NK_WIRE || kind == NK_WIRE2D) {
    int o, part;
    const float* Wm = W;
    const float* bm = b;
    if (kind == NK_WIRE) {
      blk_decode(j, o, part);
    } else {
      const int sub = (j >> 5) & 3;
      o = ((j >> 7) << 5) + (j & 31);
      part = sub & 1;
      if (sub >= 2) { Wm = V; bm = c; }
    }
    int i, ipart;
    blk_decode(k, i, ipart);
    if (o < K) {
      bv = bm[2 * o + part];
      if (i < Kin) {
        const float wr = Wm[((size_t)o * Kin + i) * 2];
        const float wi = Wm[((size_t)o * Kin + i) * 2 + 1];
        val = part == 0 ? (ipart == 0 ? wr : -wi) : (ipart == 0 ? wi : wr);
      }
    }
  } else {
    if (j < K) {
      bv = b[j];
      if (k < Kin) val = W[(size_t)j * Kin + k];
    }
  }
  Bt_fwd[(size_t)j * Pin + k] = val;
  Bt_dgrad[(size_t)k * Nc + j] = val;
  if (k == 0) bias[j] = bv;
}
__global__ void pack_hidden_kernel(int kind, const float* __restrict__ W, const float* __restrict__ b,
                                   const float* __restrict__ V, const float* __restrict__ c, int K,
                                   int Kin, int P, int Pin, int Nc, float* __restrict__ Bt_fwd,
                                   float* __restrict__ Bt_dgrad, float* __restrict__ bias) {
  pack_hidden_body(kind, W, b, V, c, K, Kin, P, Pin, Nc, Bt_fwd, Bt_dgrad, bias);
}
// the same for up to PACK_MAXB layers of one shape in one launch (blockIdx.z = layer): wire_pack_params runs once per
// optimizer step, and a chain of ~5 us launches per layer costs more in launch gaps than in work
__global__ void pack_hidden_batch_kernel(int kind, PackBatch pb, int K, int Kin, int P, int Pin, int Nc) {
  const int z = blockIdx.z;
  pack_hidden_body(kind, pb.W[z], pb.b[z], pb.V[z], pb.c[z], K, Kin, P, Pin, Nc, pb.fwd[z], pb.dg[z], pb.bias[z]);
}
hipError_t launch_pack_hidden_batch(hipStream_t s, int kind, const PackBatch& pb, int nb, int K, int Kin, int P,
                                    int Pin) {
  if (nb < 1 || nb > PACK_MAXB) return hipErrorInvalidValue;
  const int Nc = (kind == NK_WIRE2D) ? 2 * P : P;
  dim3 grid(cdiv(Pin, 128), (unsigned)Nc, (unsigned)nb);
  hipLaunchKernelGGL(pack_hidden_batch_kernel, grid, dim3(128), 0, s, kind, pb, K, Kin, P, Pin, Nc);
  return hipGetLastError();
}

hipError_t launch_pack_hidden(hipStream_t s, int kind, const float* W, const float* b,
                              const float* V, const float* c, int K, int Kin, int P, int Pin,
                              float* Bt_fwd, float* Bt_dgrad, float* bias) {
  const int Nc = (kind == NK_WIRE2D) ? 2 * P : P;
  dim3 grid(cdiv(Pin, 128), (unsigned)Nc);
  hipLaunchKernelGGL(pack_hidden_kernel, grid, dim3(128), 0, s, kind, W, b, V, c, K, Kin, P, Pin,
                     Nc, Bt_fwd, Bt_dgrad, bias);
  return hipGetLastError();
}

// final nn.Linear(K, O, cfloat) + .real (modules/wire.py:156-157,164-165):
//   y = z_re W_re^T - z_im W_im^T + Re b
__global__ void pack_final_kernel(int kind, const float* __restrict__ Wf, const float* __restrict__ bf,
                                  int K, int P, int O, float* __restrict__ wf, float* __restrict__ bfr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (c >= P) return;
  float val = 0.f;
  if (kind == NK_WIRE || kind == NK_WIRE2D) {
    int i, part;
    blk_decode(c, i, part);
    if (i < K) {
      const float w = Wf[((size_t)o * K + i) * 2 + part];
      val = part == 0 ? w : -w;
    }
    if (c == 0) bfr[o] = bf[2 * o];
  } else {
    if (c < K) val = Wf[(size_t)o * K + c];
    if (c == 0) bfr[o] = bf[o];
  }
  wf[(size_t)o * P + c] = val;
}

hipError_t launch_pack_final(hipStream_t s, int kind, const float* Wf, const float* bf, int K,
                             int P, int O, float* wf, float* bfr) {
  dim3 grid(cdiv(P, 128), (unsigned)O);
  hipLaunchKernelGGL(pack_final_kernel, grid, dim3(128), 0, s, kind, Wf, bf, K, P, O, wf, bfr);
  return hipGetLastError();
}

// ===========================================================================
// first layer: coords [n][D] (D <= 4) -> activations.  One thread per output
// feature, 64 rows per block: W0 row in registers, coordinates broadcast.
// ===========================================================================
#define FIRST_ROWS 64
template <int KIND>
__global__ void first_fwd_kernel(const float* __restrict__ coords, long long n, int D,
                                 const float* __restrict__ W0, const float* __restrict__ b0,
                                 const float* __restrict__ V0, const float* __restrict__ c0, int K,
                                 int P, float omega, float scale, float* __restrict__ lin,
                                 float* __restrict__ out, unsigned* __restrict__ amax_out) {
  constexpr bool cplx = (KIND == NK_WIRE || KIND == NK_WIRE2D);
  const int nfeat = cplx ? (P >> 1) : P;
  const int f = blockIdx.y * blockDim.x + threadIdx.x;
  const bool live = f < nfeat;                  // (no early return: the maximum below is a whole-wave reduction)
  const bool valid = f < K;
  float amx = 0.f;
  float w[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
  float bb = 0.f, bv = 0.f;
  if (valid) {
    bb = b0[f];
    for (int d = 0; d < D; ++d) w[d] = W0[f * D + d];
    if (KIND == NK_WIRE2D) {
      bv = c0[f];
      for (int d = 0; d < D; ++d) wv[d] = V0[f * D + d];
    }
  }
  const long long r0 = (long long)blockIdx.x * FIRST_ROWS;
  long long r1 = r0 + FIRST_ROWS;
  if (r1 > n) r1 = n;
  const int c_re = cplx ? blk_col(f, 0) : f;
  for (long long row = r0; row < r1; ++row) {
    float u = bb, p = bv;
    for (int d = 0; d < D; ++d) {
      const float x = coords[row * D + d];
      u = __builtin_fmaf(x, w[d], u);
      if (KIND == NK_WIRE2D) p = __builtin_fmaf(x, wv[d], p);
    }
    if (KIND == NK_WIRE) {
      float o_re, o_im;
      gabor_fwd_real(u, omega, scale, o_re, o_im);
      o_re = valid ? o_re : 0.f; o_im = valid ? o_im : 0.f;
      amx = __builtin_fmaxf(amx, __builtin_fmaxf(__builtin_fabsf(o_re), __builtin_fabsf(o_im)));
      if (live) {
        out[row * P + c_re] = o_re;
        out[row * P + c_re + 32] = o_im;
        if (lin) lin[row * nfeat + f] = valid ? u : 0.f;      // per-layer API only: real u, [n][P / 2]
      }
    } else if (KIND == NK_WIRE2D) {
      float o_re, o_im;
      gabor2d_fwd(u, 0.f, p, 0.f, omega, scale, o_re, o_im);
      o_re = valid ? o_re : 0.f; o_im = valid ? o_im : 0.f;
      amx = __builtin_fmaxf(amx, __builtin_fmaxf(__builtin_fabsf(o_re), __builtin_fabsf(o_im)));
      if (live) {
        out[row * P + c_re] = o_re;
        out[row * P + c_re + 32] = o_im;
        if (lin) {                                            // per-layer API only: real (u | p), [n][2 * P / 2]
          lin[row * (2 * nfeat) + f] = valid ? u : 0.f;
          lin[row * (2 * nfeat) + nfeat + f] = valid ? p : 0.f;
        }
      }
    } else {
      constexpr int ACT = KIND - NK_SIREN;
      float o = real_act_fwd<ACT < 0 ? 0 : ACT>(u, omega, scale);
      o = valid ? o : 0.f;
      amx = __builtin_fmaxf(amx, __builtin_fabsf(o));
      if (live) {
        if (lin) lin[row * P + f] = valid ? u : 0.f;
        out[row * P + f] = o;
      }
    }
  }
  // max |out_0| for the 2 x fp16 split GEMM that reads it (wire_gemmx2h.hip)
  if (amax_out) wire_amax_publish(amax_out, amx, threadIdx.x & 63);
}

hipError_t launch_first_fwd(hipStream_t s, int kind, const float* coords, int64_t n, int D,
                            const float* W0, const float* b0, const float* V0, const float* c0,
                            int K, int P, float omega, float scale, float* lin, float* out, unsigned* amax_out) {
  if (n <= 0) return hipSuccess;
  if (D > 4) return hipErrorInvalidValue;
  const bool cplx = (kind == NK_WIRE || kind == NK_WIRE2D);
  const int nfeat = cplx ? P / 2 : P;
  const int bx = nfeat >= 256 ? 256 : ((nfeat + 63) / 64) * 64;
  dim3 grid(cdiv(n, FIRST_ROWS), cdiv(nfeat, bx));
#define FIRST_LAUNCH(KK)                                                                       \
  hipLaunchKernelGGL(first_fwd_kernel<KK>, grid, dim3(bx), 0, s, coords, (long long)n, D, W0, b0, \
                     V0, c0, K, P, omega, scale, lin, out, amax_out)
  switch (kind) {
    case NK_WIRE: FIRST_LAUNCH(NK_WIRE); break;
    case NK_WIRE2D: FIRST_LAUNCH(NK_WIRE2D); break;
    case NK_SIREN: FIRST_LAUNCH(NK_SIREN); break;
    case NK_GAUSS: FIRST_LAUNCH(NK_GAUSS); break;
    case NK_RELU: FIRST_LAUNCH(NK_RELU); break;
    default: return hipErrorInvalidValue;
  }
#undef FIRST_LAUNCH
  return hipGetLastError();
}

// ===========================================================================
// final linear forward: one wave per row, wf staged in LDS, butterfly reduce.
// ===========================================================================
__global__ __launch_bounds__(256) void final_fwd_kernel(const float* __restrict__ z, long long n,
                                                        int P, int O, const float* __restrict__ wf,
                                                        const float* __restrict__ bfr,
                                                        float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float swf[];   // [O][P]
  for (int i = threadIdx.x; i < O * P; i += blockDim.x) swf[i] = wf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long wstride = (long long)gridDim.x * 4;
  for (long long row = (long long)blockIdx.x * 4 + wave; row < n; row += wstride) {
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.f;
    const float* zr = z + row * P;
    for (int c = lane * 4; c < P; c += 256) {
      const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + c);
#pragma unroll
      for (int o = 0; o < MAXO; ++o)
        if (o < O) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&swf[o * P + c]);
          acc[o] += zv[0] * wv[0] + zv[1] * wv[1] + zv[2] * wv[2] + zv[3] * wv[3];
        }
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
      if (o < O) {
        float v = acc[o];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
        if (lane == o) y[row * O + o] = v + bfr[o];
      }
  }
}

hipError_t launch_final_fwd(hipStream_t s, const float* z, int64_t n, int P, int O,
                            const float* wf, const float* bfr, float* y) {
  if (n <= 0) return hipSuccess;
  if (O > MAXO || (P & 3)) return hipErrorInvalidValue;
  unsigned grid = cdiv(n, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(final_fwd_kernel, dim3(grid), dim3(256), (size_t)O * P * sizeof(float), s, z,
                     (long long)n, P, O, wf, bfr, y);
  return hipGetLastError();
}

// ===========================================================================
// final linear backward + last hidden activation gradient.
// ===========================================================================
#define FB_ROWS 256
int final_bwd_blocks(int64_t n) { return (int)((n + FB_ROWS - 1) / FB_ROWS); }

template <int KIND, bool RAW>
__global__ __launch_bounds__(256) void final_bwd_kernel(
    const float* __restrict__ g_y, long long n, int O, const float* __restrict__ wf,
    const float* __restrict__ lin, const float* __restrict__ out, int K, int P, float omega,
    float scale, float* __restrict__ g_lin, float* __restrict__ part_w,
    float* __restrict__ part_b, unsigned* __restrict__ amax_g) {
  constexpr bool cplx = (KIND == NK_WIRE || KIND == NK_WIRE2D);
  __shared__ float sgy[FB_ROWS * MAXO];
  float amx = 0.f;                                        // max |g_lin| for the 2 x fp16 split GEMMs that read it
  const long long r0 = (long long)blockIdx.x * FB_ROWS;
  long long r1 = r0 + FB_ROWS;
  if (r1 > n) r1 = n;
  const int nr = (int)(r1 - r0);
  for (int i = threadIdx.x; i < nr * O; i += blockDim.x) sgy[i] = g_y[r0 * O + i];
  __syncthreads();

  const int nfeat = cplx ? (P >> 1) : P;
  const int f = blockIdx.y * blockDim.x + threadIdx.x;
  if (f < nfeat) {
    const int c0 = cplx ? blk_col(f, 0) : f;
    float w0[MAXO], w1[MAXO], a0[MAXO], a1[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
      w0[o] = 0.f; w1[o] = 0.f; a0[o] = 0.f; a1[o] = 0.f;
      if (o < O) {
        w0[o] = wf[(size_t)o * P + c0];
        if (cplx) w1[o] = wf[(size_t)o * P + c0 + 32];
      }
    }
    const float m2s2 = -2.f * scale * scale;
    const int Pl = (KIND == NK_WIRE2D) ? 2 * P : P;
    const int lc = (KIND == NK_WIRE2D) ? (((f >> 5) << 7) + (f & 31)) : c0;
    // rows in batches of 8: every load of a batch is issued before its first use (a per-row load -> use -> store
    // chain left this kernel at 2 TB/s; profiles/r02_siren_kernel_stats.csv)
    constexpr int RB = 8;
    for (int rb = 0; rb < nr; rb += RB) {
      float pr[RB], pi[RB], l0[RB], l1[RB], l2[RB], l3[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = rb + q < nr ? rb + q : nr - 1;       // clamped: tail rows re-read the last one
        const long long row = r0 + r;
        pr[q] = out[row * P + c0];
        pi[q] = cplx ? out[row * P + c0 + 32] : 0.f;
        l0[q] = l1[q] = l2[q] = l3[q] = 0.f;
        if (!RAW) {
          if (KIND == NK_WIRE2D) {
            const float* L = lin + row * Pl + lc;
            l0[q] = L[0]; l1[q] = L[32]; l2[q] = L[64]; l3[q] = L[96];
          } else if (KIND != NK_RELU) {             // relu: lin is not stored (lin > 0 <=> out > 0)
            l0[q] = lin[row * P + c0];
            if (KIND == NK_WIRE) l1[q] = lin[row * P + c0 + 32];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int r = rb + q;
        if (r < nr) {
          const long long row = r0 + r;
          float gr = 0.f, gi = 0.f;
#pragma unroll
          for (int o = 0; o < MAXO; ++o)
            if (o < O) {
              const float g = sgy[r * O + o];
              gr = __builtin_fmaf(g, w0[o], gr);
              if (cplx) gi = __builtin_fmaf(g, w1[o], gi);
              a0[o] = __builtin_fmaf(g, pr[q], a0[o]);
              if (cplx) a1[o] = __builtin_fmaf(g, pi[q], a1[o]);
            }
          if (RAW) {
            g_lin[row * P + c0] = gr;
            if (cplx) g_lin[row * P + c0 + 32] = gi;
          } else if (KIND == NK_WIRE) {
            float gl_re, gl_im;
            gabor_bwd(gr, gi, l0[q], l1[q], pr[q], pi[q], omega, m2s2, gl_re, gl_im);
            g_lin[row * P + c0] = gl_re;
            g_lin[row * P + c0 + 32] = gl_im;
            amx = __builtin_fmaxf(amx, __builtin_fmaxf(__builtin_fabsf(gl_re), __builtin_fabsf(gl_im)));
          } else if (KIND == NK_WIRE2D) {
            const float c_r = __builtin_fmaf(pr[q], gr, pi[q] * gi);
            const float c_i = __builtin_fmaf(pr[q], gi, -(pi[q] * gr));
            const float t = m2s2 * c_r;
            float* Gp = g_lin + row * Pl + lc;
            const float g0 = __builtin_fmaf(t, l0[q], omega * c_i), g1 = __builtin_fmaf(t, l1[q], -(omega * c_r));
            const float g2 = t * l2[q], g3 = t * l3[q];
            Gp[0] = g0; Gp[32] = g1; Gp[64] = g2; Gp[96] = g3;
            amx = __builtin_fmaxf(amx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1)),
                                                       __builtin_fmaxf(__builtin_fabsf(g2), __builtin_fabsf(g3))));
          } else {
            constexpr int ACT = (KIND - NK_SIREN) < 0 ? 0 : (KIND - NK_SIREN);
            const float gl = real_act_bwd<ACT>(gr, l0[q], pr[q], omega, scale);
            g_lin[row * P + c0] = gl;
            amx = __builtin_fmaxf(amx, __builtin_fabsf(gl));
          }
        }
      }
    }
    float* pw = part_w + (size_t)blockIdx.x * O * P;
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
      if (o < O) {
        pw[(size_t)o * P + c0] = a0[o];
        if (cplx) pw[(size_t)o * P + c0 + 32] = a1[o];
      }
  }
  // bias partial: thread o < O of the first feature block sums g_y[:, o]
  if (blockIdx.y == 0 && threadIdx.x < O) {
    float sacc = 0.f;
    for (int r = 0; r < nr; ++r) sacc += sgy[r * O + threadIdx.x];
    part_b[(size_t)blockIdx.x * O + threadIdx.x] = sacc;
  }
  if (amax_g) wire_amax_publish(amax_g, amx, threadIdx.x & 63);
}

hipError_t launch_final_bwd(hipStream_t s, int kind, int raw, const float* g_y, int64_t n, int O,
                            const float* wf, const float* lin, const float* out, int K, int P,
                            float omega, float scale, float* g_lin, float* part_w,
                            float* part_b, unsigned* amax_g) {
  if (n <= 0) return hipSuccess;
  if (O > MAXO) return hipErrorInvalidValue;
  const bool cplx = (kind == NK_WIRE || kind == NK_WIRE2D);
  const int nfeat = cplx ? P / 2 : P;
  dim3 grid((unsigned)final_bwd_blocks(n), cdiv(nfeat, 256));
#define FB_LAUNCH(KK, RR)                                                                        \
  hipLaunchKernelGGL((final_bwd_kernel<KK, RR>), grid, dim3(256), 0, s, g_y, (long long)n, O, wf, \
                     lin, out, K, P, omega, scale, g_lin, part_w, part_b, amax_g)
  if (raw) {
    if (cplx) FB_LAUNCH(NK_WIRE, true); else FB_LAUNCH(NK_RELU, true);
  } else {
    switch (kind) {
      case NK_WIRE: FB_LAUNCH(NK_WIRE, false); break;
      case NK_WIRE2D: FB_LAUNCH(NK_WIRE2D, false); break;
      case NK_SIREN: FB_LAUNCH(NK_SIREN, false); break;
      case NK_GAUSS: FB_LAUNCH(NK_GAUSS, false); break;
      case NK_RELU: FB_LAUNCH(NK_RELU, false); break;
      default: return hipErrorInvalidValue;
    }
  }
#undef FB_LAUNCH
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fused final stage (training path, wire): one wave per row.
// lane l owns features f = 32 g + 4 (l & 7) + j, j = 0..3, of group g = (l >> 3) + 8 pass: its row
// slice is two float4 (re, im).  y_o is a butterfly over the wave; everything after it (loss term,
// dL/dy, g_out, Gabor gradient, g_Wf partials) stays in registers.
// ---------------------------------------------------------------------------
#define FF_MAXO 4
#define FF_MAXPASS 2      // P <= 1024 floats per row
#define FF_MAXROWS 512    // rows of one block (FB_ROWS <= rows <= FF_MAXROWS, chosen by the launcher)
__global__ void mse_final_kernel(const float* __restrict__ partial, int nb, float lscale,
                                 float* __restrict__ loss_out);
// (the kernel keeps 4 O P floats of W_f / partial sums in dynamic LDS: it must fit the 64 KB a launch gets
// without an opt-in -- O = 4 with P = 1024 does not, and runs the unfused sequence instead)
// dynamic LDS of the fused final stage: [4 waves][O][P] g_wf partials, [4][FF_MAXO + 1] bias / loss partials, then the
// block's gathered targets [FF_MAXROWS][FF_MAXO] and source indices [FF_MAXROWS] (int64)
static size_t final_fused_shm(int P, int O) {
  return ((size_t)4 * O * P + 4 * (FF_MAXO + 1) + (size_t)FF_MAXROWS * FF_MAXO) * sizeof(float) + (size_t)FF_MAXROWS * 8;
}
bool final_fused_supported(int P, int O) {
  return (P % 64) == 0 && P <= 512 * FF_MAXPASS && O <= FF_MAXO && final_fused_shm(P, O) <= 65536;
}

// KIND: the activation whose gradient is fused (wire / wire2d: complex pairs (re | im) 32 columns apart; siren / gauss /
//   relu: the two 4-column chunks of a lane are just 8 real features).  wire2d: lin is the 2P-wide (lin | sy) row.
// RECOMP (wire only): out_L is not read -- it is evaluated again from lin_L with the lean forward form (bit-identical
//   to what the 16 x 16 x 32 forward epilogue would have stored; wire_api.hip selects this only when that kernel ran
//   layer L, and then does not let it write out_L at all): 1 GB instead of 1.5 GB of HBM traffic for this pass.
// OT >= O (1..FF_MAXO): the weight / partial-sum registers are sized for the actual number of outputs
// RPW: rows per wave slot -- 2 when a row has at most 256 floats (siren / gauss / relu and wire2d at 256 features):
// lanes 0-31 take one row, lanes 32-63 the next, instead of leaving half the wave idle
WIRE_DEVINL void ff_amax4(float& m, const f32x4& v) {
  m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])),
                                         __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
}
template <int NPASS, int KIND, bool RECOMP, int OT, int RPW>
__global__ __launch_bounds__(256) void final_fused_kernel(
    const float* __restrict__ out, const float* __restrict__ lin, long long n, int P, int O, int kvalid,
    const float* __restrict__ wf, const float* __restrict__ bfr, const float* __restrict__ target,
    const int64_t* __restrict__ idx, long long first, float gscale, float omega, float scale,
    float* __restrict__ y, float* __restrict__ rec, float* __restrict__ g_lin, float* __restrict__ part_w,
    float* __restrict__ part_b, float* __restrict__ loss_partial, int rows_pb, unsigned* __restrict__ amax_g) {
  static_assert(!RECOMP || KIND != NK_RELU, "relu keeps out and has no lin to recompute it from");
  float amx = 0.f;                                     // max |g_lin| for the 2 x fp16 split GEMMs that read it
  static_assert(RPW == 1 || (RPW == 2 && NPASS == 1), "two rows per wave slot: one pass of at most 256 columns");
  constexpr int RL = 64 / RPW;                         // lanes of one row
  constexpr bool HAS_LIN = (KIND != NK_RELU);          // relu: lin is never stored (lin > 0 <=> out > 0)
  constexpr int NL = (KIND == NK_WIRE2D) ? 4 : 2;      // 4-column chunks of lin per lane and pass
  extern __shared__ float sm[];                        // [4 waves][O][P] g_wf partials + [4][O+1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / RL, lr_ = lane & (RL - 1);    // row of the slot, lane inside the row
  const int ngrp = P >> 6;
  const int Pl = (KIND == NK_WIRE2D) ? 2 * P : P;      // row stride of lin / g_lin
  const long long r0 = (long long)blockIdx.x * rows_pb;
  long long r1 = r0 + rows_pb;
  if (r1 > n) r1 = n;
  const float m2s2 = -2.f * scale * scale;

  // this lane's columns and weights
  int col[NPASS], lcol[NPASS];
  bool live[NPASS];
  f32x4 wre[NPASS][OT], wim[NPASS][OT];
  f32x4 are[NPASS][OT], aim[NPASS][OT];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int g = (lr_ >> 3) + 8 * ps;
    live[ps] = g < ngrp;
    col[ps] = (g << 6) + 4 * (lr_ & 7);
    lcol[ps] = (KIND == NK_WIRE2D) ? (g << 7) + 4 * (lr_ & 7) : col[ps];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      wre[ps][o] = z4; wim[ps][o] = z4; are[ps][o] = z4; aim[ps][o] = z4;
      if (live[ps] && o < O) {
        wre[ps][o] = *reinterpret_cast<const f32x4*>(wf + (size_t)o * P + col[ps]);
        wim[ps][o] = *reinterpret_cast<const f32x4*>(wf + (size_t)o * P + col[ps] + 32);
      }
    }
  }
  float bsum[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) bsum[o] = 0.f;
  float lsum = 0.f;

  // the block's source indices and targets, gathered once by all 256 threads: inside the row loop the chain
  // idx[row] -> target[src] would be two dependent global loads per row and wave (it bounded the pass)
  float* s_tg = sm + (size_t)4 * O * P + 4 * (FF_MAXO + 1);                       // [FF_MAXROWS][FF_MAXO]
  long long* s_src = reinterpret_cast<long long*>(s_tg + FF_MAXROWS * FF_MAXO);   // [FF_MAXROWS]
  for (int t = threadIdx.x; t < rows_pb; t += 256) {
    const long long grow = r0 + t;
    if (grow < r1) {
      const long long src = idx ? idx[grow] : first + grow;
      s_src[t] = src;
#pragma unroll
      for (int o = 0; o < OT; ++o)
        if (o < O) s_tg[t * FF_MAXO + o] = target[src * O + o];
    }
  }
  __syncthreads();

  // software pipeline over this wave's rows: the loads of the next PF rows of this wave (r + 4, r + 8, ...) are in
  // flight while row r goes through its butterfly / gradient chain -- the pass is bound by memory latency, and one
  // row is only 4 KB (2 KB when out is recomputed) per wave in flight
  constexpr int PF = RECOMP ? 3 : 2;
  f32x4 nzr[PF][NPASS], nzi[PF][NPASS], nl[PF][NPASS][NL];
  auto load_row = [&](const int slot, long long row) {
    row += sub;                                          // this lane's row of the slot
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      nzr[slot][ps] = z4; nzi[slot][ps] = z4;
#pragma unroll
      for (int c = 0; c < NL; ++c) nl[slot][ps][c] = z4;
      if (live[ps] && row < r1) {
        if (!RECOMP) {
          const size_t o_ = (size_t)row * P + col[ps];
          nzr[slot][ps] = *reinterpret_cast<const f32x4*>(out + o_);
          nzi[slot][ps] = *reinterpret_cast<const f32x4*>(out + o_ + 32);
        }
        if (HAS_LIN) {
          const float* Lp = lin + (size_t)row * Pl + lcol[ps];
#pragma unroll
          for (int c = 0; c < NL; ++c) nl[slot][ps][c] = *reinterpret_cast<const f32x4*>(Lp + 32 * c);
        }
      }
    }
  };
  auto process = [&](const long long row_slot, f32x4 (&zr)[NPASS], f32x4 (&zi)[NPASS],
                     const f32x4 (&ll)[NPASS][NL]) {
    const long long row = row_slot + sub;                // this lane's row; past the end: contributes nothing
    const bool rl = row < r1;
    if (RECOMP && (KIND == NK_WIRE || KIND == NK_WIRE2D)) {
      const float w0l2e = omega * 1.44269502f, ns2l2e = -(scale * scale) * 1.44269502f;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int f0 = ((col[ps] >> 6) << 5) + (col[ps] & 31);     // features f0 .. f0 + 3 of this lane
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a_, b_;
          if (KIND == NK_WIRE2D)
            gabor2d_fwd_lean(ll[ps][0][j], ll[ps][1][j], ll[ps][NL - 2][j], ll[ps][NL - 1][j], omega, scale, a_, b_);
          else
            gabor_fwd_lean(ll[ps][0][j], ll[ps][1][j], omega, w0l2e, ns2l2e, a_, b_);
          const bool valid = live[ps] && f0 + j < kvalid;          // pad features are 0
          zr[ps][j] = valid ? a_ : 0.f;
          zi[ps][j] = valid ? b_ : 0.f;
        }
      }
    } else if (RECOMP) {                                           // siren / gauss: 8 real features per lane and pass
      constexpr int ACT = (KIND - NK_SIREN) < 0 ? 0 : (KIND - NK_SIREN);
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a_ = real_act_fwd_lean<ACT>(ll[ps][0][j], omega, scale);
          const float b_ = real_act_fwd_lean<ACT>(ll[ps][1][j], omega, scale);
          zr[ps][j] = (live[ps] && col[ps] + j < kvalid) ? a_ : 0.f;
          zi[ps][j] = (live[ps] && col[ps] + 32 + j < kvalid) ? b_ : 0.f;
        }
    }
    float yo[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc = __builtin_fmaf(zr[ps][j], wre[ps][o][j], __builtin_fmaf(zi[ps][j], wim[ps][o][j], acc));
      yo[o] = acc;
    }
#pragma unroll
    for (int o = 0; o < OT; ++o)
      if (o < O) {
#pragma unroll
        for (int sft = RL / 2; sft >= 1; sft >>= 1) yo[o] += __shfl_xor(yo[o], sft);
      }
    const int lrow = rl ? (int)(row - r0) : 0;
    const long long src = s_src[lrow];
    float gy[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      gy[o] = 0.f;
      if (o < O) {
        const float yy = yo[o] + bfr[o];
        const float dlt = rl ? yy - s_tg[lrow * FF_MAXO + o] : 0.f;
        gy[o] = gscale * dlt;
        if (lr_ == o && rl) {
          y[row * O + o] = yy;
          if (rec) rec[src * O + o] = yy;
        }
        lsum = __builtin_fmaf(dlt, dlt, lsum);
        bsum[o] += gy[o];
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!live[ps] || !rl) continue;
      f32x4 gr = {0.f, 0.f, 0.f, 0.f}, gi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < OT; ++o)
        if (o < O) {
          gr += gy[o] * wre[ps][o];
          gi += gy[o] * wim[ps][o];
          are[ps][o] += gy[o] * zr[ps];
          aim[ps][o] += gy[o] * zi[ps];
        }
      float* Gp = g_lin + (size_t)row * Pl + lcol[ps];
      if (KIND == NK_WIRE) {
        f32x4 glr, gli;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a_, b_;
          gabor_bwd(gr[j], gi[j], ll[ps][0][j], ll[ps][1][j], zr[ps][j], zi[ps][j], omega, m2s2, a_, b_);
          glr[j] = a_; gli[j] = b_;
        }
        *reinterpret_cast<f32x4*>(Gp) = glr;
        *reinterpret_cast<f32x4*>(Gp + 32) = gli;
        ff_amax4(amx, glr); ff_amax4(amx, gli);
      } else if (KIND == NK_WIRE2D) {
        // c = conj(out) g;  g_lin = -2 s^2 Re(c) lin - j w0 c;  g_sy = -2 s^2 Re(c) sy   (modules/wire2d.py:56-67)
        f32x4 g0, g1, g2, g3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c_r = __builtin_fmaf(zr[ps][j], gr[j], zi[ps][j] * gi[j]);
          const float c_i = __builtin_fmaf(zr[ps][j], gi[j], -(zi[ps][j] * gr[j]));
          const float t = m2s2 * c_r;
          g0[j] = __builtin_fmaf(t, ll[ps][0][j], omega * c_i);
          g1[j] = __builtin_fmaf(t, ll[ps][1][j], -(omega * c_r));
          g2[j] = t * ll[ps][NL - 2][j];
          g3[j] = t * ll[ps][NL - 1][j];
        }
        *reinterpret_cast<f32x4*>(Gp) = g0;
        *reinterpret_cast<f32x4*>(Gp + 32) = g1;
        *reinterpret_cast<f32x4*>(Gp + 64) = g2;
        *reinterpret_cast<f32x4*>(Gp + 96) = g3;
        ff_amax4(amx, g0); ff_amax4(amx, g1); ff_amax4(amx, g2); ff_amax4(amx, g3);
      } else {
        // the same forms final_bwd_kernel uses (sin'= w0 cos, gauss' = -2 s^2 lin out, relu' = [out > 0])
        constexpr int ACT = (KIND - NK_SIREN) < 0 ? 0 : (KIND - NK_SIREN);
        f32x4 glr, gli;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          glr[j] = real_act_bwd<ACT>(gr[j], ll[ps][0][j], zr[ps][j], omega, scale);
          gli[j] = real_act_bwd<ACT>(gi[j], ll[ps][1][j], zi[ps][j], omega, scale);
        }
        *reinterpret_cast<f32x4*>(Gp) = glr;
        *reinterpret_cast<f32x4*>(Gp + 32) = gli;
        ff_amax4(amx, glr); ff_amax4(amx, gli);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) load_row(d, r0 + (wave + 4 * d) * RPW);
  for (long long row = r0 + wave * RPW; row < r1; row += 4 * RPW * PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const long long rr_ = row + 4 * RPW * d;
      if (rr_ < r1) {                                   // wave-uniform
        f32x4 zr[NPASS], zi[NPASS], ll[NPASS][NL];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          zr[ps] = nzr[d][ps]; zi[ps] = nzi[d][ps];
#pragma unroll
          for (int c = 0; c < NL; ++c) ll[ps][c] = nl[d][ps][c];
        }
        load_row(d, rr_ + 4 * RPW * PF);
        process(rr_, zr, zi, ll);
      }
    }
  }

  // ---- combine the 4 waves: g_wf partials [O][P], bias partial [O], loss
  float* swf = sm;                                   // [4][O][P]
  float* sb = sm + 4 * O * P;                        // [4][FF_MAXO + 1]
  if (RPW == 2) {                                    // the two half waves hold partial sums of the same columns
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          are[ps][o][j] += __shfl_xor(are[ps][o][j], 32);
          aim[ps][o][j] += __shfl_xor(aim[ps][o][j], 32);
        }
      bsum[o] += __shfl_xor(bsum[o], 32);
    }
    lsum += __shfl_xor(lsum, 32);
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps)
    if (live[ps] && sub == 0)
#pragma unroll
      for (int o = 0; o < OT; ++o)
        if (o < O) {
          *reinterpret_cast<f32x4*>(swf + ((size_t)wave * O + o) * P + col[ps]) = are[ps][o];
          *reinterpret_cast<f32x4*>(swf + ((size_t)wave * O + o) * P + col[ps] + 32) = aim[ps][o];
        }
  // lsum is identical in every lane only per row-term count; every lane added the same terms -> use lane 0
  if (lane == 0) {
#pragma unroll
    for (int o = 0; o < OT; ++o) sb[wave * (FF_MAXO + 1) + o] = bsum[o];
    sb[wave * (FF_MAXO + 1) + FF_MAXO] = lsum;
  }
  __syncthreads();
  float* pw = part_w + (size_t)blockIdx.x * O * P;
  for (int e = threadIdx.x; e < O * P; e += 256)
    pw[e] = (swf[e] + swf[O * P + e]) + (swf[2 * O * P + e] + swf[3 * O * P + e]);
  if (threadIdx.x < O)
    part_b[(size_t)blockIdx.x * O + threadIdx.x] =
        (sb[threadIdx.x] + sb[(FF_MAXO + 1) + threadIdx.x]) +
        (sb[2 * (FF_MAXO + 1) + threadIdx.x] + sb[3 * (FF_MAXO + 1) + threadIdx.x]);
  if (threadIdx.x == 0)
    loss_partial[blockIdx.x] = (sb[FF_MAXO] + sb[(FF_MAXO + 1) + FF_MAXO]) +
                               (sb[2 * (FF_MAXO + 1) + FF_MAXO] + sb[3 * (FF_MAXO + 1) + FF_MAXO]);
  if (amax_g) wire_amax_publish(amax_g, amx, lane);
}

// kind: NK_*.  out = nullptr (wire only): out_L is recomputed from lin_L (kvalid = number of valid complex
// features).  lin: [n][P] (wire2d: the 2P-wide (lin | sy) rows; relu: unused).  g_lin has the layout of lin.
hipError_t launch_final_fused(hipStream_t s, int kind, const float* out, const float* lin, int64_t n, int P, int O,
                              int kvalid, const float* wf, const float* bfr, const float* target, const int64_t* idx,
                              int64_t first, float weight, float omega, float scale, float* y, float* rec,
                              float* g_lin, float* part_w, float* part_b, float* loss_partial,
                              float* loss_out, unsigned* amax_g) {
  if (n <= 0) return hipSuccess;
  if (!final_fused_supported(P, O)) return hipErrorInvalidValue;
  if (!out && kind == NK_RELU) return hipErrorInvalidValue;
  // Rows per block: the kernel runs 3 blocks per CU (registers), i.e. 768 at a time; 1024 blocks of FB_ROWS rows at
  // n = 262 144 would be one full round and a second one on a third of the chip.  So a block takes as many rows as
  // it needs for ONE round (342 -> 767 blocks), within the FB_ROWS-block partial-sum layout of launch_final_reduce:
  // fewer blocks than final_bwd_blocks(n), the unused partial slots are zeroed.
  const int nbf = final_bwd_blocks(n);
  int rows_pb = FB_ROWS;
  if (nbf > 768) {
    rows_pb = (int)((n + 767) / 768);
    rows_pb = (rows_pb + 7) & ~7;
    if (rows_pb > FF_MAXROWS) rows_pb = FF_MAXROWS;
  }
  const int nblk = (int)((n + rows_pb - 1) / rows_pb);
  if (nblk < nbf) {
    hipError_t e = hipMemsetAsync(part_w + (size_t)nblk * O * P, 0, (size_t)(nbf - nblk) * O * P * sizeof(float), s);
    if (e == hipSuccess) e = hipMemsetAsync(part_b + (size_t)nblk * O, 0, (size_t)(nbf - nblk) * O * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  const float inv = (float)(1.0 / ((double)n * (double)O));
  const size_t shm = final_fused_shm(P, O);
#define FF_LAUNCH(NP, RW, KD, RC, OT)                                                                        \
  hipLaunchKernelGGL((final_fused_kernel<NP, KD, RC, OT, RW>), dim3((unsigned)nblk), dim3(256), shm, s, out, \
                     lin, (long long)n, P, O, kvalid, wf, bfr, target, idx, (long long)first,                \
                     weight * 2.f * inv, omega, scale, y, rec, g_lin, part_w, part_b, loss_partial, rows_pb, amax_g)
#define FF_LAUNCH_O(NP, RW, KD, RC)                                                                          \
  switch (O) {                                                                                               \
    case 1: FF_LAUNCH(NP, RW, KD, RC, 1); break;                                                             \
    case 2: case 3: FF_LAUNCH(NP, RW, KD, RC, 3); break;                                                     \
    default: FF_LAUNCH(NP, RW, KD, RC, 4); break;                                                            \
  }
#define FF_LAUNCH_K(NP, RW)                                                                                  \
  switch (kind) {                                                                                            \
    case NK_WIRE:                                                                                            \
      if (out) { FF_LAUNCH_O(NP, RW, NK_WIRE, false); } else { FF_LAUNCH_O(NP, RW, NK_WIRE, true); }         \
      break;                                                                                                 \
    case NK_WIRE2D:                                                                                          \
      if (out) { FF_LAUNCH_O(NP, RW, NK_WIRE2D, false); } else { FF_LAUNCH_O(NP, RW, NK_WIRE2D, true); }     \
      break;                                                                                                 \
    case NK_SIREN:                                                                                           \
      if (out) { FF_LAUNCH_O(NP, RW, NK_SIREN, false); } else { FF_LAUNCH_O(NP, RW, NK_SIREN, true); }       \
      break;                                                                                                 \
    case NK_GAUSS:                                                                                           \
      if (out) { FF_LAUNCH_O(NP, RW, NK_GAUSS, false); } else { FF_LAUNCH_O(NP, RW, NK_GAUSS, true); }       \
      break;                                                                                                 \
    case NK_RELU: FF_LAUNCH_O(NP, RW, NK_RELU, false); break;                                                \
    default: return hipErrorInvalidValue;                                                                    \
  }
  if (P <= 256) { FF_LAUNCH_K(1, 2); } else if (P <= 512) { FF_LAUNCH_K(1, 1); } else { FF_LAUNCH_K(2, 1); }
#undef FF_LAUNCH_K
#undef FF_LAUNCH_O
#undef FF_LAUNCH
  // the loss partials are summed by the MSE final kernel (one block)
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, loss_partial, nblk, weight * inv, loss_out);
  return hipGetLastError();
}

// pre-reduction of per-row-block partials: in[nblk][C] -> out[nchunk][C], chunk c sums blocks
// [c*per, (c+1)*per).  Keeps the final reductions short (they were latency-bound over 1024 blocks).
#define PRE_CHUNKS 32
__global__ __launch_bounds__(256) void prereduce_kernel(const float* __restrict__ in, int nblk, int C, int per,
                                                        float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int ch = blockIdx.y;
  if (c >= C) return;
  const int b0 = ch * per;
  int b1 = b0 + per;
  if (b1 > nblk) b1 = nblk;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    a0 += in[(size_t)b * C + c];
    a1 += in[(size_t)(b + 1) * C + c];
    a2 += in[(size_t)(b + 2) * C + c];
    a3 += in[(size_t)(b + 3) * C + c];
  }
  for (; b < b1; ++b) a0 += in[(size_t)b * C + c];
  out[(size_t)ch * C + c] = (a0 + a1) + (a2 + a3);
}
// returns the number of blocks left (nblk itself when no pre-reduction was worth it)
static int prereduce(hipStream_t s, const float* in, int nblk, int C, float* out) {
  if (nblk <= 2 * PRE_CHUNKS) return nblk;
  const int per = (nblk + PRE_CHUNKS - 1) / PRE_CHUNKS;
  const int nch = (nblk + per - 1) / per;
  hipLaunchKernelGGL(prereduce_kernel, dim3(cdiv(C, 256), (unsigned)nch), dim3(256), 0, s, in, nblk, C, per, out);
  return nch;
}

// g_Wf = g_y^T conj(z):  re = sum g z_re, im = -sum g z_im;  g_bf = sum g + 0j
// block = 64 columns x 4 partial groups (each group strides over the row blocks), LDS combine
__global__ __launch_bounds__(256) void final_reduce_kernel(int kind, const float* __restrict__ part_w,
                                    const float* __restrict__ part_b, int nblk, int O, int K, int P,
                                    float* __restrict__ gWf, float* __restrict__ gbf) {
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const int o = blockIdx.y;
  const bool cplx = (kind == NK_WIRE || kind == NK_WIRE2D);
  float sr = 0.f, si = 0.f;
  if (i < K) {
    const int c = cplx ? blk_col(i, 0) : i;
    for (int b = ty; b < nblk; b += 4) {
      const float* pw = part_w + ((size_t)b * O + o) * P;
      sr += pw[c];
      if (cplx) si += pw[c + 32];
    }
  }
  red[0][ty][tx] = sr;
  red[1][ty][tx] = si;
  __syncthreads();
  if (ty == 0 && i < K) {
    sr = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
    si = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
    if (cplx) {
      gWf[((size_t)o * K + i) * 2] = sr;
      gWf[((size_t)o * K + i) * 2 + 1] = -si;
    } else {
      gWf[(size_t)o * K + i] = sr;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float sb = 0.f;
    for (int b = 0; b < nblk; ++b) sb += part_b[(size_t)b * O + o];
    if (cplx) { gbf[2 * o] = sb; gbf[2 * o + 1] = 0.f; } else gbf[o] = sb;
  }
}

hipError_t launch_final_reduce(hipStream_t s, int kind, float* part_w, float* part_b,
                               int nblk, int O, int K, int P, float* gWf, float* gbf) {
  // the partial buffers have room for nblk + PRE_CHUNKS blocks (final_bwd_blocks adds the slack)
  float* w2 = part_w + (size_t)nblk * O * P;
  float* b2 = part_b + (size_t)nblk * O;
  const int nw = prereduce(s, part_w, nblk, O * P, w2);
  const int nb = prereduce(s, part_b, nblk, O, b2);
  dim3 grid(cdiv(K, 64), (unsigned)O);
  hipLaunchKernelGGL(final_reduce_kernel, grid, dim3(256), 0, s, kind, nw == nblk ? part_w : w2,
                     nb == nblk ? part_b : b2, nw == nblk ? nblk : nw, O, K, P, gWf, gbf);
  return hipGetLastError();
}

// ===========================================================================
// hidden weight-gradient reduction.  M = G^T Z in blocked-planar real form;
//   g_W = g_lin^T conj(z):  re = M[(o,re),(i,re)] + M[(o,im),(i,im)]
//                           im = M[(o,im),(i,re)] - M[(o,re),(i,im)]
// ===========================================================================
// block = 64 input features x 4 groups of row splits (group q sums splits q, q + 4, ... with four loads in
// flight); the four partial sums are combined through LDS in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int kind, const float* __restrict__ slab,
                                    const float* __restrict__ bslab, int S, int K, int Kin, int Pm,
                                    int Pn, float* __restrict__ gW, float* __restrict__ gb,
                                    float* __restrict__ gV, float* __restrict__ gc) {
  __shared__ float red[4][4][64];       // [value][group][feature]
  const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const int o = blockIdx.y;
  const size_t sstride = (size_t)Pm * Pn;
  const bool cplx = (kind == NK_WIRE || kind == NK_WIRE2D);
  const int nmat = (kind == NK_WIRE2D) ? 2 : 1;
  for (int mat = 0; mat < nmat; ++mat) {
    int r_re, r_im = 0;
    if (kind == NK_WIRE) { r_re = blk_col(o, 0); r_im = r_re + 32; }
    else if (kind == NK_WIRE2D) { r_re = ((o >> 5) << 7) + 64 * mat + (o & 31); r_im = r_re + 32; }
    else r_re = o;
    float sr = 0.f, si = 0.f, br = 0.f, bi = 0.f;
    if (i < Kin) {
      if (cplx) {
        const int c_re = blk_col(i, 0), c_im = c_re + 32;
        const size_t a = (size_t)r_re * Pn + c_re, b = (size_t)r_im * Pn + c_im, c = (size_t)r_im * Pn + c_re,
                     d = (size_t)r_re * Pn + c_im;
#pragma unroll 4
        for (int sp = q; sp < S; sp += 4) {
          const float* m = slab + sp * sstride;
          sr += m[a] + m[b];
          si += m[c] - m[d];
        }
      } else {
        const size_t a = (size_t)o * Pn + i;
#pragma unroll 4
        for (int sp = q; sp < S; sp += 4) sr += slab[sp * sstride + a];
      }
    }
    if (tx == 0 && blockIdx.x == 0)
      for (int sp = q; sp < S; sp += 4) {
        br += bslab[(size_t)sp * Pm + r_re];
        if (cplx) bi += bslab[(size_t)sp * Pm + r_im];
      }
    red[0][q][tx] = sr; red[1][q][tx] = si; red[2][q][tx] = br; red[3][q][tx] = bi;
    __syncthreads();
    if (q == 0) {
      float* gWm = mat == 0 ? gW : gV;
      float* gbm = mat == 0 ? gb : gc;
      const float wr = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
      const float wi = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
      if (i < Kin) {
        if (cplx) {
          gWm[((size_t)o * Kin + i) * 2] = wr;
          gWm[((size_t)o * Kin + i) * 2 + 1] = wi;
        } else {
          gWm[(size_t)o * Kin + i] = wr;
        }
      }
      if (tx == 0 && blockIdx.x == 0) {
        const float b0 = (red[2][0][0] + red[2][1][0]) + (red[2][2][0] + red[2][3][0]);
        const float b1 = (red[3][0][0] + red[3][1][0]) + (red[3][2][0] + red[3][3][0]);
        if (cplx) { gbm[2 * o] = b0; gbm[2 * o + 1] = b1; }
        else gbm[o] = b0;
      }
    }
    __syncthreads();
  }
}

hipError_t launch_wgrad_reduce(hipStream_t s, int kind, const float* slab, const float* bslab,
                               int S, int K, int Kin, int Pm, int Pn, float* gW, float* gb,
                               float* gV, float* gc) {
  dim3 grid(cdiv(Kin, 64), (unsigned)K);
  hipLaunchKernelGGL(wgrad_reduce_kernel, grid, dim3(256), 0, s, kind, slab, bslab, S, K, Kin, Pm,
                     Pn, gW, gb, gV, gc);
  return hipGetLastError();
}

// ===========================================================================
// first-layer weight gradient: tall-skinny  G[n][C]^T [x | 1]
//   stage 1: block = 64 column quads (float4) x 4 row lanes over CR_ROWS rows -> partial[blk][C][5]
//   stage 2: block = 64 columns x 4 groups over the row blocks
// ===========================================================================
#define CR_ROWS 256
int colreduce_blocks(int64_t n) { return (int)((n + CR_ROWS - 1) / CR_ROWS); }

__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ G, int ldg, int C,
                                                        const float* __restrict__ x, int D,
                                                        long long n, float* __restrict__ partial) {
  __shared__ float red[4][64][20];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = (blockIdx.y * 64 + tx) * 4;
  const long long r0 = (long long)blockIdx.x * CR_ROWS;
  long long r1 = r0 + CR_ROWS;
  if (r1 > n) r1 = n;
  float acc[4][5];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < 5; ++d) acc[q][d] = 0.f;
  if (c0 < ldg) {
    for (long long row = r0 + ty; row < r1; row += 4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(G + row * ldg + c0);
      float xv[4] = {0.f, 0.f, 0.f, 0.f};
      for (int d = 0; d < D; ++d) xv[d] = x[row * D + d];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[q][d] = __builtin_fmaf(g[q], xv[d], acc[q][d]);
        acc[q][4] += g[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < 5; ++d) red[ty][tx][q * 5 + d] = acc[q][d];
  __syncthreads();
  if (ty == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      if (c < C) {
        float* p = partial + ((size_t)blockIdx.x * C + c) * 5;
#pragma unroll
        for (int d = 0; d < 5; ++d)
          p[d] = (red[0][tx][q * 5 + d] + red[1][tx][q * 5 + d]) + (red[2][tx][q * 5 + d] + red[3][tx][q * 5 + d]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ partial, int nblk,
                                                              int C, int D, float* __restrict__ gW0,
                                                              float* __restrict__ gb0) {
  __shared__ float red[4][64][5];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (int b = ty; b < nblk; b += 4) {
      const float* p = partial + ((size_t)b * C + c) * 5;
#pragma unroll
      for (int d = 0; d < 5; ++d) acc[d] += p[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 5; ++d) red[ty][tx][d] = acc[d];
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int d = 0; d < 5; ++d) acc[d] = (red[0][tx][d] + red[1][tx][d]) + (red[2][tx][d] + red[3][tx][d]);
    for (int d = 0; d < D; ++d) gW0[c * D + d] = acc[d];
    gb0[c] = acc[4];
  }
}

// stage 2 alone: partial[colreduce_blocks(n)][C][5] already holds the per-256-row sums (written by the data-gradient
// epilogue of wire_gemmx3h.hip, GemmEpiParams::cr_partial)
hipError_t launch_colreduce_final(hipStream_t s, int C, int D, int64_t n, float* partial, float* gW0, float* gb0) {
  return launch_colreduce_final_blocks(s, C, D, colreduce_blocks(n), partial, gW0, gb0);
}
// the same over nblk blocks of any row count (the data-gradient chain of wire_fused.hip writes one per workgroup); partial
// must have room for (nblk + 32) * C * 5 floats
hipError_t launch_colreduce_final_blocks(hipStream_t s, int C, int D, int nblk, float* partial, float* gW0, float* gb0) {
  if (D > 4 || nblk < 1) return hipErrorInvalidValue;
  float* p2 = partial + (size_t)nblk * C * 5;            // slack reserved by colreduce_partial_floats
  const int nb = prereduce(s, partial, nblk, C * 5, p2);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(C, 64)), dim3(256), 0, s, nb == nblk ? partial : p2, nb,
                     C, D, gW0, gb0);
  return hipGetLastError();
}
hipError_t launch_colreduce(hipStream_t s, const float* G, int ldg, int C, const float* x, int D,
                            int64_t n, float* partial, float* gW0, float* gb0) {
  if (D > 4 || (ldg & 3)) return hipErrorInvalidValue;
  const int nblk = colreduce_blocks(n);
  dim3 grid((unsigned)nblk, cdiv(C, 256));
  hipLaunchKernelGGL(colreduce_kernel, grid, dim3(256), 0, s, G, ldg, C, x, D, (long long)n, partial);
  return launch_colreduce_final(s, C, D, n, partial, gW0, gb0);
}

// ===========================================================================
// layout conversion (per-layer API only; the fused path never leaves the
// blocked layout)
// ===========================================================================
__global__ void c64_to_blocked_kernel(const float* __restrict__ src, long long n, int K, int P,
                                      float* __restrict__ dst) {
  const long long row = blockIdx.x;
  for (int c = threadIdx.x; c < P; c += blockDim.x) {
    int o, part;
    blk_decode(c, o, part);
    dst[row * P + c] = o < K ? src[(row * K + o) * 2 + part] : 0.f;
  }
}
__global__ void blocked_to_c64_kernel(const float* __restrict__ src, long long n, int K, int P,
                                      float* __restrict__ dst) {
  const long long row = blockIdx.x;
  for (int e = threadIdx.x; e < 2 * K; e += blockDim.x) {
    const int o = e >> 1, part = e & 1;
    dst[row * 2 * K + e] = src[row * P + blk_col(o, part)];
  }
}
__global__ void pad_rows_kernel(const float* __restrict__ src, long long n, int K, int P,
                                float* __restrict__ dst) {
  const long long row = blockIdx.x;
  for (int c = threadIdx.x; c < P; c += blockDim.x) dst[row * P + c] = c < K ? src[row * K + c] : 0.f;
}
__global__ void unpad_rows_kernel(const float* __restrict__ src, long long n, int K, int P,
                                  float* __restrict__ dst) {
  const long long row = blockIdx.x;
  for (int c = threadIdx.x; c < K; c += blockDim.x) dst[row * K + c] = src[row * P + c];
}

hipError_t launch_c64_to_blocked(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(c64_to_blocked_kernel, dim3((unsigned)n), dim3(256), 0, s, src, (long long)n, K, P, dst);
  return hipGetLastError();
}
hipError_t launch_blocked_to_c64(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(blocked_to_c64_kernel, dim3((unsigned)n), dim3(256), 0, s, src, (long long)n, K, P, dst);
  return hipGetLastError();
}
hipError_t launch_pad_rows(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, src, (long long)n, K, P, dst);
  return hipGetLastError();
}
hipError_t launch_unpad_rows(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, src, (long long)n, K, P, dst);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// elementwise Gabor gradient (per-layer API; the fused path does this inside
// the data-gradient GEMM's epilogue)
// ---------------------------------------------------------------------------
__global__ void gabor_bwd_point_kernel(const float* __restrict__ g, const float* __restrict__ lin,
                                       const float* __restrict__ out, long long n, int P, float omega,
                                       float scale, float* __restrict__ g_lin) {
  const long long row = blockIdx.x;
  const float m2s2 = -2.f * scale * scale;
  for (int f = threadIdx.x; f < (P >> 1); f += blockDim.x) {
    const size_t c = (size_t)row * P + blk_col(f, 0);
    float gl_re, gl_im;
    gabor_bwd(g[c], g[c + 32], lin[c], lin[c + 32], out[c], out[c + 32], omega, m2s2, gl_re, gl_im);
    g_lin[c] = gl_re;
    g_lin[c + 32] = gl_im;
  }
}
__global__ void gabor_bwd_first_point_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                             const float* __restrict__ coords, int D,
                                             const float* __restrict__ W0, const float* __restrict__ b0,
                                             long long n, int K, int P, float omega, float scale,
                                             float* __restrict__ g_u, int ldu) {
  const long long row = blockIdx.x;
  const float m2s2 = -2.f * scale * scale;
  for (int f = threadIdx.x; f < ldu; f += blockDim.x) {
    float gu = 0.f;
    if (f < K) {
      float u = b0[f];
      for (int d = 0; d < D; ++d) u = __builtin_fmaf(coords[row * D + d], W0[f * D + d], u);
      const size_t c = (size_t)row * P + blk_col(f, 0);
      gu = gabor_bwd_real(g[c], g[c + 32], u, out[c], out[c + 32], omega, m2s2);
    }
    g_u[row * ldu + f] = gu;
  }
}
// 2-D Gabor (modules/wire2d.py:56-67) activation gradient for the per-layer API.
// g, out: [n][P] blocked planar; linsy, g_linsy: [n][2P] in 128-column groups (u | v | p | q) of 32 features.
//   c = conj(out) g, t = -2 s0^2 Re c:  g_lin = t lin - j w0 c,  g_sy = t sy
__global__ void gabor2d_bwd_point_kernel(const float* __restrict__ g, const float* __restrict__ linsy,
                                         const float* __restrict__ out, long long n, int P, float omega,
                                         float scale, float* __restrict__ g_linsy) {
  const long long row = blockIdx.x;
  const float m2s2 = -2.f * scale * scale;
  for (int f = threadIdx.x; f < (P >> 1); f += blockDim.x) {
    const size_t oc = (size_t)row * P + blk_col(f, 0);
    const size_t lc = (size_t)row * 2 * P + ((f >> 5) << 7) + (f & 31);
    const float gr = g[oc], gi = g[oc + 32], pr = out[oc], pi = out[oc + 32];
    const float c_r = __builtin_fmaf(pr, gr, pi * gi);
    const float c_i = __builtin_fmaf(pr, gi, -(pi * gr));
    const float t = m2s2 * c_r;
    g_linsy[lc] = __builtin_fmaf(t, linsy[lc], omega * c_i);
    g_linsy[lc + 32] = __builtin_fmaf(t, linsy[lc + 32], -(omega * c_r));
    g_linsy[lc + 64] = t * linsy[lc + 64];
    g_linsy[lc + 96] = t * linsy[lc + 96];
  }
}
// real first layer of wire2d: u = W0 x + b0, p = V0 x + c0;  g_u = t u + w0 Im c,  g_p = t p;  g_up [n][2 ldu]
__global__ void gabor2d_bwd_first_point_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                               const float* __restrict__ coords, int D,
                                               const float* __restrict__ W0, const float* __restrict__ b0,
                                               const float* __restrict__ V0, const float* __restrict__ c0,
                                               long long n, int K, int P, float omega, float scale,
                                               float* __restrict__ g_up, int ldu) {
  const long long row = blockIdx.x;
  const float m2s2 = -2.f * scale * scale;
  for (int f = threadIdx.x; f < ldu; f += blockDim.x) {
    float gu = 0.f, gp = 0.f;
    if (f < K) {
      float u = b0[f], pp = c0[f];
      for (int d = 0; d < D; ++d) {
        const float x = coords[row * D + d];
        u = __builtin_fmaf(x, W0[f * D + d], u);
        pp = __builtin_fmaf(x, V0[f * D + d], pp);
      }
      const size_t c = (size_t)row * P + blk_col(f, 0);
      const float gr = g[c], gi = g[c + 32], pr = out[c], pi = out[c + 32];
      const float c_r = __builtin_fmaf(pr, gr, pi * gi);
      const float c_i = __builtin_fmaf(pr, gi, -(pi * gr));
      const float t = m2s2 * c_r;
      gu = __builtin_fmaf(t, u, omega * c_i);
      gp = t * pp;
    }
    g_up[row * 2 * ldu + f] = gu;
    g_up[row * 2 * ldu + ldu + f] = gp;
  }
}
hipError_t launch_gabor2d_bwd_point(hipStream_t s, const float* g, const float* linsy, const float* out,
                                    int64_t n, int P, float omega, float scale, float* g_linsy) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gabor2d_bwd_point_kernel, dim3((unsigned)n), dim3(256), 0, s, g, linsy, out, (long long)n, P,
                     omega, scale, g_linsy);
  return hipGetLastError();
}
hipError_t launch_gabor2d_bwd_first_point(hipStream_t s, const float* g, const float* out, const float* coords,
                                          int D, const float* W0, const float* b0, const float* V0,
                                          const float* c0, int64_t n, int K, int P, float omega, float scale,
                                          float* g_up, int ldu) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gabor2d_bwd_first_point_kernel, dim3((unsigned)n), dim3(256), 0, s, g, out, coords, D, W0,
                     b0, V0, c0, (long long)n, K, P, omega, scale, g_up, ldu);
  return hipGetLastError();
}
template <int ACT>
__global__ void real_act_bwd_point_kernel(const float* __restrict__ g, const float* __restrict__ lin,
                                          const float* __restrict__ out, long long n, int P, float omega,
                                          float scale, float* __restrict__ g_lin) {
  const long long row = blockIdx.x;
  for (int c = threadIdx.x; c < P; c += blockDim.x) {
    const size_t i = (size_t)row * P + c;
    g_lin[i] = real_act_bwd<ACT>(g[i], lin[i], out[i], omega, scale);
  }
}
hipError_t launch_real_act_bwd_point(hipStream_t s, int kind, const float* g, const float* lin,
                                     const float* out, int64_t n, int P, float omega, float scale,
                                     float* g_lin) {
  if (n <= 0) return hipSuccess;
  dim3 grid((unsigned)n), blk(256);
  switch (kind) {
    case NK_SIREN: hipLaunchKernelGGL(real_act_bwd_point_kernel<ACT_SIREN>, grid, blk, 0, s, g, lin, out, (long long)n, P, omega, scale, g_lin); break;
    case NK_GAUSS: hipLaunchKernelGGL(real_act_bwd_point_kernel<ACT_GAUSS>, grid, blk, 0, s, g, lin, out, (long long)n, P, omega, scale, g_lin); break;
    case NK_RELU: hipLaunchKernelGGL(real_act_bwd_point_kernel<ACT_RELU>, grid, blk, 0, s, g, lin, out, (long long)n, P, omega, scale, g_lin); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_gabor_bwd_point(hipStream_t s, const float* g, const float* lin, const float* out,
                                  int64_t n, int P, float omega, float scale, float* g_lin) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gabor_bwd_point_kernel, dim3((unsigned)n), dim3(256), 0, s, g, lin, out,
                     (long long)n, P, omega, scale, g_lin);
  return hipGetLastError();
}
hipError_t launch_gabor_bwd_first_point(hipStream_t s, const float* g, const float* out,
                                        const float* coords, int D, const float* W0, const float* b0,
                                        int64_t n, int K, int P, float omega, float scale, float* g_u,
                                        int ldu) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gabor_bwd_first_point_kernel, dim3((unsigned)n), dim3(256), 0, s, g, out, coords,
                     D, W0, b0, (long long)n, K, P, omega, scale, g_u, ldu);
  return hipGetLastError();
}

// ===========================================================================
// positional encoding, modules/relu.py:62-75: [c, {sin(2^i pi c_j), cos(2^i pi c_j)}_{i,j}]
// ===========================================================================
__global__ void posenc_kernel(const float* __restrict__ coords, long long n, int D, int F, int Pin,
                              float* __restrict__ dst) {
  const long long row = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int c = threadIdx.x & 63;
  for (int col = c; col < Pin; col += 64) {
    float v = 0.f;
    if (col < D) {
      v = coords[row * D + col];
    } else if (col < D + 2 * D * F) {
      const int e = col - D;
      const int i = e / (2 * D);
      const int j = (e % (2 * D)) >> 1;
      const float freq = (float)((double)(1 << i) * 3.14159265358979323846);
      float sn, cs;
      wire_sincos(freq * coords[row * D + j], sn, cs);
      v = (e & 1) ? cs : sn;
    }
    dst[row * Pin + col] = v;
  }
}
hipError_t launch_posenc(hipStream_t s, const float* coords, int64_t n, int D, int F, int Pin,
                         float* dst) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(posenc_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, coords, (long long)n, D, F,
                     Pin, dst);
  return hipGetLastError();
}

// ===========================================================================
// training glue
// ===========================================================================
// Keyed bijection of [0, n): the per-epoch shuffle of wire_image_denoise.py:142 / wire_occupancy.py:137
// (torch.randperm) as a function of the POSITION, so that a rank generates exactly the slice of the epoch's
// permutation it trains on -- O(shard) work and memory per rank whatever the world size and the grid size
// (randperm sorts all n keys on every rank: 0.57 ms at n = 262 144, 1 GB of int64 at 512^3).
// b = bits of the smallest power of two >= n; four rounds of  x = (x * M_r + K_r) mod 2^b; x ^= x >> s  (each
// invertible mod 2^b), then cycle-walking: re-apply while x >= n (2^b < 2 n, so < 2 applications on average;
// it terminates because the walk stays on the cycle of a permutation of [0, 2^b) that contains the start).
// (tests/test_shuffle.py holds the bit-exact numpy twin.)
__device__ __host__ inline unsigned long long perm_splitmix(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
struct PermKeys { unsigned long long m[4], k[4]; };
__device__ inline unsigned long long perm_apply(unsigned long long x, unsigned long long n, int b, int sh,
                                                const PermKeys& K) {
  const unsigned long long mask = (b >= 64) ? ~0ull : ((1ull << b) - 1ull);
  do {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x = (x * K.m[r] + K.k[r]) & mask;
      x ^= x >> sh;
    }
  } while (x >= n);
  return x;
}
__global__ void perm_indices_kernel(PermKeys K, long long n_total, int b, int sh, long long first, long long count,
                                    int64_t* __restrict__ idx_out) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  idx_out[r] = (int64_t)perm_apply((unsigned long long)(first + r), (unsigned long long)n_total, b, sh, K);
}
hipError_t launch_perm_indices(hipStream_t s, uint64_t seed, int64_t n_total, int64_t first, int64_t count,
                               int64_t* idx_out) {
  if (count <= 0) return hipSuccess;
  if (n_total < 1 || first < 0 || first + count > n_total) return hipErrorInvalidValue;
  int b = 0;
  while (b < 63 && (1ull << b) < (unsigned long long)n_total) ++b;
  if (b == 0) b = 1;                       // n = 1: domain {0, 1}, cycle-walking maps 0 -> 0
  const int sh = b / 2 > 0 ? b / 2 : 1;
  PermKeys K;
  for (int r = 0; r < 4; ++r) {
    K.m[r] = perm_splitmix(seed * 8 + r) | 1ull;          // odd multiplier: a bijection mod 2^b
    K.k[r] = perm_splitmix(seed * 8 + 4 + r);
  }
  hipLaunchKernelGGL(perm_indices_kernel, dim3(cdiv(count, 256)), dim3(256), 0, s, K, (long long)n_total, b, sh,
                     (long long)first, (long long)count, idx_out);
  return hipGetLastError();
}

__global__ void coords_kernel(const int64_t* __restrict__ idx, long long first, long long n,
                              const float* __restrict__ tx, int W, const float* __restrict__ ty,
                              int H, const float* __restrict__ tz, int T,
                              float* __restrict__ coords) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  long long id = idx ? idx[r] : first + r;
  if (tz) {
    const long long k = id % T;
    id /= T;
    const long long j = id % W, i = id / W;
    coords[r * 3 + 0] = tx[j];
    coords[r * 3 + 1] = ty[i];
    coords[r * 3 + 2] = tz[k];
  } else {
    const long long j = id % W, i = id / W;
    coords[r * 2 + 0] = tx[j];
    coords[r * 2 + 1] = ty[i];
  }
}
hipError_t launch_coords(hipStream_t s, const int64_t* idx, int64_t first, int64_t n,
                         const float* tx, int W, const float* ty, int H, const float* tz, int T,
                         float* coords) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(coords_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, idx, (long long)first,
                     (long long)n, tx, W, ty, H, tz, T, coords);
  return hipGetLastError();
}

#define MSE_BLOCKS 1024
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ y,
                                                       const float* __restrict__ target,
                                                       const int64_t* __restrict__ idx,
                                                       long long first, long long n, int O,
                                                       float gscale, float* __restrict__ g_y,
                                                       float* __restrict__ rec,
                                                       float* __restrict__ partial) {
  __shared__ float red[256];
  const long long total = n * O;
  float acc = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / O;
    const int o = (int)(e - r * O);
    const long long src = idx ? idx[r] : first + r;
    const float yy = y[e];
    const float d = yy - target[src * O + o];
    g_y[e] = gscale * d;
    if (rec) rec[src * O + o] = yy;
    acc = __builtin_fmaf(d, d, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void mse_final_kernel(const float* __restrict__ partial, int nb, float lscale,
                                 float* __restrict__ loss_out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] * lscale;
}
// loss_out[0] = lscale * sum of nb partial sums (one block)
hipError_t launch_mse_final(hipStream_t s, const float* partial, int nb, float lscale, float* loss_out) {
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, lscale, loss_out);
  return hipGetLastError();
}
hipError_t launch_mse_grad(hipStream_t s, const float* y, const float* target, const int64_t* idx,
                           int64_t first, int64_t n, int O, float weight, float* g_y,
                           float* loss_out, float* rec, float* partial) {
  if (n <= 0) return hipSuccess;
  unsigned nb = cdiv(n * O, 256);
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  const float inv = (float)(1.0 / ((double)n * (double)O));
  hipLaunchKernelGGL(mse_grad_kernel, dim3(nb), dim3(256), 0, s, y, target, idx, (long long)first,
                     (long long)n, O, weight * 2.f * inv, g_y, rec, partial);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)nb, weight * inv,
                     loss_out);
  return hipGetLastError();
}

// ===========================================================================
// super-resolution loss (wire_SISR.py:151-161): rec = AvgPool2d(scale)(rec_hr as [O][H][W]),
// loss = mean((gt_lr - rec)^2) over H2 W2 O, H2 = H / scale, W2 = W / scale (floor: AvgPool2d's default
// ceil_mode = False drops ragged borders).  One thread per pooled element; it also scatters
// dL/d rec_hr = 2 (rec - gt_lr) / (H2 W2 O scale^2) to its scale x scale window.  y, g_y: [H W][O].
// ===========================================================================
__global__ __launch_bounds__(256) void avgpool_mse_grad_kernel(const float* __restrict__ y, int H, int W, int O,
                                                               int sc, int H2, int W2,
                                                               const float* __restrict__ gt_lr, float gscale,
                                                               float* __restrict__ g_y,
                                                               float* __restrict__ rec_lr,
                                                               float* __restrict__ partial) {
  __shared__ float red[256];
  const long long total = (long long)H2 * W2 * O;
  const float inv = 1.f / (float)(sc * sc);
  float acc = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long pix = e / O;
    const int o = (int)(e - pix * O);
    const int pi = (int)(pix / W2), pj = (int)(pix - (long long)pi * W2);
    const float* src = y + ((size_t)(pi * sc) * W + (size_t)pj * sc) * O + o;
    float sum = 0.f;
    for (int a = 0; a < sc; ++a)
      for (int b = 0; b < sc; ++b) sum += src[((size_t)a * W + b) * O];
    const float pool = sum * inv;
    const float d = pool - gt_lr[e];
    if (rec_lr) rec_lr[e] = pool;
    const float g = gscale * d;
    float* dst = g_y + ((size_t)(pi * sc) * W + (size_t)pj * sc) * O + o;
    for (int a = 0; a < sc; ++a)
      for (int b = 0; b < sc; ++b) dst[((size_t)a * W + b) * O] = g;
    acc = __builtin_fmaf(d, d, acc);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
hipError_t launch_avgpool_mse_grad(hipStream_t s, const float* y, int H, int W, int O, int scale,
                                   const float* gt_lr, float* g_y, float* rec_lr, float* loss_out,
                                   float* partial) {
  const int H2 = H / scale, W2 = W / scale;
  if (H2 < 1 || W2 < 1) return hipErrorInvalidValue;
  // ragged borders receive no gradient
  if (H2 * scale != H || W2 * scale != W) {
    hipError_t e = hipMemsetAsync(g_y, 0, (size_t)H * W * O * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  const long long total = (long long)H2 * W2 * O;
  unsigned nb = cdiv(total, 256);
  if (nb > MSE_BLOCKS) nb = MSE_BLOCKS;
  const double invn = 1.0 / (double)total;
  hipLaunchKernelGGL(avgpool_mse_grad_kernel, dim3(nb), dim3(256), 0, s, y, H, W, O, scale, H2, W2, gt_lr,
                     (float)(2.0 * invn / ((double)scale * scale)), g_y, rec_lr, partial);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)nb, (float)invn, loss_out);
  return hipGetLastError();
}

// torch.optim.Adam (_single_tensor_adam): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) m/denom
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, long long count,
                            float step_size, float beta1, float beta2, float eps,
                            float inv_sqrt_bc2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float gg = g[i];
  const float mm = __builtin_fmaf(beta1, m[i], (1.f - beta1) * gg);
  const float vv = __builtin_fmaf(beta2, v[i], (1.f - beta2) * gg * gg);
  m[i] = mm;
  v[i] = vv;
  const float denom = __builtin_fmaf(__builtin_sqrtf(vv), inv_sqrt_bc2, eps);
  p[i] = p[i] - step_size * (mm / denom);
}
hipError_t launch_adam(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t count,
                       float step_size, float beta1, float beta2, float eps, float inv_sqrt_bc2) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(count, 256)), dim3(256), 0, s, p, g, m, v,
                     (long long)count, step_size, beta1, beta2, eps, inv_sqrt_bc2);
  return hipGetLastError();
}

// ===========================================================================
// evaluation metrics on device (wire_image_denoise.py:161-178, wire_occupancy.py:160-162):
//   mode 0: out = { sum (gt - rec)^2, max gt }            -> PSNR = 10 log10(max(x) / mse)
//   mode 1: out = { |pred>=thres AND gt!=0|, |pred>=thres OR gt!=0| }   -> IoU
// two-level deterministic reduction (block partials, then one block)
// ===========================================================================
#define MET_BLOCKS 1024
__global__ __launch_bounds__(256) void metric_kernel(int mode, const float* __restrict__ rec,
                                                     const float* __restrict__ gt, long long count,
                                                     float thres, float* __restrict__ partial) {
  __shared__ float r0[256], r1[256];
  float a = 0.f, b = mode == 0 ? -3.4e38f : 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
    const float x = gt[e], y = rec[e];
    if (mode == 0) {
      const float d = x - y;
      a = __builtin_fmaf(d, d, a);
      b = x > b ? x : b;
    } else {
      const bool p = y >= thres, q = x != 0.f;
      a += (p && q) ? 1.f : 0.f;
      b += (p || q) ? 1.f : 0.f;
    }
  }
  r0[threadIdx.x] = a; r1[threadIdx.x] = b;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) {
      r0[threadIdx.x] += r0[threadIdx.x + sft];
      r1[threadIdx.x] = mode == 0 ? (r1[threadIdx.x] > r1[threadIdx.x + sft] ? r1[threadIdx.x] : r1[threadIdx.x + sft])
                                  : r1[threadIdx.x] + r1[threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = r0[0]; partial[2 * blockIdx.x + 1] = r1[0]; }
}
__global__ void metric_final_kernel(int mode, const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float r0[256], r1[256];
  float a = 0.f, b = mode == 0 ? -3.4e38f : 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) {
    a += partial[2 * i];
    const float v = partial[2 * i + 1];
    b = mode == 0 ? (v > b ? v : b) : b + v;
  }
  r0[threadIdx.x] = a; r1[threadIdx.x] = b;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if (threadIdx.x < sft) {
      r0[threadIdx.x] += r0[threadIdx.x + sft];
      r1[threadIdx.x] = mode == 0 ? (r1[threadIdx.x] > r1[threadIdx.x + sft] ? r1[threadIdx.x] : r1[threadIdx.x + sft])
                                  : r1[threadIdx.x] + r1[threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = r0[0]; out[1] = r1[0]; }
}
hipError_t launch_metric(hipStream_t s, int mode, const float* rec, const float* gt, int64_t count, float thres,
                         float* out, float* partial) {
  if (count <= 0) return hipErrorInvalidValue;
  unsigned nb = cdiv(count, 256);
  if (nb > MET_BLOCKS) nb = MET_BLOCKS;
  hipLaunchKernelGGL(metric_kernel, dim3(nb), dim3(256), 0, s, mode, rec, gt, (long long)count, thres, partial);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(256), 0, s, mode, partial, (int)nb, out);
  return hipGetLastError();
}

// ===========================================================================
// 3M complex path (wire_gemm3m.hip): weights as blocked-planar complex matrices
//   Wb_fwd[o][(i,re|im)] = W[o][i]            (lin = z W^T)
//   Wb_dg [i][(o,re|im)] = conj(W[o][i])      (g_z = g_lin conj(W))
// ===========================================================================
WIRE_DEVINL void pack3m_body(const float* __restrict__ W, const float* __restrict__ b, int K, int Kin,
                             int Kp, int Kpin, float* __restrict__ Wb_fwd, float* __restrict__ Wb_dg,
                             float* __restrict__ bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // input feature (padded)
  const int o = blockIdx.y;                              // output feature (padded)
  if (i >= Kpin) return;
  float wr = 0.f, wi = 0.f;
  if (o < K && i < Kin) {
    wr = W[((size_t)o * Kin + i) * 2];
    wi = W[((size_t)o * Kin + i) * 2 + 1];
  }
  const int ci = blk_col(i, 0), co = blk_col(o, 0);
  Wb_fwd[(size_t)o * (2 * Kpin) + ci] = wr;
  Wb_fwd[(size_t)o * (2 * Kpin) + ci + 32] = wi;
  Wb_dg[(size_t)i * (2 * Kp) + co] = wr;
  Wb_dg[(size_t)i * (2 * Kp) + co + 32] = -wi;
  if (i == 0) {
    bias[co] = o < K ? b[2 * o] : 0.f;
    bias[co + 32] = o < K ? b[2 * o + 1] : 0.f;
  }
}
__global__ void pack3m_kernel(const float* __restrict__ W, const float* __restrict__ b, int K, int Kin,
                              int Kp, int Kpin, float* __restrict__ Wb_fwd, float* __restrict__ Wb_dg,
                              float* __restrict__ bias) {
  pack3m_body(W, b, K, Kin, Kp, Kpin, Wb_fwd, Wb_dg, bias);
}
__global__ void pack3m_batch_kernel(PackBatch pb, int K, int Kin, int Kp, int Kpin) {
  const int z = blockIdx.z;
  pack3m_body(pb.W[z], pb.b[z], K, Kin, Kp, Kpin, pb.fwd[z], pb.dg[z], pb.bias[z]);
}
// up to PACK_MAXB layers per launch: pb.fwd / pb.dg / pb.bias = the 3M images and the (shared) blocked bias
hipError_t launch_pack3m_batch(hipStream_t s, const PackBatch& pb, int nb, int K, int Kin, int Kp, int Kpin) {
  if (nb < 1 || nb > PACK_MAXB) return hipErrorInvalidValue;
  dim3 grid(cdiv(Kpin, 64), (unsigned)Kp, (unsigned)nb);
  hipLaunchKernelGGL(pack3m_batch_kernel, grid, dim3(64), 0, s, pb, K, Kin, Kp, Kpin);
  return hipGetLastError();
}
hipError_t launch_pack3m(hipStream_t s, const float* W, const float* b, int K, int Kin, int Kp, int Kpin,
                         float* Wb_fwd, float* Wb_dg, float* bias) {
  dim3 grid(cdiv(Kpin, 64), (unsigned)Kp);
  hipLaunchKernelGGL(pack3m_kernel, grid, dim3(64), 0, s, W, b, K, Kin, Kp, Kpin, Wb_fwd, Wb_dg, bias);
  return hipGetLastError();
}

// g_W.re = P1 + P2, g_W.im = P3 - P1 + P2 summed over the row splits; g_b from the column sums
__global__ __launch_bounds__(256) void wgrad3m_reduce_kernel(const float* __restrict__ slab,
                                      const float* __restrict__ bslab, int S, int K, int Kin, int Kp_o,
                                      int Kp_i, float* __restrict__ gW, float* __restrict__ gb) {
  // block = 64 input features x 4 groups of row splits; partial sums combined through LDS
  __shared__ float red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  const int o = blockIdx.y;
  const size_t plane = (size_t)Kp_o * Kp_i;
  float sr = 0.f, si = 0.f;
  if (i < Kin) {
    for (int s = ty; s < S; s += 4) {
      const float* p = slab + (size_t)s * 3 * plane + (size_t)o * Kp_i + i;
      const float p1 = p[0], p2 = p[plane], p3 = p[2 * plane];
      sr += p1 + p2;
      si += (p3 - p1) + p2;
    }
  }
  red[0][ty][tx] = sr;
  red[1][ty][tx] = si;
  __syncthreads();
  if (ty == 0 && i < Kin) {
    gW[((size_t)o * Kin + i) * 2] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
    gW[((size_t)o * Kin + i) * 2 + 1] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float br = 0.f, bi = 0.f;
    for (int s = 0; s < S; ++s) {
      br += bslab[((size_t)s * 2 + 0) * Kp_o + o];
      bi += bslab[((size_t)s * 2 + 1) * Kp_o + o];
    }
    gb[2 * o] = br;
    gb[2 * o + 1] = bi;
  }
}
hipError_t launch_wgrad3m_reduce(hipStream_t s, const float* slab, const float* bslab, int S, int K, int Kin,
                                 int Kp_o, int Kp_i, float* gW, float* gb) {
  dim3 grid(cdiv(Kin, 64), (unsigned)K);
  hipLaunchKernelGGL(wgrad3m_reduce_kernel, grid, dim3(256), 0, s, slab, bslab, S, K, Kin, Kp_o, Kp_i, gW, gb);
  return hipGetLastError();
}

// ===========================================================================
// trainable omega_0 / scale_0 of a ComplexGaborLayer (modules/wire.py:80-81, trainable=True):
//   out = exp(j w0 lin - s0^2 |lin|^2)  =>  d out / d w0 = j lin out,  d out / d s0 = -2 s0 |lin|^2 out
// and with c = conj(out) g (g = dL/dRe + j dL/dIm):
//   dL/dw0 = sum Re(conj(g) j lin out) = sum (lin_re c_im - lin_im c_re),   dL/ds0 = -2 s0 sum |lin|^2 Re c.
// lin: complex rows [n][P] blocked planar, or (is_first) real u [n][P/2]; g, out [n][P] blocked planar.
// Two-stage deterministic reduction: partial[2][nblk] then one block.
// ===========================================================================
#define HP_ROWS 32
__global__ __launch_bounds__(256) void gabor_hparam_partial_kernel(const float* __restrict__ g,
                                                                   const float* __restrict__ lin,
                                                                   const float* __restrict__ out, long long n, int K,
                                                                   int P, int is_first, float* __restrict__ partial) {
  __shared__ float red[2][256];
  const long long r0 = (long long)blockIdx.x * HP_ROWS;
  long long r1 = r0 + HP_ROWS;
  if (r1 > n) r1 = n;
  float aw = 0.f, as = 0.f;
  for (long long row = r0; row < r1; ++row)
    for (int f = threadIdx.x; f < K; f += 256) {
      const size_t c = (size_t)row * P + blk_col(f, 0);
      const float gr = g[c], gi = g[c + 32], pr = out[c], pi = out[c + 32];
      float lr, li;
      if (is_first) { lr = lin[(size_t)row * (P >> 1) + f]; li = 0.f; }
      else { lr = lin[c]; li = lin[c + 32]; }
      const float c_r = __builtin_fmaf(pr, gr, pi * gi);
      const float c_i = __builtin_fmaf(pr, gi, -(pi * gr));
      aw += __builtin_fmaf(lr, c_i, -(li * c_r));
      as += __builtin_fmaf(lr, lr, li * li) * c_r;
    }
  red[0][threadIdx.x] = aw;
  red[1][threadIdx.x] = as;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if ((int)threadIdx.x < sft) {
      red[0][threadIdx.x] += red[0][threadIdx.x + sft];
      red[1][threadIdx.x] += red[1][threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0][0];
    partial[gridDim.x + blockIdx.x] = red[1][0];
  }
}
__global__ __launch_bounds__(256) void gabor_hparam_final_kernel(const float* __restrict__ partial, int nblk,
                                                                 float scale, float* __restrict__ out2) {
  __shared__ float red[2][256];
  float aw = 0.f, as = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) { aw += partial[b]; as += partial[nblk + b]; }
  red[0][threadIdx.x] = aw;
  red[1][threadIdx.x] = as;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if ((int)threadIdx.x < sft) {
      red[0][threadIdx.x] += red[0][threadIdx.x + sft];
      red[1][threadIdx.x] += red[1][threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = red[0][0];
    out2[1] = -2.f * scale * red[1][0];
  }
}
// the same for ComplexGaborLayer2D (modules/wire2d.py:56-67): out = exp(j w0 lin) exp(-s0^2 (|lin|^2 + |sy|^2))
//   d out / d w0 = j lin out,  d out / d s0 = -2 s0 (|lin|^2 + |sy|^2) out.
// linsy: [n][2P] in 128-column groups (u | v | p | q), or (is_first) real (u | p) [n][2 * P/2].
__global__ __launch_bounds__(256) void gabor2d_hparam_partial_kernel(const float* __restrict__ g,
                                                                     const float* __restrict__ linsy,
                                                                     const float* __restrict__ out, long long n, int K,
                                                                     int P, int is_first, float* __restrict__ partial) {
  __shared__ float red[2][256];
  const long long r0 = (long long)blockIdx.x * HP_ROWS;
  long long r1 = r0 + HP_ROWS;
  if (r1 > n) r1 = n;
  const int Kp = P >> 1;
  float aw = 0.f, as = 0.f;
  for (long long row = r0; row < r1; ++row)
    for (int f = threadIdx.x; f < K; f += 256) {
      const size_t c = (size_t)row * P + blk_col(f, 0);
      const float gr = g[c], gi = g[c + 32], pr = out[c], pi = out[c + 32];
      float u, v = 0.f, pp, qq = 0.f;
      if (is_first) {
        u = linsy[(size_t)row * (2 * Kp) + f];
        pp = linsy[(size_t)row * (2 * Kp) + Kp + f];
      } else {
        const float* L = linsy + (size_t)row * (2 * P) + ((f >> 5) << 7) + (f & 31);
        u = L[0]; v = L[32]; pp = L[64]; qq = L[96];
      }
      const float c_r = __builtin_fmaf(pr, gr, pi * gi);
      const float c_i = __builtin_fmaf(pr, gi, -(pi * gr));
      aw += __builtin_fmaf(u, c_i, -(v * c_r));
      as += (__builtin_fmaf(u, u, v * v) + __builtin_fmaf(pp, pp, qq * qq)) * c_r;
    }
  red[0][threadIdx.x] = aw;
  red[1][threadIdx.x] = as;
  __syncthreads();
  for (int sft = 128; sft >= 1; sft >>= 1) {
    if ((int)threadIdx.x < sft) {
      red[0][threadIdx.x] += red[0][threadIdx.x + sft];
      red[1][threadIdx.x] += red[1][threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0][0];
    partial[gridDim.x + blockIdx.x] = red[1][0];
  }
}
int hparam_blocks(int64_t n) { return (int)((n + HP_ROWS - 1) / HP_ROWS); }
hipError_t launch_gabor2d_hparam_grad(hipStream_t s, const float* g, const float* linsy, const float* out, int64_t n,
                                      int K, int P, int is_first, float scale, float* partial, float* out2) {
  if (n <= 0) return hipErrorInvalidValue;
  const int nblk = (int)((n + HP_ROWS - 1) / HP_ROWS);
  hipLaunchKernelGGL(gabor2d_hparam_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, g, linsy, out, (long long)n,
                     K, P, is_first, partial);
  hipLaunchKernelGGL(gabor_hparam_final_kernel, dim3(1), dim3(256), 0, s, partial, nblk, scale, out2);
  return hipGetLastError();
}
hipError_t launch_gabor_hparam_grad(hipStream_t s, const float* g, const float* lin, const float* out, int64_t n,
                                    int K, int P, int is_first, float scale, float* partial, float* out2) {
  if (n <= 0) return hipErrorInvalidValue;
  const int nblk = hparam_blocks(n);
  hipLaunchKernelGGL(gabor_hparam_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, g, lin, out, (long long)n,
                     K, P, is_first, partial);
  hipLaunchKernelGGL(gabor_hparam_final_kernel, dim3(1), dim3(256), 0, s, partial, nblk, scale, out2);
  return hipGetLastError();
}

// ===========================================================================
// best-reconstruction tracking without a host round trip (wire_image_denoise.py:176-178:
//   if (mse_array[epoch] < best_mse) or (epoch == 0): best_mse = ...; best_img = imrec
// wire_occupancy.py:170-172: if lossval < best_mse: ...; best_img = copy.deepcopy(im_estim)).
// Every thread reads the two scalars; the scalar itself is updated by a second one-thread launch.
// ===========================================================================
__global__ void best_copy_kernel(const float* __restrict__ metric, const float* __restrict__ best, int force,
                                 const float* __restrict__ src, float* __restrict__ dst, long long count) {
  if (!(force || metric[0] < best[0])) return;
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < count) {
    *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
  } else {
    for (long long j = i; j < count; ++j) dst[j] = src[j];
  }
}
__global__ void best_scalar_kernel(const float* __restrict__ metric, float* __restrict__ best, int force,
                                   int* __restrict__ updated) {
  const bool take = force || metric[0] < best[0];
  if (take) best[0] = metric[0];
  if (updated) updated[0] = take ? 1 : 0;
}
hipError_t launch_track_best(hipStream_t s, const float* metric, float* best, int force, const float* src,
                             float* dst, int64_t count, int* updated) {
  if (count > 0)
    hipLaunchKernelGGL(best_copy_kernel, dim3(cdiv(count, 1024)), dim3(256), 0, s, metric, best, force, src, dst,
                       (long long)count);
  hipLaunchKernelGGL(best_scalar_kernel, dim3(1), dim3(1), 0, s, metric, best, force, updated);
  return hipGetLastError();
}

// torch.sigmoid of the dense occupancy query before the cube is written out (modules/volutils.py:128-131)
__global__ void sigmoid_kernel(float* __restrict__ x, long long count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) x[i] = 1.f / (1.f + wire_exp(-x[i]));
}
hipError_t launch_sigmoid(hipStream_t s, float* x, int64_t count) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(cdiv(count, 256)), dim3(256), 0, s, x, (long long)count);
  return hipGetLastError();
}

// ===========================================================================
// CT forward operator (modules/lin_inverse.py:19-40, wire_ct.py:128-133): the parallel-beam Radon transform the
// reference builds from kornia.geometry.rotate (kornia 0.6.5: get_rotation_matrix2d about ((W-1)/2, (H-1)/2),
// positive angle = counter-clockwise, warp_affine -> affine_grid + grid_sample, bilinear, zero padding,
// align_corners = True) followed by a sum over the rows:
//     sino[a][j] = sum_i  bilinear(img, x = c (j - cx) - s (i - cy) + cx,  y = s (j - cx) + c (i - cy) + cy)
// with c = cos(theta_a), s = sin(theta_a).  Samples within one pixel outside the image interpolate against zero.
// Pinned by the gt -> sinogram pair the reference stores (multiscale_results/ct/.../info.mat).
// Forward: one thread per (angle, column), rows in the loop -> reads of a row walk a straight line of the image.
// Backward (the adjoint, for dL/dimg): same traversal, four atomic adds per sample.
// ===========================================================================
template <bool BWD>
__global__ __launch_bounds__(256) void radon_kernel(const float* __restrict__ img, const float* __restrict__ angles,
                                                    int H, int W, int A, float* __restrict__ sino,
                                                    const float* __restrict__ g_sino, float* __restrict__ g_img) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = blockIdx.y;
  if (j >= W) return;
  const float th = angles[a] * 0.017453292519943295f;
  float sn, cs;
  wire_sincos(th, sn, cs);
  const float cx = 0.5f * (float)(W - 1), cy = 0.5f * (float)(H - 1);
  const float xj = (float)j - cx;
  const float g = BWD ? g_sino[(size_t)a * W + j] : 0.f;
  float acc = 0.f;
  for (int i = 0; i < H; ++i) {
    const float yi = (float)i - cy;
    const float x = __builtin_fmaf(cs, xj, -(sn * yi)) + cx;
    const float y = __builtin_fmaf(sn, xj, cs * yi) + cy;
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    if (x0 < -1 || x0 >= W || y0 < -1 || y0 >= H) continue;
    const float wx1 = x - xf, wy1 = y - yf, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0, vx1 = x0 + 1 < W, vy0 = y0 >= 0, vy1 = y0 + 1 < H;
    if (!BWD) {
      float v = 0.f;
      if (vy0) {
        const float* r = img + (size_t)y0 * W;
        if (vx0) v += wy0 * wx0 * r[x0];
        if (vx1) v += wy0 * wx1 * r[x0 + 1];
      }
      if (vy1) {
        const float* r = img + (size_t)(y0 + 1) * W;
        if (vx0) v += wy1 * wx0 * r[x0];
        if (vx1) v += wy1 * wx1 * r[x0 + 1];
      }
      acc += v;
    } else {
      if (vy0) {
        float* r = g_img + (size_t)y0 * W;
        if (vx0) atomicAdd(r + x0, g * wy0 * wx0);
        if (vx1) atomicAdd(r + x0 + 1, g * wy0 * wx1);
      }
      if (vy1) {
        float* r = g_img + (size_t)(y0 + 1) * W;
        if (vx0) atomicAdd(r + x0, g * wy1 * wx0);
        if (vx1) atomicAdd(r + x0 + 1, g * wy1 * wx1);
      }
    }
  }
  if (!BWD) sino[(size_t)a * W + j] = acc;
}
hipError_t launch_radon_fwd(hipStream_t s, const float* img, const float* angles, int H, int W, int A, float* sino) {
  dim3 grid(cdiv(W, 256), (unsigned)A);
  hipLaunchKernelGGL(radon_kernel<false>, grid, dim3(256), 0, s, img, angles, H, W, A, sino, nullptr, nullptr);
  return hipGetLastError();
}
hipError_t launch_radon_bwd(hipStream_t s, const float* g_sino, const float* angles, int H, int W, int A,
                            float* g_img) {
  hipError_t e = hipMemsetAsync(g_img, 0, (size_t)H * W * sizeof(float), s);
  if (e != hipSuccess) return e;
  dim3 grid(cdiv(W, 256), (unsigned)A);
  hipLaunchKernelGGL(radon_kernel<true>, grid, dim3(256), 0, s, nullptr, angles, H, W, A, nullptr, g_sino, g_img);
  return hipGetLastError();
}
