// wire_point.h -- launchers of the bandwidth-bound kernels around the GEMMs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { NK_WIRE = 0, NK_WIRE2D = 1, NK_SIREN = 2, NK_GAUSS = 3, NK_RELU = 4 };

// ---- weight packing (native nn.Parameter layout -> padded real-expanded image)
hipError_t launch_pack_hidden(hipStream_t s, int kind, const float* W, const float* b,
                              const float* V, const float* c, int K, int Kin, int P, int Pin,
                              float* Bt_fwd, float* Bt_dgrad, float* bias);
// several layers of one shape per launch (grid.z = layer)
#define PACK_MAXB 16
struct PackBatch {
  const float* W[PACK_MAXB]; const float* b[PACK_MAXB]; const float* V[PACK_MAXB]; const float* c[PACK_MAXB];
  float* fwd[PACK_MAXB]; float* dg[PACK_MAXB]; float* bias[PACK_MAXB];
};
hipError_t launch_pack_hidden_batch(hipStream_t s, int kind, const PackBatch& pb, int nb, int K, int Kin, int P,
                                    int Pin);
hipError_t launch_pack3m_batch(hipStream_t s, const PackBatch& pb, int nb, int K, int Kin, int Kp, int Kpin);
hipError_t launch_pack_final(hipStream_t s, int kind, const float* Wf, const float* bf, int K,
                             int P, int O, float* wf, float* bfr);

// ---- first layer (D <= 4 inputs): elementwise, VALU/HBM-write bound
// out [n][P] blocked; lin (optional): real nets [n][P]; wire: real u [n][P/2] (per-layer API)
hipError_t launch_first_fwd(hipStream_t s, int kind, const float* coords, int64_t n, int D,
                            const float* W0, const float* b0, const float* V0, const float* c0,
                            int K, int P, float omega, float scale, float* lin, float* out,
                            unsigned* amax_out = nullptr);   // amax_out: max |out| slots (wire_dev.h) or null

// ---- final linear: y[n][O] = z[n][P] . wf[O][P] + bf
hipError_t launch_final_fwd(hipStream_t s, const float* z, int64_t n, int P, int O,
                            const float* wf, const float* bfr, float* y);

// ---- final-layer backward fused with the last hidden activation gradient.
// g_out = g_y * wf; g_lin = act'(g_out; lin, out).  Also per-block partials of
// g_wf[o][c] = sum_n g_y[n][o] out[n][c] and of sum_n g_y[n][o].
// raw != 0: skip the activation gradient and write g_out itself (per-layer API).
int final_bwd_blocks(int64_t n);
hipError_t launch_final_bwd(hipStream_t s, int kind, int raw, const float* g_y, int64_t n, int O,
                            const float* wf, const float* lin, const float* out, int K, int P,
                            float omega, float scale, float* g_lin, float* part_w,
                            float* part_b, unsigned* amax_g = nullptr);   // amax_g: max |g_lin| slots or null
// part_w / part_b must have room for final_bwd_blocks(n) + 32 blocks (pre-reduction scratch)
hipError_t launch_final_reduce(hipStream_t s, int kind, float* part_w, float* part_b,
                               int nblk, int O, int K, int P, float* gWf, float* gbf);

// ---- fused final stage of a training step (wire, O <= 4): y = Re(z Wf^T + bf), MSE loss + dL/dy,
// rec scatter, g_out = g_y conj(Wf), Gabor gradient of layer L, and the per-block partials of g_Wf /
// g_bf in the SAME layout launch_final_bwd produces (FB_ROWS rows per block) for launch_final_reduce.
bool final_fused_supported(int P, int O);
hipError_t launch_final_fused(hipStream_t s, int kind, const float* out, const float* lin, int64_t n, int P, int O,
                              int kvalid, const float* wf, const float* bfr, const float* target, const int64_t* idx,
                              int64_t first, float weight, float omega, float scale, float* y, float* rec,
                              float* g_lin, float* part_w, float* part_b, float* loss_partial,
                              float* loss_out, unsigned* amax_g = nullptr);   // amax_g: max |g_lin| slots or null

// ---- weight-gradient slab reduction: slab[S][Pm][Pn] (+ bslab[S][Pm]) -> native grads
hipError_t launch_wgrad_reduce(hipStream_t s, int kind, const float* slab, const float* bslab,
                               int S, int K, int Kin, int Pm, int Pn, float* gW, float* gb,
                               float* gV, float* gc);

// ---- first-layer weight gradient: g_W0[c][d] = sum_n G[n][c] x[n][d], g_b0[c] = sum_n G[n][c]
int colreduce_blocks(int64_t n);
// partial must have room for (colreduce_blocks(n) + 32) * C * 5 floats
hipError_t launch_colreduce_final(hipStream_t s, int C, int D, int64_t n, float* partial, float* gW0, float* gb0);
hipError_t launch_colreduce_final_blocks(hipStream_t s, int C, int D, int nblk, float* partial, float* gW0, float* gb0);
hipError_t launch_colreduce(hipStream_t s, const float* G, int ldg, int C, const float* x, int D,
                            int64_t n, float* partial, float* gW0, float* gb0);

// ---- layout conversion for the per-layer API
hipError_t launch_c64_to_blocked(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst);
hipError_t launch_blocked_to_c64(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst);
// real [n][K] <-> padded [n][P]
hipError_t launch_pad_rows(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst);
hipError_t launch_unpad_rows(hipStream_t s, const float* src, int64_t n, int K, int P, float* dst);

// ---- elementwise Gabor gradient for the per-layer API
hipError_t launch_gabor_bwd_point(hipStream_t s, const float* g, const float* lin, const float* out,
                                  int64_t n, int P, float omega, float scale, float* g_lin);
hipError_t launch_gabor_bwd_first_point(hipStream_t s, const float* g, const float* out,
                                        const float* coords, int D, const float* W0, const float* b0,
                                        int64_t n, int K, int P, float omega, float scale, float* g_u,
                                        int ldu);

// 2-D Gabor (wire2d) forms: linsy / g_linsy [n][2P] in (u | v | p | q) groups; first layer g_up [n][2 ldu]
hipError_t launch_gabor2d_bwd_point(hipStream_t s, const float* g, const float* linsy, const float* out,
                                    int64_t n, int P, float omega, float scale, float* g_linsy);
hipError_t launch_gabor2d_bwd_first_point(hipStream_t s, const float* g, const float* out, const float* coords,
                                          int D, const float* W0, const float* b0, const float* V0,
                                          const float* c0, int64_t n, int K, int P, float omega, float scale,
                                          float* g_up, int ldu);

hipError_t launch_real_act_bwd_point(hipStream_t s, int kind, const float* g, const float* lin,
                                     const float* out, int64_t n, int P, float omega, float scale,
                                     float* g_lin);

// ---- positional encoding (modules/relu.py:62-75) into a [n][Pin] padded row
hipError_t launch_posenc(hipStream_t s, const float* coords, int64_t n, int D, int F, int Pin,
                         float* dst);

// ---- training glue
// idx_out[r] = pi_seed(first + r), r < count: a keyed bijection pi_seed of [0, n_total) (the epoch's shuffle)
hipError_t launch_perm_indices(hipStream_t s, uint64_t seed, int64_t n_total, int64_t first, int64_t count,
                               int64_t* idx_out);
hipError_t launch_coords(hipStream_t s, const int64_t* idx, int64_t first, int64_t n,
                         const float* tx, int W, const float* ty, int H, const float* tz, int T,
                         float* coords);
hipError_t launch_mse_final(hipStream_t s, const float* partial, int nb, float lscale, float* loss_out);
hipError_t launch_mse_grad(hipStream_t s, const float* y, const float* target, const int64_t* idx,
                           int64_t first, int64_t n, int O, float weight, float* g_y,
                           float* loss_out, float* rec, float* partial);
// super-resolution loss: AvgPool2d(scale) of the [H W][O] reconstruction against gt_lr [H2 W2][O]
// (wire_SISR.py:151-161): loss, dL/dy (g_y [H W][O]), optionally the pooled image; partial >= 1024 floats
hipError_t launch_avgpool_mse_grad(hipStream_t s, const float* y, int H, int W, int O, int scale,
                                   const float* gt_lr, float* g_y, float* rec_lr, float* loss_out,
                                   float* partial);
hipError_t launch_adam(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t count,
                       float step_size, float beta1, float beta2, float eps, float inv_sqrt_bc2);

// ---- evaluation metrics: mode 0 -> {sum sq err, max gt}; mode 1 -> {intersection, union}
hipError_t launch_metric(hipStream_t s, int mode, const float* rec, const float* gt, int64_t count, float thres,
                         float* out, float* partial);

// ---- 3M complex path: blocked-planar complex weight matrices + slab reduction
hipError_t launch_pack3m(hipStream_t s, const float* W, const float* b, int K, int Kin, int Kp, int Kpin,
                         float* Wb_fwd, float* Wb_dg, float* bias);
hipError_t launch_wgrad3m_reduce(hipStream_t s, const float* slab, const float* bslab, int S, int K, int Kin,
                                 int Kp_o, int Kp_i, float* gW, float* gb);

// ---- trainable omega_0 / scale_0 (ComplexGaborLayer(trainable=True), modules/wire.py:80-81):
// out2 = { dL/d omega_0, dL/d scale_0 }; partial: 2 * hparam_blocks(n) floats
int hparam_blocks(int64_t n);
hipError_t launch_gabor_hparam_grad(hipStream_t s, const float* g, const float* lin, const float* out, int64_t n,
                                    int K, int P, int is_first, float scale, float* partial, float* out2);
hipError_t launch_gabor2d_hparam_grad(hipStream_t s, const float* g, const float* linsy, const float* out, int64_t n,
                                      int K, int P, int is_first, float scale, float* partial, float* out2);

// ---- best-so-far tracking on the device and the sigmoid of the mesh-export query
hipError_t launch_track_best(hipStream_t s, const float* metric, float* best, int force, const float* src,
                             float* dst, int64_t count, int* updated);
hipError_t launch_sigmoid(hipStream_t s, float* x, int64_t count);

// ---- CT forward operator: rotate-and-sum Radon transform (modules/lin_inverse.py:19-40) and its adjoint
hipError_t launch_radon_fwd(hipStream_t s, const float* img, const float* angles, int H, int W, int A, float* sino);
hipError_t launch_radon_bwd(hipStream_t s, const float* g_sino, const float* angles, int H, int W, int A,
                            float* g_img);
