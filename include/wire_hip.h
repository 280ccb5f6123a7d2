/* wire_hip.h -- C ABI of libwire_hip.so: the MI355X (gfx950) WIRE INR hot path.
 *
 * The reference (Annatk26/wire @ 2024_08_07) has no native layer and no FFI:
 * its hot path is eager PyTorch.  Each entry point
 * below names the reference code whose arithmetic it replaces; the binding a
 * maintainer adds on the reference side is a ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    the name ends in _host; `stream` is a hipStream_t passed as void*.
 *  - every function returns 0 on success or a negative wire_status; it never
 *    throws and never synchronises the device.  wire_last_error() returns the
 *    message of the calling thread's most recent failure.
 *  - the library owns no device memory: parameters, packed weights,
 *    activations and scratch live in caller buffers sized by the *_floats /
 *    *_bytes queries.  Re-entrant; callable from any host thread (autograd
 *    runs backward on its own thread).
 *  - "params" / "grads" are arrays (in HOST memory) of device pointers, one
 *    per trainable tensor in the reference's state_dict order and native
 *    layout (complex64 = interleaved re,im floats):
 *       wire   : W0[K][D] f32, b0[K] f32, {W_l[K][K] c64, b_l[K] c64} l=1..L,
 *                W_f[O][K] c64, b_f[O] c64        (modules/wire.py:127-157)
 *       wire2d : per layer l=0..L: W, b, V(scale_orth), c; then W_f, b_f
 *                                                  (modules/wire2d.py:98-123)
 *       siren/gauss/relu : {W_l, b_l} l=0..L, W_f, b_f, all f32
 *                (modules/siren.py:64-88, gauss.py:44-67, relu.py:99-120)
 *  - internal activation layout ("blocked planar"): a
 *    complex row of K features is stored as P = roundup(2K,64) floats; group
 *    g of 32 features occupies columns [64g,64g+32) = real parts and
 *    [64g+32,64g+64) = imaginary parts.  Pad features read as 0.
 */
#ifndef WIRE_HIP_H
#define WIRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WIRE_ABI_VERSION 1

typedef enum wire_status {
  WIRE_OK = 0,
  WIRE_ERR_ARG = -1,     /* bad argument / unsupported shape            */
  WIRE_ERR_HIP = -2,     /* a HIP runtime call or launch failed          */
  WIRE_ERR_SIZE = -3     /* caller buffer too small                      */
} wire_status;

/* model_dict keys of modules/models.py:15-21 that are in scope */
typedef enum wire_kind {
  WIRE_KIND_WIRE = 0,    /* modules/wire.py   ComplexGaborLayer          */
  WIRE_KIND_WIRE2D = 1,  /* modules/wire2d.py ComplexGaborLayer2D        */
  WIRE_KIND_SIREN = 2,   /* modules/siren.py  SineLayer                  */
  WIRE_KIND_GAUSS = 3,   /* modules/gauss.py  GaussLayer                 */
  WIRE_KIND_RELU = 4     /* modules/relu.py   ReLULayer (+PosEncoding)   */
} wire_kind;

/* Architecture + hyper-parameters of one INR (modules/wire.py:96-159). */
typedef struct wire_net_desc {
  int32_t kind;            /* wire_kind                                           */
  int32_t in_features;     /* D: coordinate dims (2 or 3)                         */
  int32_t width;           /* K: features per hidden layer AFTER the reference's
                              own rescale (int(h/sqrt2) wire, int(h/2) wire2d)   */
  int32_t hidden_layers;   /* L                                                   */
  int32_t out_features;    /* O (<= 8)                                            */
  int32_t posenc_freqs;    /* relu only: PosEncoding.num_frequencies, 0 = off     */
  float first_omega0;      /* omega of net[0]                                     */
  float hidden_omega0;     /* omega of net[1..L]                                  */
  float scale0;            /* Gaussian scale s0                                   */
} wire_net_desc;

int wire_abi_version(void);
const char* wire_last_error(void);

/* ---- sizes ------------------------------------------------------------ */
/* number of trainable tensors (length of the params / grads arrays)       */
int wire_num_param_tensors(const wire_net_desc* d);
/* floats in tensor t in its native layout (complex counts 2 per element)  */
int64_t wire_param_tensor_floats(const wire_net_desc* d, int t);
/* floats of the packed (padded, real-expanded) weight image               */
int64_t wire_packed_floats(const wire_net_desc* d);
/* bytes of activation storage mlp_fwd writes for `n` rows (save_for_bwd=1
 * keeps what mlp_bwd needs; 0 = inference ping-pong only)                  */
int64_t wire_act_bytes(const wire_net_desc* d, int64_t n, int save_for_bwd);
/* bytes of scratch mlp_bwd needs for `n` rows                              */
int64_t wire_bwd_scratch_bytes(const wire_net_desc* d, int64_t n);

/* ---- whole-network path ------------------------------------------------ */
/* Refresh the packed weight image from the native parameters.  Call after
 * every optimizer step (replaces nothing in the reference: ATen reads the
 * nn.Parameter storage directly, modules/wire.py:89).                      */
int wire_pack_params(void* stream, const wire_net_desc* d,
                     const void* const* params_host, float* packed);

/* INR.forward (modules/wire.py:161-167; wire2d.py:122-127; siren.py:90-96;
 * gauss.py:71-74; relu.py:124-130): coords[n][D] f32 -> y[n][O] f32.
 * `act` receives the per-layer activations (wire_act_bytes).               */
int wire_mlp_fwd(void* stream, const wire_net_desc* d, const float* packed,
                 const float* coords, int64_t n, float* y,
                 void* act, int64_t act_bytes, int save_for_bwd);

/* Backward of the above (autograd graph of modules/wire.py:89-93,156-165):
 * g_y[n][O] = dL/dy.  Writes every parameter gradient in native layout
 * (PyTorch complex convention dL/dRe + j dL/dIm) through `grads_host`;
 * grads are OVERWRITTEN (= zero_grad + backward).                          */
int wire_mlp_bwd(void* stream, const wire_net_desc* d, const float* packed,
                 const float* coords, int64_t n, const float* g_y,
                 const void* act, int64_t act_bytes,
                 void* scratch, int64_t scratch_bytes,
                 void* const* grads_host);

/* One training step's compute in a single call (wire_image_denoise.py:148-156:
 * model(b_coords) -> rec[b_indices] = pix -> mse -> backward): forward, MSE against
 * target[idx[r]] (or target[first + r] when idx is NULL), loss_out[0] = weight *
 * mean((y - t)^2), optional rec scatter, and every parameter gradient of
 * weight * loss into grads_host (overwritten).  For every net kind with at least one
 * hidden layer and O <= 4 the final linear forward, the loss, its gradient, the final
 * linear backward and the last layer's activation gradient are ONE kernel (a single pass
 * over out_L / lin_L, final_fused_kernel); nets without a hidden layer or with O > 4 run
 * wire_mlp_fwd + wire_mse_grad + wire_mlp_bwd internally.
 * y [n][O] and g_y [n][O] are outputs; partial >= 4096 floats.                 */
int wire_train_fwd_bwd(void* stream, const wire_net_desc* d, const float* packed,
                       const float* coords, int64_t n, const float* target,
                       const int64_t* idx, int64_t first, float weight, float* y,
                       float* g_y, float* loss_out, float* rec, float* partial,
                       void* act, int64_t act_bytes, void* scratch,
                       int64_t scratch_bytes, void* const* grads_host);

/* The same step for data-parallel callers that overlap the gradient exchange with the rest of the backward
 * (wire_occupancy.py:137-158 sharded over the GPUs of a node).  `ready(user, first_tensor, n_tensors)` is called ON THE
 * HOST, from inside this call, each time the kernels that produce the FINAL value of the parameter gradients
 * grads_host[first_tensor .. first_tensor + n_tensors) (state_dict order) have been enqueued on `stream` -- the final
 * linear layer first (loss_out is complete by then too), then the hidden layers from the last to the first, the first layer
 * last; every tensor is announced exactly once.  The callee typically records an event on `stream` and starts the
 * all-reduce of that slice on another stream; it must not wait for the device.  ready == NULL: wire_train_fwd_bwd.     */
typedef void (*wire_grad_ready_fn)(void* user, int first_tensor, int n_tensors);
int wire_train_fwd_bwd_hooked(void* stream, const wire_net_desc* d, const float* packed,
                              const float* coords, int64_t n, const float* target,
                              const int64_t* idx, int64_t first, float weight, float* y,
                              float* g_y, float* loss_out, float* rec, float* partial,
                              void* act, int64_t act_bytes, void* scratch,
                              int64_t scratch_bytes, void* const* grads_host,
                              wire_grad_ready_fn ready, void* user);

/* ---- per-layer path (ComplexGaborLayer.forward, modules/wire.py:88-93) -- */
/* x: [n][in] f32 when is_first else [n][in] c64; W: [out][in] f32/c64;
 * act_out [n][out] c64 (interleaved); lin_out (optional, may be NULL) receives the pre-activation
 * `lin` of modules/wire.py:89 in the dtype of W's product: f32 [n][out] if is_first else c64.
 * The GEMM runs on the family the tuning knobs select -- the same kernels as wire_mlp_fwd.
 * ws: scratch of wire_layer_ws_bytes(n, in, out).                          */
int64_t wire_layer_ws_bytes(int64_t n, int in_features, int out_features);
int wire_gabor_fwd(void* stream, const void* x, const void* W, const void* b,
                   float omega0, float scale0, int64_t n, int in_features,
                   int out_features, int is_first, void* lin_out,
                   void* act_out, void* ws, int64_t ws_bytes);
/* backward: g_act [n][out] c64 -> g_x (NULL when is_first), g_W, g_b.      */
int wire_gabor_bwd(void* stream, const void* g_act, const void* x,
                   const void* W, const void* b, float omega0, float scale0,
                   int64_t n, int in_features, int out_features, int is_first,
                   void* g_x, void* g_W, void* g_b, void* ws, int64_t ws_bytes);
/* gradient of a TRAINABLE omega_0 / scale_0 (ComplexGaborLayer(..., trainable=True), modules/wire.py:80-81):
 * out2 (device, 2 floats) = { dL/d omega_0, dL/d scale_0 } for the upstream gradient g_act; same operands and
 * workspace as wire_gabor_bwd.                                                                       */
int wire_gabor_hparam_grad(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                           float omega0, float scale0, int64_t n, int in_features, int out_features,
                           int is_first, float* out2, void* ws, int64_t ws_bytes);
/* final nn.Linear(K,O,cfloat) + .real (modules/wire.py:156-157,164-165)    */
int wire_final_fwd(void* stream, const void* z, const void* Wf, const void* bf,
                   int64_t n, int in_features, int out_features, float* y,
                   void* ws, int64_t ws_bytes);
int wire_final_bwd(void* stream, const float* g_y, const void* z,
                   const void* Wf, int64_t n, int in_features,
                   int out_features, void* g_z, void* g_Wf, void* g_bf,
                   void* ws, int64_t ws_bytes);

/* SineLayer / GaussLayer / ReLULayer .forward (modules/siren.py:48-49,
 * gauss.py:27-28, relu.py:28-29) and their backward on native f32 tensors:
 * x [n][in], W [out][in], b [out] -> act [n][out].  kind = WIRE_KIND_SIREN /
 * _GAUSS / _RELU; ws as for wire_gabor_fwd.                                   */
int wire_real_layer_fwd(void* stream, int kind, const float* x, const float* W,
                        const float* b, float omega0, float scale0, int64_t n,
                        int in_features, int out_features, float* act_out,
                        void* ws, int64_t ws_bytes);
int wire_real_layer_bwd(void* stream, int kind, const float* g_act, const float* x,
                        const float* W, const float* b, float omega0, float scale0,
                        int64_t n, int in_features, int out_features, float* g_x,
                        float* g_W, float* g_b, void* ws, int64_t ws_bytes);

/* ---- training-step glue (wire_image_denoise.py:142-157,
 *      wire_occupancy.py:137-158) ---------------------------------------- */
/* coords[r] = grid point of flat index idx[r] (idx NULL -> first + r).
 * 2-D (tz NULL): idx = i*W + j -> (tx[j], ty[i])   (wire_image_denoise.py:63-66)
 * 3-D: idx = (i*W + j)*T + k -> (tx[j], ty[i], tz[k]) (modules/utils.py:171-176)
 * tx/ty/tz are the caller's linspace tables (device).                      */
int wire_coords_from_index(void* stream, const int64_t* idx, int64_t first,
                           int64_t n, const float* tx, int W, const float* ty,
                           int H, const float* tz, int T, float* coords);
/* The per-epoch shuffle (torch.randperm(H*W) at wire_image_denoise.py:142, wire_occupancy.py:137) as a keyed
 * bijection pi_seed of [0, n_total) evaluated per position: idx_out[r] = pi_seed(first + r), r < count.  A rank
 * of a data-parallel job generates only the slice of the epoch's permutation it trains on (cost O(count),
 * independent of n_total and of the world size).  Integer arithmetic (a numpy twin in the
 * test suite reproduces it bit for bit); NOT the sequence torch.randperm draws.                                                  */
int wire_perm_indices(void* stream, uint64_t seed, int64_t n_total, int64_t first, int64_t count,
                      int64_t* idx_out);
/* loss = mean((y - target[idx])^2) over n*O elements scaled by `weight`
 * (= n/B for a shard of a global batch B; 1 for a whole batch);
 * g_y = weight * 2/(n*O) * (y - t).  loss_out[0] += is NOT used: it is
 * overwritten.  rec (optional): rec[idx[r]][:] = y[r][:]
 * (wire_image_denoise.py:150-153).  partial: scratch of >= 4096 floats.    */
int wire_mse_grad(void* stream, const float* y, const float* target,
                  const int64_t* idx, int64_t first, int64_t n, int O,
                  float weight, float* g_y, float* loss_out, float* rec,
                  float* partial);
/* Super-resolution loss of wire_SISR.py:151-161: rec = torch.nn.AvgPool2d(scale) of the full-grid
 * reconstruction y [H*W][O] (row n = i*W + j), loss = mean((gt_lr - rec)^2) over [H/scale][W/scale][O]
 * (floor: ragged borders are dropped and receive zero gradient).  Writes loss_out[0], g_y = dL/dy [H*W][O]
 * and, if rec_lr != NULL, the pooled image.  partial: >= 1024 floats of scratch.                       */
int wire_avgpool_mse_grad(void* stream, const float* y, int H, int W, int O, int scale,
                          const float* gt_lr, float* g_y, float* rec_lr, float* loss_out,
                          float* partial);

/* torch.optim.Adam single step over a flat fp32 buffer (complex tensors as
 * real pairs; wire_image_denoise.py:123-125).  step is 1-based.            */
int wire_adam_step_flat(void* stream, float* param, const float* grad,
                        float* exp_avg, float* exp_avg_sq, int64_t count,
                        float lr, float beta1, float beta2, float eps,
                        int64_t step);

/* Evaluation metrics without a device->host copy of the reconstruction
 * (the reference copies the whole image to the host every epoch,
 * wire_image_denoise.py:161-178, and binarises the volume on the host side,
 * modules/volutils.py:74-91).  out2 / partial: device; partial >= 2048 floats.
 *   mode 0: out2 = { sum (gt-rec)^2, max(gt) }  -> utils.psnr = 10 log10(max / (sum/count))
 *                                                   (modules/utils.py:67-82: max(x), not max^2)
 *   mode 1: out2 = { |rec>=thres & gt!=0|, |rec>=thres or gt!=0| }  -> IoU = out2[0] / out2[1]
 *           (rec is NOT binarised in place, unlike get_I_and_U)               */
int wire_eval_metric(void* stream, int mode, const float* rec, const float* gt,
                     int64_t count, float thres, float* out2, float* partial);

/* Best-reconstruction tracking of the drivers without a device->host copy per epoch
 * (wire_image_denoise.py:176-178: `if (mse_array[epoch] < best_mse) or (epoch == 0): best_mse = ...; best_img = imrec`;
 * wire_occupancy.py:170-172 with lossval): if force != 0 or metric[0] < best_metric[0], copy src[0..count) to dst
 * and metric[0] to best_metric[0]; updated (optional, device int) receives 1 / 0.  All pointers device.      */
int wire_track_best(void* stream, const float* metric, float* best_metric, int force, const float* src,
                    float* dst, int64_t count, int* updated);
/* CT forward operator of wire_ct.py:128-133 -- lin_inverse.radon (modules/lin_inverse.py:19-40): every angle
 * rotates the image (kornia.geometry.rotate of kornia 0.6.5: about ((W-1)/2, (H-1)/2), counter-clockwise degrees,
 * bilinear, zero padding, align_corners) and sums over the rows: img [H][W] -> sino [nangles][W].  wire_radon_bwd
 * is its adjoint, g_sino [nangles][W] -> g_img [H][W] (overwritten; float atomics: summation order, hence the last
 * bits, vary from run to run).                                                                              */
int wire_radon_fwd(void* stream, const float* img, const float* angles_deg, int H, int W, int nangles, float* sino);
int wire_radon_bwd(void* stream, const float* g_sino, const float* angles_deg, int H, int W, int nangles,
                   float* g_img);
/* torch.sigmoid of the dense occupancy query before the cube is written (modules/volutils.py:128-131)       */
int wire_sigmoid_inplace(void* stream, float* x, int64_t count);

/* ---- ComplexGaborLayer2D (modules/wire2d.py:21-67) on native tensors ---------------------
 * act = exp(j w0 lin) exp(-s0^2 (|lin|^2 + |sy|^2)),  lin = x W^T + b,  sy = x V^T + c (scale_orth).
 * Same conventions as wire_gabor_fwd / wire_gabor_bwd: x, act, g_act, g_x complex64 [n][features]
 * (is_first: x real float32 [n][in <= 4], W, b, V, c real, g_x unused); W, V [out][in], b, c [out];
 * the backward recomputes the forward from x.  g_x may be NULL.                                   */
int64_t wire_layer2d_ws_bytes(int64_t n, int in_features, int out_features);
int wire_gabor2d_fwd(void* stream, const void* x, const void* W, const void* b, const void* V,
                     const void* c, float omega0, float scale0, int64_t n, int in_features,
                     int out_features, int is_first, void* act_out, void* ws, int64_t ws_bytes);
int wire_gabor2d_bwd(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                     const void* V, const void* c, float omega0, float scale0, int64_t n,
                     int in_features, int out_features, int is_first, void* g_x, void* g_W, void* g_b,
                     void* g_V, void* g_c, void* ws, int64_t ws_bytes);

/* trainable omega_0 / scale_0 of ComplexGaborLayer2D (modules/wire2d.py:42-43 with trainable=True):
 * out2 (device, 2 floats) = { dL/d omega_0, dL/d scale_0 }; operands and workspace as wire_gabor2d_bwd. */
int wire_gabor2d_hparam_grad(void* stream, const void* g_act, const void* x, const void* W, const void* b,
                             const void* V, const void* c, float omega0, float scale0, int64_t n, int in_features,
                             int out_features, int is_first, float* out2, void* ws, int64_t ws_bytes);

/* ---- positional encoding (PosEncoding.forward, modules/relu.py:62-75) ----
 * out[n][D + 2 D F]: the raw coordinates, then for each frequency i < F and dimension j < D: sin(2^i pi c_j),
 * cos(2^i pi c_j) (the product 2^i pi c_j rounded to fp32 first, as the reference's python-float x tensor is).  The
 * whole-net path (wire_mlp_fwd with posenc_freqs > 0) runs the same kernel into its padded first-layer input.   */
int wire_posenc_fwd(void* stream, const float* coords, int64_t n, int D, int F, float* out);

/* ---- layout helpers ---------------------------------------------------- */
int wire_blocked_width(int K);   /* P = roundup(2K, 64) */
/* Float offset of the stored activations out_l (n rows of P floats -- blocked planar for wire / wire2d, P = roundup(K, 64)
 * plain features for siren / gauss / relu; l = 0..hidden_layers) inside an act buffer that wire_mlp_fwd / wire_train_fwd_bwd
 * filled for n rows with save_for_bwd = 1: the per-layer activations of modules/utils.py:246-252 without re-running the
 * layers.  (With recompute_out the last hidden layer of a fused training step is not stored, except for relu; with
 * split_out the inner hidden layers 1 .. hidden_layers - 1 hold fp16 pairs, see the knob -- set it to 0 to read them.) */
int64_t wire_act_out_offset(const wire_net_desc* d, int64_t n, int layer);
int wire_c64_to_blocked(void* stream, const void* src, int64_t n, int K, float* dst);
int wire_blocked_to_c64(void* stream, const float* src, int64_t n, int K, void* dst);

/* ---- tuning knobs ---------------------------------------------------------
 * "split_f16" (default 1; environment WIRE_SPLIT_F16; needs split_bf16 = 1 and the x3_h16 bits of the net kind): the
 *     hidden-layer GEMMs of batches of >= 4096 rows run as a TWO-way fp16 split on v_mfma_f32_16x16x32_f16
 *     (wire_gemmx2h.hip): x s = h + l with s a power of two that maps the operand tensor's max |value| to [2^14, 2^15),
 *     three partial products per fp32 product (h h + h l + l h), fp32 accumulate -- fp32-accurate (measured rms error
 *     0.64 - 0.78 x the fp32 MFMA chain's, tools/f16x2_numerics.hip) at half the matrix-core work of the 3 x bf16
 *     split.  The maxima are tracked on the device by each tensor's producer kernel (64 sharded atomicMax slots inside
 *     the act / scratch / packed buffers) and read by the consumer kernel: no host round trip.  "x2_amode" (default 2):
 *     how the forward / data-gradient GEMM reads its activation operand -- 2 = through LDS in whole 128-byte lines (8 rows
 *     per LDS-DMA instruction, bank swizzle on the source address), 1 = straight into fragment registers (16 rows x 64
 *     bytes per instruction), 0 = through LDS in such half-line pieces; bit-identical results.
 *     0 = the 3 x bf16 kernels below for every batch size.
 * "split_out" (default 1; environment WIRE_SPLIT_OUT; with split_f16): the activations out_l of the INNER hidden layers
 *     (1 <= l < hidden_layers) of a wire / wire2d / siren / gauss net are stored ALREADY SPLIT by the forward epilogue --
 *     per 4 consecutive columns the 16 bytes [h h h h | l l l l] (fp16) instead of 4 floats, scale 2^(15 - e) from the
 *     activation's a-priori bound exp(w0^2 / 4 s0^2) < 2^e (sine, Gaussian: 1) -- and read in that form by the next
 *     layer's forward GEMM and by the weight-gradient GEMM, which then spend no vector instructions on the split.  Used
 *     when the bound is <= 16 (w0 / s0 <= 3.33), recompute_out = 1 (nothing else reads out_l) and the widths have a
 *     2 x fp16 weight-gradient shape; otherwise, and always for out_0, out_L and relu, activations stay fp32 with the
 *     maximum tracked on the device.  Inside the act buffer only: every pointer of this interface still carries fp32.
 * "split_bf16" (default 1; environment WIRE_SPLIT_BF16): every GEMM of every net
 *     kind runs on the bf16 matrix cores with each fp32 operand split exactly into
 *     three bf16 terms (6 partial products, fp32 accumulate; fp32-accurate --
 *     wire_gemmx3.hip).  0 = the fp32-MFMA kernels below.
 * "complex_3m" (default 1; used when split_bf16 = 0): wire layers use the
 *     3-multiplication complex GEMMs on the fp32 MFMA (6 real flop per complex
 *     MAC); 0 = 4-multiplication real-expanded GEMMs.
 * "x3_tall" (default 1), "x3_tn_tall" (default 0): 256-row tiles in the split-bf16
 *     NT / TN kernels.   "nt_bk" (16 | 32): K-slab depth of the fp32 4M NT kernel.
 * "x3_h16" (default 15): the v_mfma_f32_16x16x32_bf16 edition (wire_gemmx3h.hip) of the NT GEMMs at M >= 4096:
 *     bit 0 wire forward, bit 1 wire data gradient, bit 2 siren / gauss / relu, bit 3 wire2d; 0 = the 32x32x16 kernels.
 * "x3_tn16" (default 1): the weight-gradient (TN) GEMM of the split-bf16 family runs its 256 x 256-tile
 *     v_mfma_f32_16x16x32_bf16 kernel when both padded widths are multiples of 256 (up to 256 row splits); 0 = the
 *     128 x 128 kernel.
 * "recompute_out" (default 1): on the 16x16x32 kernels the backward of a wire / wire2d / siren / gauss net evaluates out = act(lin) again
 *     (same lean form, same bits) instead of reading it back: data-gradient epilogues, the fused final stage of
 *     wire_train_fwd_bwd (whose last hidden layer then does not store out at all).  0 = read the stored activations.
 * "first_sums" (default 1): wire nets on the 16x16x32 kernels -- the epilogue of the last data-gradient GEMM forms the
 *     first layer's per-tile gradient sums itself instead of storing g_u for a separate reduction pass.
 * (The LDS-DMA 32x32x16 edition of round 2, "x3_glds", left the library: tools/wire_gemmx3g.hip, harness builds only.)
 * Buffer sizes (wire_packed_floats, wire_act_bytes, wire_bwd_scratch_bytes) and the layout of packed weights do not
 * depend on the knobs: wire_pack_params writes the weight image of every family, activations are 4-byte-per-element
 * blocked rows for all of them (fp32, or the fp16 pairs of split_out).  The knobs must NOT change between a forward and the backward that consumes its activation buffer: the
 * backward re-derives from them which kernel edition produced the activations (recompute_out: the lean forward form
 * whose bits it reproduces) and whether the forward filled the max-|value| slots the 2 x fp16 kernels scale by.     */
int wire_tune_set(const char* key, int value);
int wire_tune_get(const char* key);   /* "split_bf16" | "split_f16" | "split_out" | "x2_amode" | "complex_3m" | "x3_h16" | "x3_tn16" | "recompute_out" -> value; < 0 = error */

/* ---- profiling hooks (bench.py roofline) -------------------------------
 * When enabled, every launch of the hot kernels is bracketed by hipEvents on
 * the launch stream; wire_prof_read synchronises those events and returns
 * per-class totals.  Classes: 0 fwd GEMM, 1 dgrad GEMM, 2 wgrad GEMM,
 * 3 everything else.                                                       */
#define WIRE_PROF_CLASSES 4
int wire_prof_enable(int on);
int wire_prof_read(double* ms_total, int64_t* launches, double* flops_total);

#ifdef __cplusplus
}
#endif
#endif /* WIRE_HIP_H */
