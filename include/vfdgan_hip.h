/*
 * vfdgan_hip.h — C ABI of libvfdgan_hip.so, the MI355X (gfx950) device library behind the
 * video-GAN training step of umaionigiri/vfd_gan.
 *
 * The reference has no FFI: its boundary to the arithmetic is the torch.nn call sites on the hot
 * path (reference paths relative to /root/reference, cited per entry point below).  Each entry point
 * replaces the ATen/cuDNN kernel(s) behind one such call site.  All pointers are DEVICE pointers
 * unless noted; tensors are dense "channels-last" blocks
 *
 *      act[N][D][H][W][Cp]      Cp = VFD_CPAD(C) = C rounded up to a multiple of 8,
 *                               pad channels hold zeros (2-D nets use D = 1)
 *
 * in `dtype` VFD_F32 (float) or VFD_BF16 (bfloat16 bit patterns in uint16_t).  Every function is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream), allocates
 * nothing, synchronises nothing and is therefore hipGraph-capturable.  Return value: 0 on success,
 * a negative VFD_E* code otherwise (message via vfd_last_error()).  No global mutable state other
 * than the thread-local error string.
 */
#ifndef VFDGAN_HIP_H
#define VFDGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFD_ABI_VERSION 1

#define VFD_F32 0
#define VFD_BF16 1
#define VFD_FP8 2     /* OCP e4m3fn bytes, 16 channels per granule (CPAD16); convolution OPERANDS only (vfd_conv_forward_fp8) */

#define VFD_OK 0
#define VFD_EINVAL (-1)   /* bad argument (shape/dtype/alignment) */
#define VFD_ELAUNCH (-2)  /* hipLaunch / hipGetLastError failure */
#define VFD_ENOSPACE (-3) /* workspace too small */

#define VFD_CPAD(c) (((c) + 7) & ~7)

/* activation codes (fused epilogues and element-wise ops) */
#define VFD_ACT_NONE 0
#define VFD_ACT_LRELU 1   /* x>0 ? x : slope*x ; slope 0 = ReLU; slope may exceed 1 (anogan.py:91 uses 64) */
#define VFD_ACT_SIGMOID 2
#define VFD_ACT_TANH 3

int vfd_abi_version(void);
const char* vfd_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Layout / dtype conversion at the boundary.
 * The reference hands (N,C,T,H,W) float32 tensors to the nets (lib/train_gan.py:69,
 * models/mygannet.py:275-286); 2-D ganomaly takes (N,C,H,W) (models/ganomaly.py:444).
 * ---------------------------------------------------------------------------------------------- */
/* src float32 [N][C][S] (S = D*H*W)  ->  dst dtype [N][S][Cp], pad channels zeroed. */
int vfd_ncs_to_nsc(int dtype, const float* src, void* dst, int64_t N, int C, int64_t S, void* stream);
/* src dtype [N][S][Cp] -> dst float32 [N][C][S] (pad channels dropped).                          */
int vfd_nsc_to_ncs(int dtype, const void* src, float* dst, int64_t N, int C, int64_t S, void* stream);

/* The reference's reshapes between a feature vector and a block, in the compute dtype:
 *   x.view(N, C, D, H, W) of an (N, C*S) vector (models/anogan.py:76):  src [N][1][C*S] -> dst [N][S][CPAD(C)]
 *   x.view(N, -1) of a block (models/anogan.py:115):                    src [N][S][CPAD(C)] -> dst [N][1][C*S]
 * C*S must be a multiple of 8 (so the flat vector carries no channel padding).                     */
int vfd_unflatten(int dtype, const void* src, void* dst, int64_t N, int C, int64_t S, void* stream);
int vfd_flatten(int dtype, const void* src, void* dst, int64_t N, int C, int64_t S, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution family: nn.Conv3d / nn.ConvTranspose3d / nn.Conv2d / nn.ConvTranspose2d / nn.Linear
 *   models/anogan.py:44,51-52,56-57,64-65,69-70,85,88-89,96-97,101,108
 *   models/mygannet.py:52,134,176 ; models/spatiotempconv.py:49-50,59-60
 *   models/ganomaly.py:37,45,54,66,96,105,116,123
 * One implicit-GEMM kernel family on MFMA (bf16: v_mfma_f32_16x16x32_bf16, f32: v_mfma_f32_16x16x4_f32).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vfd_conv_desc {
  int32_t N;
  int32_t Di, Hi, Wi, Cin;  /* input block  [N][Di][Hi][Wi][CPAD(Cin)]  */
  int32_t Do, Ho, Wo, Cout; /* output block [N][Do][Ho][Wo][CPAD(Cout)] */
  int32_t kd, kh, kw;       /* filter taps  */
  int32_t sd, sh, sw;       /* stride       */
  int32_t pd, ph, pw;       /* zero padding */
  int32_t transposed;       /* 0: out[o] = sum_k in[o*s-p+k] w[k]  (nn.ConvNd)
                               1: out[o] = sum_k in[(o+p-k)/s] w[k] (nn.ConvTransposeNd; Do.. given by caller,
                                  which encodes output_padding)                                            */
  int32_t dtype;            /* VFD_F32 | VFD_BF16: element type of in, out and packed filter; accumulate f32 */
  int32_t act;              /* VFD_ACT_* fused into the epilogue (after bias)                               */
  float slope;              /* LRELU slope                                                                  */
} vfd_conv_desc;

/* Filter packing.  A torch filter is float32 w[A][B][T] (A,B channel dims, T = kd*kh*kw taps, row-major).
 * The kernel wants rows of GEMM-K:  packed[R][T][CPAD(Cc)]  (R = output channel of the GEMM, Cc = contracted).
 *   transpose_ab = 0 : R = A, Cc = B   packed[a][t][b] = w[a][b][t]   (Conv forward; ConvTranspose dgrad)
 *   transpose_ab = 1 : R = B, Cc = A   packed[b][t][a] = w[a][b][t]   (ConvTranspose forward; Conv dgrad)   */
int vfd_pack_filter(int dtype, const float* w, void* packed, int A, int B, int T, int transpose_ab,
                    void* stream);

/* Many vfd_pack_filter jobs in one launch (all filter copies of one optimiser, right after its step).  jobs_dev: device
 * array of njobs x 8 int64 {w, packed, A, B, T, transpose_ab, dtype, first_block}; first_block = running sum of
 * vfd_pack_filter_blocks() over the preceding jobs, total_blocks the sum over all.                                   */
int64_t vfd_pack_filter_blocks(int A, int B, int T, int transpose_ab);
int vfd_pack_filters(const int64_t* jobs_dev, int njobs, int64_t total_blocks, void* stream);

/* y = act(conv(x, packed) + bias).  `bias` float32[Cout] or NULL.
 * When stats != NULL (float64 [VFD_STATS_REPLICAS][2][CPAD(Cout)], pre-zeroed) the epilogue also accumulates
 * the per-channel sum and sum of squares of the pre-activation output (BatchNorm batch statistics,
 * models/spatiotempconv.py:51, models/mygannet.py:19, models/ganomaly.py:46,56,97,106).  The variance is later formed as
 * E[x^2] - mean^2, where float32 sums lose (|mean|/sigma)^2 digits: a workgroup sums SHIFTED values (x - c, c = the channel's
 * value in its first tile row) in float32 and adds the raw sums, formed in double, to one of the replica rows with one double
 * atomic per channel and sum; vfd_bn_stats_from_sums / vfd_bn_act_forward_sums fold the rows (in double). */
#define VFD_STATS_REPLICAS 8
int vfd_conv_forward(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                     double* stats, size_t stats_bytes, void* ws, size_t ws_bytes, void* stream);
/* Scratch bytes vfd_conv_forward wants for `d` (0 for most shapes).  Convolutions with few output pixels and a
 * long reduction (ganomaly Encoder final conv / NetD classifier: 512 pixels x K=25088) split K over workgroups
 * into float32 partial tiles in `ws` and fold them in a finish kernel; without `ws` they run unsplit.      */
int vfd_conv_workspace(const vfd_conv_desc* d, int want_stats, size_t* bytes);

/* ---- fp8 operands (BASELINE configs[4]: "fp8 weights/activations (CDNA4 fp8 MFMA)") --------------------------------
 * OCP e4m3fn, per-tensor CURRENT scaling: vfd_amax takes max|x| of the tensor into a device scalar (zeroing it first),
 * vfd_quantize_fp8 writes x_q = e4m3(x * 448 / amax) with 16 channels per granule ([rows][CPAD16(C)]; amax == NULL:
 * scale 1) and publishes the scale it used in *scale_out; vfd_pack_filter_fp8 does both for a filter (same index map as
 * vfd_pack_filter, channels padded to 16).  vfd_conv_forward_fp8 (d->dtype == VFD_FP8) = the implicit GEMM on
 * v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 FLOPs per clock): y (bf16) = act(acc / (*scale_x * *scale_w) + bias), optional
 * BatchNorm statistics as vfd_conv_forward.  The scales are read on the device, so a captured step follows them.
 * Serves the forward pass and, with the A/B-swapped packing, the data gradient (as for bf16).                      */
int vfd_amax(int dtype, const void* x, int64_t rows, int C, float* amax, void* stream);
int vfd_quantize_fp8(int dtype, const void* x, void* q, int64_t rows, int C, const float* amax, float* scale_out,
                     void* stream);
int vfd_pack_filter_fp8(const float* w, void* packed, int A, int B, int T, int transpose_ab, float* amax,
                        float* scale_out, void* stream);
int vfd_dequantize_fp8(const void* q, float* y, int64_t n, const float* scale, void* stream);
int vfd_conv_forward_fp8(const vfd_conv_desc* d, const void* x, const float* scale_x, const void* packed,
                         const float* scale_w, const float* bias, void* y, double* stats, size_t stats_bytes,
                         void* stream);

/* y = (conv(x, packed) + bias) * act'(mul_src), mul_src a tensor of y's shape and dtype holding the OUTPUT of an
 * activation (LeakyReLU / Sigmoid / Tanh, derivative taken from the output).  This is the data gradient of a layer
 * whose input was produced by a conv with a fused activation (nn.Sequential(Conv, LeakyReLU, Conv, ...),
 * models/ganomaly.py:39-46): the producer's activation gradient rides in this kernel's epilogue instead of a
 * separate read-read-write pass.  No statistics, no split-K, no workspace. */
int vfd_conv_forward_mul(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                         const void* mul_src, int mul_act, float mul_slope, void* stream);

/* The same hand-over for a producer that is BatchNorm + activation (nn.Sequential(Conv, BatchNorm, LeakyReLU, Conv, ...),
 * models/ganomaly.py:52-58,102-108; models/spatiotempconv.py:48-57): `d` is the data gradient of the consumer layer,
 * bn_x the BatchNorm's INPUT (y's shape and dtype), mean/rstd its batch statistics.  With xh = (bn_x - mean) * rstd the
 * epilogue stores g = conv(x, packed) * act'(gamma * xh + beta) and adds the per-channel sums of g and of g * xh into
 * sums (float32 [VFD_STATS_REPLICAS][2][CPAD(Cout)], pre-zeroed): BatchNorm's backward reduce pass, without its two
 * reads.  vfd_bn_backward_apply_sums finishes the BatchNorm backward from (g, sums).  bf16 with more than 32 output
 * channels only; no bias, no activation of its own, no split-K, no workspace. */
int vfd_conv_bn_backward_supported(const vfd_conv_desc* d);   /* possible AND worth it: Cout >= the threshold below */
int vfd_conv_set_bn_handover_min_channels(int c);              /* default: never (measured: no gain once the reduce pass overlaps the
                                                                  side-stream filter gradients); returns the previous value */
int vfd_conv_forward_bn_backward(const vfd_conv_desc* d, const void* x, const void* packed, void* y, const void* bn_x,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                 float slope, float* sums, size_t sums_bytes, void* stream);

/* Name of the kernel vfd_conv_forward() dispatches this layer to ("conv_igemm<bf16,256c_x_128p>", "conv_cin8<bf16>",
 * "convt_thin<bf16>", ...), NUL-terminated into buf[n]: what the profiler rows of bench.py and profiles/ are keyed
 * on, so that they follow the dispatch rules instead of restating them. */
int vfd_conv_kernel_name(const vfd_conv_desc* d, int want_stats, char* buf, size_t n);

/* Dispatch switch of the halo-tiled kernel (conv_halo.hip: unit-input-stride layers with <= 64 output channels, bf16):
 * 0 = default rules (layers with >= 512 work items), 1 = never (every layer goes to conv_igemm / conv_small: the A/B
 * baseline of the layer benchmarks and of the parity tests that compare the two kernels on identical inputs),
 * 2 = whenever the shape is eligible, however small (tests).  Returns the previous mode.  Process-wide; not meant
 * to be flipped while launches are in flight on other threads. */
int vfd_conv_set_halo_mode(int mode);

/* Pixel extent of the 16-wave 256-channel conv_igemm tile (bf16, more than 128 output channels), in 16-pixel sub-tiles:
 * 0 = chosen per layer (12..16, the count that minimises rounds x work per round on the device's CUs: conv_igemm.hip
 * pick_tile_sub), 12..16 = forced (tests / A-B timing; 16 = the fixed 256-pixel tile of rounds 1-2).  Returns the previous
 * setting.  Process-wide, like the switches above. */
int vfd_conv_set_tile_sub(int sub);

/* Filter gradient.  Computes, for the conv described by `d` (same desc as forward),
 *     dWp[r][t][c] = sum_{n,q} S[n,q][r] * G[n, q*s-p+t][c]
 * with (S,G) = (dy, x) for transposed = 0 and (x, dy) for transposed = 1, i.e. in the packed layout of
 * vfd_pack_filter(transpose_ab = 0) over the torch filter w[A][B][T] with A = channels of S, B = channels of G.
 * The reduction over pixels is split `nsplit` ways into float32 slabs ws[nsplit][A][T*CPAD(B)]; call
 * vfd_wgrad_workspace() for nsplit and the byte size, then vfd_wgrad_reduce() to fold the slabs into the
 * torch-layout gradient dw[A][B][T] (dw = beta*dw + sum).  Also returns the bias gradient when db != NULL:
 * db[c] = beta*db[c] + sum over pixels of dy[..][c].                                                       */
int vfd_wgrad_workspace(const vfd_conv_desc* d, int32_t* nsplit, size_t* bytes);
int vfd_conv_wgrad(const vfd_conv_desc* d, const void* x, const void* dy, void* ws, size_t ws_bytes,
                   void* stream);
int vfd_wgrad_reduce(const vfd_conv_desc* d, const void* ws, float* dw, float beta, void* stream);
/* ... and, in the same launch, db[Cout] += the fold of VFD_STATS_REPLICAS replica rows bias_rep[r * rep_stride + c]: the
 * layer's bias gradient = the column sums of its output gradient, left there by whoever produced that gradient — the
 * consuming BatchNorm's apply pass (vfd_bn_backward_apply_sums, rep_stride = CPAD(Cout)) or the consuming convolution's
 * data-gradient launch run with a statistics buffer (conv -> conv chains, models/anogan.py:51-52,85-86: the sum row of
 * [VFD_STATS_REPLICAS][2][CPAD(Cout)], rep_stride = 2 * CPAD(Cout), rep_f64 = 1: those rows are doubles; the BatchNorm's
 * rows are float32, rep_f64 = 0).                                                                                  */
int vfd_wgrad_reduce_bias(const vfd_conv_desc* d, const void* ws, float* dw, float beta, const void* bias_rep,
                          int rep_stride, int rep_f64, float* db, void* stream);
/* Dispatch switch of the halo-tiled filter-gradient kernel (conv_wgrad_halo.hip: stride-1 layers with a 3 x 3 in-plane
 * footprint, kd 1 or 3, >= 33 channels on both sides, bf16): 0 = default rules, 1 = never (conv_wgrad's per-tap
 * gather), 2 = whenever eligible (tests).  Returns the previous mode.  vfd_wgrad_workspace / vfd_conv_wgrad /
 * vfd_wgrad_reduce of one layer must see the same mode. */
int vfd_wgrad_set_halo_mode(int mode);
/* Name of the kernel vfd_conv_wgrad() dispatches this layer to ("conv_wgrad<bf16,64x256>", "conv_wgrad_halo<bf16>"). */
int vfd_wgrad_kernel_name(const vfd_conv_desc* d, char* buf, size_t n);
size_t vfd_bias_grad_workspace(int C);
int vfd_bias_grad(int dtype, const void* dy, float* db, int64_t rows, int C, float beta, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm (training mode) fused with the activation that follows it.
 *   nn.BatchNorm3d/2d/1d + LeakyReLU/ReLU: models/mygannet.py:19-20,109-110; models/spatiotempconv.py:51-52;
 *   models/anogan.py:45-46,53-54,58-59,66-67,86-87,90-91,98-99,102-103; models/ganomaly.py:46-49,56-59,97-100,106-109
 * x is [rows][Cp]; statistics are over rows (biased variance for normalisation, unbiased for running_var).
 * ---------------------------------------------------------------------------------------------- */
/* partial[blocks][3][Cp] workspace reduction -> mean[Cp], rstd[Cp]; updates running stats when non-NULL:
 * running = (1-momentum)*running + momentum*batch (unbiased var), and *num_batches_tracked += 1 (the counter torch's
 * BatchNorm keeps next to them; it rides in this launch instead of its own).  ws must hold vfd_bn_workspace() bytes. */
size_t vfd_bn_workspace(int64_t rows, int C);
int vfd_bn_stats(int dtype, const void* x, int64_t rows, int C, float eps, float momentum, float* mean,
                 float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws,
                 void* stream);
/* Same, from the conv epilogue's sum / sum-of-squares buffer (stats[VFD_STATS_REPLICAS][2][Cp]).         */
int vfd_bn_stats_from_sums(const double* stats, int64_t rows, int C, float eps, float momentum, float* mean,
                           float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           void* stream);
/* One more running-statistics update (and counter increment) from batch statistics already at hand: stands in for a
 * forward that the reference repeats on identical input and weights (models/ganomaly.py:485 evaluates netd(input) a
 * second time inside backward_g; only its BatchNorm side effect differs from re-using the first result).            */
int vfd_bn_running_update(const float* mean, const float* rstd, int64_t rows, int C, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream);
/* y = act((x-mean)*rstd*gamma + beta) */
int vfd_bn_act_forward(int dtype, const void* x, void* y, int64_t rows, int C, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, int act, float slope,
                       void* stream);
/* vfd_bn_stats_from_sums + vfd_bn_act_forward in ONE launch: every thread folds the replica rows of its 8 channels, the
 * first row of workgroups publishes mean / rstd (saved for backward), the running statistics and the counter.       */
int vfd_bn_act_forward_sums(int dtype, const void* x, void* y, int64_t rows, int C, const double* sums, float eps,
                            float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, const float* gamma, const float* beta, int act, float slope,
                            void* stream);
/* Backward: g = dy*act'(.), dgamma = sum g*xhat, dbeta = sum g,
 * dx = gamma*rstd*(g - dbeta/rows - xhat*dgamma/rows).  dgamma/dbeta (scratch, [C]) are OVERWRITTEN;
 * dgamma_acc/dbeta_acc (NULL or the parameters' gradient buffers) are ACCUMULATED into.                   */
int vfd_bn_act_backward(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C,
                        const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                        float slope, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc, void* ws,
                        void* stream);

/* BatchNorm -> activation -> AvgPool3d (kernel = stride = (pd,ph,pw), each 1 or 2) in one forward pass and two backward
 * launches, for an activation whose only consumer is the pool (models/anogan.py:84-105: NetD's three blocks;
 * models/mygannet.py:132-133,174-175: SDisc / TDisc): x is [N][D][H][W][Cp], y the POOLED tensor
 * [N][D/pd][H/ph][W/pw][Cp]; the full-resolution activation is never written, and the backward takes the POOLED gradient
 * `gpool` (dy = gpool / (pd*ph*pw) at each input voxel of the window) and writes dx at full resolution.  Statistics, sums,
 * dgamma / dbeta, colsum_acc as in vfd_bn_act_forward_sums / vfd_bn_act_backward_sums (over the INPUT voxels).
 * y_full / g_full (NULL or full-resolution tensors): the activation ALSO has a full-resolution consumer (a U-Net skip
 * connection, models/mygannet.py:74-94): the forward writes it as well, and the backward adds the gradient that arrived
 * for it: dy = g_full + gpool / (pd*ph*pw) — no pooling-backward pass, no gradient-sum pass.                        */
int vfd_bn_act_pool_forward_sums(int dtype, const void* x, void* y, int N, int D, int H, int W, int pd, int ph, int pw,
                                 int C, const double* sums,
                                 float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, const float* gamma,
                                 const float* beta, int act, float slope, void* y_full, void* stream);
int vfd_bn_act_pool_backward_sums(int dtype, const void* x, const void* gpool, void* dx, int N, int D, int H, int W,
                                  int pd, int ph, int pw, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  int act, float slope, float* sums, float* dgamma, float* dbeta, float* dgamma_acc,
                                  float* dbeta_acc, float* colsum_acc, const void* g_full, void* stream);

/* The apply pass alone, for a gradient that arrives as g = dy*act'(.) with sums[VFD_STATS_REPLICAS][2][Cp] = the
 * per-channel sums of g and g*xhat (vfd_conv_forward_bn_backward): dgamma/dbeta are published by the first row of
 * workgroups (OVERWRITTEN; *_acc ACCUMULATED into), dx as above.  One launch instead of three.
 * colsum_acc (NULL or float32 [VFD_STATS_REPLICAS][Cp], pre-zeroed, ACCUMULATED into with float atomics): the
 * per-channel sums of dx in replica rows, i.e. the bias gradient of the convolution that feeds this BatchNorm
 * (nn.Conv3d(bias=True) -> BatchNorm3d, models/anogan.py:44-70) without vfd_bias_grad's own pass over dx;
 * vfd_wgrad_reduce_bias folds the rows into the parameter's gradient.                                              */
int vfd_bn_backward_apply_sums(int dtype, const void* x, const void* g, void* dx, int64_t rows, int C,
                               const float* mean, const float* rstd, const float* gamma, const float* sums,
                               float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc, float* colsum_acc,
                               void* stream);

/* vfd_bn_act_backward in TWO launches: the reduce pass adds its workgroup partials into `sums` (float32
 * [VFD_STATS_REPLICAS][2][Cp], pre-zeroed; float atomics) and the apply pass folds them itself.                    */
int vfd_bn_act_backward_sums(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                             float slope, float* sums, float* dgamma, float* dbeta, float* dgamma_acc,
                             float* dbeta_acc, float* colsum_acc, void* stream);

/* Element-wise activation and its backward from the OUTPUT (all supported activations are invertible in
 * sign / expressible from y): dx = dy * act'(y).                                                          */
int vfd_act_forward(int dtype, const void* x, void* y, int64_t rows, int C, int act, float slope, void* stream);
int vfd_act_backward(int dtype, const void* y, const void* dy, void* dx, int64_t rows, int C, int act,
                     float slope, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling / resampling / channel plumbing.
 *   nn.AvgPool3d: models/mygannet.py:41,132-133,174-175; models/anogan.py:92,100,104
 *   nn.Upsample(scale 2, trilinear, align_corners=True): models/mygannet.py:50
 *   torch.cat(dim=1): models/mygannet.py:79,84,89,94 ; gray2rgb: lib/utils.py:91-92
 *   nn.Dropout(p): models/mygannet.py:49 ; models/anogan.py:50,55,63,68
 * ---------------------------------------------------------------------------------------------- */
int vfd_avgpool_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int kd, int kh,
                        int kw, void* stream);
int vfd_avgpool_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int kd,
                         int kh, int kw, void* stream);
int vfd_upsample2x_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, void* stream);
int vfd_upsample2x_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C,
                            void* stream);
/* nn.Upsample(scale_factor=(fd,fh,fw), mode='trilinear', align_corners=True), every factor 1 or 2: models/xception.py:84 uses
 * (1,2,2); (2,2,2) is vfd_upsample2x_*.  y [N][fd D][fh H][fw W][CPAD(C)]. */
int vfd_upsample_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int fd, int fh, int fw, void* stream);
int vfd_upsample_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int fd, int fh, int fw, void* stream);
/* nn.MaxPool3d(kernel, stride, padding) (models/xception.py:57: (1,3,3), (1,s,s), (0,1,1)); floor mode, <= 255 taps, torch's
 * first-maximum rule.  y and idx are [N][Do][Ho][Wo][CPAD(C)] (idx: uint8, the window-local index of each maximum, which the
 * backward — a deterministic gather — reads instead of recomputing the windows). */
int vfd_maxpool_forward(int dtype, const void* x, void* y, void* idx, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                        int sd, int sh, int sw, int pd, int ph, int pw, void* stream);
int vfd_maxpool_backward(int dtype, const void* dy, const void* idx, void* dx, int N, int D, int H, int W, int C, int kd, int kh,
                         int kw, int sd, int sh, int sw, int pd, int ph, int pw, void* stream);
/* y = a + b (+ c when non-null), n elements (a multiple of 8), summed in float32 and rounded once: the residual join
 * `x += skip` of models/xception.py:68, and the sum of the gradients returning to a tensor with several consumers. */
int vfd_add(int dtype, const void* a, const void* b, const void* c, void* y, int64_t n, void* stream);
/* Zero fill of `bytes` bytes at a 16-byte-aligned address (statistics / sum pools, the gradient arenas of
 * optimizer.zero_grad(), models/ganomaly.py:514,517): capture-safe like everything else here. */
int vfd_zero(void* p, size_t bytes, void* stream);
/* The scalar arithmetic of a step's loss terms on the device, n = 1..4 float32 scalars:
 *   *out = sum_i w_i * *t_i   (models/ganomaly.py:487-490 err_g, :511 err_d; models/mygannet.py:416-433), left to right in float32;
 * vfd_scale4 is its backward: out[i] = *g * w_i. */
int vfd_weighted_sum4(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1, float w2,
                      float w3, int n, float* out, void* stream);
int vfd_scale4(const float* g, float w0, float w1, float w2, float w3, int n, float* out, void* stream);
/* torch.cat([Upsample(x), skip], dim=1) in one pass (U-Net decoder joint, models/mygannet.py:78-94): x [N][D][H][W][Ca]
 * (Ca a multiple of 8), skip [N][2D][2H][2W][CPAD(Cb)], y [N][2D][2H][2W][Ca + CPAD(Cb)]; the up-sampled tensor is never
 * materialised.  Backward: dx from the first Ca channels of dcat read in place, dskip = the remaining channels.     */
int vfd_upsample2x_cat_forward(int dtype, const void* x, const void* skip, void* y, int N, int D, int H, int W, int Ca,
                               int Cb, void* stream);
int vfd_upsample2x_cat_backward(int dtype, const void* dcat, void* dx, void* dskip, int N, int D, int H, int W, int Ca,
                                int Cb, void* stream);
/* Evaluation sweep post-processing on float32 planes [planes][H][W] (a (N,1,T,H,W) mask, planes = N*T): optional
 * threshold (x > threshold ? 1 : 0, lib/utils.py:149-152) followed by the 5 x 5 morphological opening of
 * lib/utils.py:139-147 (cv2.morphologyEx(MORPH_OPEN, ones(5,5)) per frame; cv2's default border: outside pixels do not take
 * part).  `tmp` holds the eroded planes.  Replaces a device -> host -> cv2 -> device round trip per test batch
 * (models/mygannet.py:396-397, models/anogan.py:180-181). */
int vfd_morph_open5x5(const float* x, float* tmp, float* y, int64_t planes, int H, int W, float threshold, int binarize,
                      void* stream);
/* dst[rows][CPAD(Ca+Cb)] = concat(a[rows][CPAD(Ca)], b[rows][CPAD(Cb)]) ; split is the backward.         */
int vfd_concat_channels(int dtype, const void* a, const void* b, void* dst, int64_t rows, int Ca, int Cb,
                        void* stream);
int vfd_split_channels(int dtype, const void* src, void* a, void* b, int64_t rows, int Ca, int Cb,
                       void* stream);
/* dst[rows][CPAD(reps)] = src[rows][CPAD(1)] channel 0 broadcast to `reps` channels (gray2rgb).            */
int vfd_broadcast_channel(int dtype, const void* src, void* dst, int64_t rows, int reps, void* stream);
/* y = x * mask / (1-p), mask ~ Bernoulli(1-p) from a counter-based generator keyed by (seed, *step_dev, element).
 * mask (uint8 [n]) is written for the backward; pass mask_in != NULL to impose a mask (parity tests).
 * step_dev (NULL or a device int64 the caller increments once per training step) keeps the masks changing when a
 * captured hipGraph of the step is replayed with a constant `seed` argument.                              */
int vfd_dropout_forward(int dtype, const void* x, void* y, uint8_t* mask_out, const uint8_t* mask_in,
                        int64_t n, float p, uint64_t seed, const int64_t* step_dev, void* stream);
int vfd_dropout_backward(int dtype, const void* dy, void* dx, const uint8_t* mask, int64_t n, float p,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses (mean reductions) on [rows][CPAD(C)] blocks; pad channels are ignored.  `b` may be NULL, then the
 * target is the constant `bconst` (the ones / zeros labels of models/mygannet.py:258-261,
 * models/anogan.py:142-143, models/ganomaly.py:449-450).
 *   l2_loss        lib/utils.py:59-63     mean((a-b)^2)
 *   nn.L1Loss      models/ganomaly.py:438 mean(|a-b|)
 *   nn.BCELoss     models/mygannet.py:267, models/anogan.py:138, models/ganomaly.py:440 (log clamped at -100)
 *   weighted_bce   lib/utils.py:65-71     clamp(a,1e-8,1-1e-8); -mean(b log a + pw (1-b) log(1-a))
 * forward : *loss (ONE float32 on the device) is overwritten; deterministic two-stage reduction through ws.
 * backward: grad_a / grad_b (either may be NULL) = scale * (*gout) * dLoss/d{a,b}; gout is a DEVICE scalar
 *           (the incoming autograd gradient) or NULL (= 1), so no host sync sits between loss and backward.
 * ---------------------------------------------------------------------------------------------- */
#define VFD_LOSS_L2 0
#define VFD_LOSS_L1 1
#define VFD_LOSS_BCE 2
#define VFD_LOSS_WBCE 3
size_t vfd_loss_workspace(int64_t rows, int C);
int vfd_loss_forward(int kind, int dtype, const void* a, const void* b, float bconst, float* loss, int64_t rows,
                     int C, float pos_weight, void* ws, void* stream);
int vfd_loss_backward(int kind, int dtype, const void* a, const void* b, float bconst, const float* gout,
                      void* grad_a, void* grad_b, int64_t rows, int C, float scale, float pos_weight,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * optim.Adam (models/mygannet.py:270-273, models/anogan.py:139-140, models/ganomaly.py:455-456):
 * eps 1e-8, no weight decay, no amsgrad, bias-corrected; one launch over a flat float32 arena.
 * grad_scale multiplies the gradient first (1/world_size for data-parallel sums).
 * ---------------------------------------------------------------------------------------------- */
int vfd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, float grad_scale, void* stream);
/* Same update with the step counter on the device (*step_dev is incremented first; bc_dev = float[2] scratch for
 * the bias corrections), so that a captured hipGraph of the whole training step can be replayed.              */
int vfd_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, int32_t* step_dev, float* bc_dev, float grad_scale,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFDGAN_HIP_H */
