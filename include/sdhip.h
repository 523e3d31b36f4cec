/*
 * sdhip.h — C ABI of libsdhip.so, the MI355X (gfx950) hot path of the joint
 * segmentation + disparity network of
 * cuevhv/PMT_learning_for_semantic_segmentation_and_disparity.
 *
 * Conventions (every entry point):
 *   - plain C, no C++ / torch types; all pointers are BORROWED device pointers;
 *     the callee never allocates or frees caller-visible memory;
 *   - activations are NHWC ("channels last"): element (b,h,w,c) of a tensor with
 *     pixel stride ld (in elements, ld >= C) lives at ((b*H + h)*W + w)*ld + c;
 *     a tensor may therefore be a channel slice of a wider slab;
 *   - dtype: SDHIP_F32 (exact f32 path, f32-input MFMA / f32 FMA) or
 *     SDHIP_BF16 (bf16 storage, f32 accumulate, bf16 MFMA); statistics and loss
 *     values are always f32/f64;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; the
 *     caller passes torch.cuda.current_stream().cuda_stream) and capturable in a
 *     hipGraph: no allocation, no synchronisation, no host read-back inside;
 *   - returns 0 on success or a negative SDHIP_ERR_* code; sdhip_last_error()
 *     returns a thread-local message; nothing ever aborts the process.
 *
 * Each function cites the reference interface (file:line under the upstream
 * repository root) it replaces.
 */
#ifndef SDHIP_H_
#define SDHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDHIP_F32 0
#define SDHIP_BF16 1

#define SDHIP_OK 0
#define SDHIP_ERR_ARG (-1)     /* bad shape / dtype / alignment */
#define SDHIP_ERR_LAUNCH (-2)  /* hip launch failure */
#define SDHIP_ERR_UNSUPPORTED (-3)

/* ABI version of this header; bumped on any signature change. */
int sdhip_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* sdhip_last_error(void);
/* Diagnostic / tuning environment switches (SDHIP_CONV_GENERIC, SDHIP_TUNE_*, ... — DESIGN.md 7b) are read once when
 * the library is loaded (a set switch is announced on stderr) and never per launch; a test or tool that changes the
 * environment for an A/B run inside one process calls this to re-read them.  No reference counterpart. */
void sdhip_diag_reload(void);
/* End a stream capture that failed half way (an uncapturable call invalidated it and the caller left its capture block
 * by an exception): hipStreamEndCapture on `stream`, the partial graph destroyed, the sticky error cleared.  Returns 1 if a
 * capture was still open, 0 if none was, < 0 if the stream cannot be brought back.  Host-side recovery for the training
 * step's hipGraph (the reference has no graph capture; its counterpart is simply running the step eagerly). */
int sdhip_abort_capture(void* stream);
/* Node inventory of a captured step: graph = hipGraph_t; counts[0..3] = kernel, memset, memcpy, other nodes.  Returns the
 * total node count (< 0: error).  The library keeps a captured step to kernel nodes (sdhip_zero_async); tests assert it. */
int sdhip_graph_node_counts(void* graph, int* counts);

/* ---------------------------------------------------------------------------
 * Spatial correlation sampler  (third-party op `SpatialCorrelationSampler`,
 * constructed at models/dsnet_t2.py:1078-1087,129-133 and called at
 * models/dsnet_t2.py:1188-1193,1233-1234,221-223 with kernel_size=1, stride=1,
 * padding=0).
 *
 *   out[b,h,w,ph*PW+pw] = sum_c in1[b,h,w,c] * in2[b, h+(ph-PH/2)*dil, w+(pw-PW/2)*dil, c]
 *   (zero where the displaced pixel leaves the image; no normalisation).
 *
 * in1,in2: NHWC (B,H,W,C) with pixel stride ld_in; out: NHWC (B,H,W,PH*PW) with
 * pixel stride ld_out — i.e. the reference's (B,PH,PW,H,W) result stored with
 * the displacement as the fastest dimension.
 * ------------------------------------------------------------------------- */
int sdhip_corr_fwd(const void* in1, const void* in2, void* out,
                   int B, int H, int W, int C, int ld_in,
                   int PH, int PW, int dil_patch, int ld_out,
                   int dtype, void* stream);
/* Gradients w.r.t. both inputs given gout = dL/dout (same layout as out). */
int sdhip_corr_bwd(const void* in1, const void* in2, const void* gout,
                   void* gin1, void* gin2,
                   int B, int H, int W, int C, int ld_in,
                   int PH, int PW, int dil_patch, int ld_out,
                   int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Direct 2-D convolution (im2col-free, LDS-staged halo tiles, MFMA contraction).
 *
 * Replaces ATen/cuDNN under conv2dSame.forward (models/torch_model.py:268-281),
 * the stride-1 ConvTranspose2dSame.forward (models/torch_model.py:320-349), the
 * DenseNet nn.Conv2d layers (models/densenet.py:25-93,131-245) and ASPP
 * (models/aspp.py:7-32).  Weights are consumed in a packed layout produced by
 * sdhip_conv_pack_weights from the module's f32 parameter:
 *     packed[q][t][m][c] = W(m, k = q*CK + c, tap t or T-1-t),  zero padded,
 *     CK = 64 (bf16) / 32 (f32), m < roundup(M,16)
 * with (stride_m, stride_k, flip) selecting which parameter axis is the output
 * ("m") and which the reduction ("k") axis:
 *     Conv2d        weight (Cout,Cin,kh,kw): forward   M=Cout K=Cin sm=Cin*T sk=T    flip=0
 *                                            data grad M=Cin  K=Cout sm=T     sk=Cin*T flip=1
 *     ConvTranspose2d weight (Cin,Cout,kh,kw) (stride 1, run as a correlation):
 *                                            forward   M=Cout K=Cin sm=T     sk=Cout*T flip=1
 *                                            data grad M=Cin  K=Cout sm=Cout*T sk=T    flip=0
 * ------------------------------------------------------------------------- */
long sdhip_conv_packed_elems(int M, int K, int T, int dtype);
int sdhip_conv_pack_weights(const float* src, void* dst, int M, int K, int T,
                            long stride_m, long stride_k, int flip, int dtype, void* stream);
/* Inverse of the packing for the f32 gradient buffer written by sdhip_conv2d_wgrad:
 * grad(m,k,t) (+)= acc[k/CK][flip ? T-1-t : t][m][k%CK]. */
int sdhip_conv_unpack_wgrad(const float* acc, float* grad, int M, int K, int T,
                            long stride_m, long stride_k, int flip, int accumulate, int dtype, void* stream);

/* y[b,oh,ow,m] = act( bias[m] + sum_{t,k} W[m,k,t] * pro(x)[b, oh*stride + kh*dil - pad_t, ow*stride + kw*dil - pad_l, k] )
 *   pro(x) = x, or relu?(x*in_scale[g][k] + in_shift[g][k]) when in_scale != NULL (the BatchNorm+ReLU
 *            that precedes the conv, models/densenet.py:41-45,75-93); zero padding is applied AFTER pro;
 *   g      = b / (B/groups): statistics group of image b (left/right tower passes share a launch);
 *   stats  : if non-NULL, f64 [groups][2][stats_ld >= Cout] (stats_ld <= 0 means Cout; a slice of a wider
 *            statistics slab is addressed by offsetting the pointer), replicated stats_nrep times
 *            ([nrep][groups][2][stats_ld]; workgroups spread their atomics over the replicas because thousands
 *            of adds into one 256-byte line serialise; consumers sum the replicas); the kernel ADDS sum(y) and sum(y^2) over pixels
 *            (of the stored, rounded values) — the batch statistics of the BatchNorm that follows;
 *   act    : 0 none, 1 ReLU, 2 sigmoid;  accumulate != 0: y += result.
 * 3-D convolutions (nn.Conv3d / nn.ConvTranspose3d of models_psmnet/submodule.py:16-19, stackhourglass.py:10-50):
 * a volume is [B][D][H][W][C]; (D, Do, kd, sd, pad_d) describe the depth axis and the packed weights gain a leading
 * depth-tap dimension [kd][q][t][m][c] (pack each depth tap with sdhip_conv_pack_weights).  2-D: D=Do=kd=sd=1, pad_d=0.
 * (pad_t, pad_l) is the top/left padding; bottom/right padding is implied by (Ho, Wo)
 * (TF-"same" padding of models/torch_model.py:276-281 is asymmetric for stride 2). */
int sdhip_conv2d_fwd(const void* x, const void* wpacked, void* y,
                     const float* bias, const float* in_scale, const float* in_shift,
                     double* stats, int stats_ld, int stats_nrep,
                     int B, int H, int W, int Cin, int ldx,
                     int Ho, int Wo, int Cout, int ldy,
                     int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                     int D, int Do, int kd, int sd, int pad_d,
                     int in_relu, int groups, int act, int accumulate,
                     int dtype, void* stream);

/* y = conv(relu(BatchNorm_train(x)), wpacked) with the BatchNorm finalized INSIDE the launch — the norm2 -> relu2 -> conv2 of a
 * DenseNet layer (models/densenet.py:41-45,86-93) without a per-channel kernel between conv1's epilogue, which wrote the batch
 * statistics in_stats (f64 [in_stats_nrep <= 4][groups][2][in_stats_ld]: sum, sum of squares), and this launch, whose input
 * prologue needs scale = gamma / sqrt(var + eps), shift = beta - mean * scale: every workgroup derives them for its statistics
 * group (arithmetic of sdhip_bn_finalize), workgroup 0 also writes scale / shift / mean / invstd ([groups][Cin], read by the
 * backward pass) and updates running_mean / running_var (groups in order, momentum as nn.BatchNorm2d; NULL: not tracked).
 * out_stats: as `stats` of sdhip_conv2d_fwd.  Stride 1, no dilation, Cin <= 1024; f32 and bf16.
 * pend_stats (optional): the norm1 -> relu1 -> conv1 of a dense layer (models/densenet.py:41-45,75-85) reads the slab statistics
 * in_stats (one replica) of which the channels [pend_c0, pend_c0 + pend_n) — the previous layer's output — are still spread over
 * the replicas pend_stats (f64 [pend_nrep <= 4][groups][2][pend_ld], channel c - pend_c0): they are taken from there and
 * workgroup 0 folds them into in_stats on the way (what sdhip_bn_fold_finalize does in a launch of its own). */
int sdhip_conv2d_fwd_bnpro(const void* x, const void* wpacked, void* y, double* out_stats, int out_stats_ld, int out_stats_nrep,
                           double* in_stats, int in_stats_ld, int in_stats_nrep,
                           const double* pend_stats, int pend_ld, int pend_nrep, int pend_c0, int pend_n,
                           const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* scale_out, float* shift_out, float* mean_out,
                           float* invstd_out, double count, float eps, float momentum,
                           int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                           int kh, int kw, int pad_t, int pad_l, int groups, int dtype, void* stream);

/* y = conv(x, wpacked) + addend: a data gradient that lands on a tensor with a second consumer — x1 / x2 of Conv2DownUp
 * (models/dsnet_t2.py:80-117) feed the next convolution AND a skip add — written as the SUM of both contributions (addend =
 * the skip's gradient, [B][Ho][Wo] pixels of ldadd elements), summed in f32 and rounded once, instead of autograd running an
 * elementwise add over the map.  Stride 1, no dilation; served by the persistent 5x5 kernel only (bf16, Cin in {8..32, 64},
 * Cout <= 64, at least 192 tiles of 16x32 output pixels, 16-byte aligned pixels): SDHIP_ERR_UNSUPPORTED otherwise. */
int sdhip_conv2d_fwd_add(const void* x, const void* wpacked, void* y, const void* addend, int ldadd,
                         int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                         int kh, int kw, int pad_t, int pad_l, int dtype, void* stream);

/* The data gradient of a stride-1 convolution whose INPUT was relu(BatchNorm_train(u)) — the 3x3 convolution of a DenseNet
 * layer behind norm2 + relu2 (models/densenet.py:41-45,75-93): y = conv(x, wpacked) is dL/d relu(bn(u)) (x = the gradient
 * of the convolution's output, wpacked = its weights packed in data-gradient form), and the epilogue, which holds every
 * value of y once, also takes the two reductions the BatchNorm backward starts with, over the stored (rounded) values:
 *     sums[rep][g][0][c] += sum_pixels gm * u,   sums[rep][g][1][c] += sum_pixels gm,
 *     gm = y where u*scale[g][c] + shift[g][c] > 0, else 0
 * (f64 [sums_nrep][groups][2][sums_ld >= Cout], zeroed by the caller) — what sdhip_affine_act_bwd computes in a pass of its
 * own over y and u.  u: [B][Ho][Wo] pixels of ldu elements, Cout of them used.  addend (optional, as sdhip_conv2d_fwd_add):
 * y = conv + addend, sums taken over that total — the convbn + ReLU layers of Conv2DownUp (models/dsnet_t2.py:80-117), whose
 * output gradient is the next layer's data gradient plus a skip gradient.  bf16, Cout % 4 == 0; consumer:
 * sdhip_bn_bwd_apply_fin_d.  SDHIP_ERR_UNSUPPORTED when the shape has no kernel with this epilogue.
 * mode 1 ("apply", addend required and allowed to be y): the launch is the 1x1 data gradient of a DenseNet layer AND the first
 * phase of norm1's backward over it (models/densenet.py:41-45,75-93: conv1 <- relu1 <- norm1 <- concatenated features):
 *     y = addend + gm * scale[g][c],  gm = conv(x) where u*scale + shift > 0 else 0
 * — the masked, scaled gradient accumulated into the slab's gradient — and the two reductions go, as f32, to
 * ((float*)sums)[which][rep][g][c] (which = 0: sum gm*u, 1: sum gm; [2][sums_nrep][groups][Cout], zeroed by the caller): the
 * dscale / dshift replicas of sdhip_affine_act_bwd, so its consumers (sdhip_stats_fix_fin, sdhip_bn_finalize_bwd) are unchanged. */
int sdhip_conv2d_fwd_bnbwd(const void* x, const void* wpacked, void* y, double* sums, int sums_ld, int sums_nrep,
                           const void* u, int ldu, const float* scale, const float* shift, const void* addend, int ldadd,
                           int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                           int kh, int kw, int dil, int pad_t, int pad_l, int groups, int mode, int dtype, void* stream);

/* One sub-pixel phase of nn.ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1) (models_psmnet/stackhourglass.py:25-29;
 * 2-D: D = kd = 1): output voxel 2m + off of an axis receives input m with kernel tap 1 when off = 0, and inputs m, m+1
 * with taps 2, 0 when off = 1 — so each of the 8 phases (off_d, off_h, off_w) is a stride-1 correlation with a
 * (1+off_d) x (1+off_h) x (1+off_w) sub-kernel over the UN-stuffed input, 27 taps in all instead of 8 x 27 over a
 * zero-stuffed volume.  x: [B][D][H][W][Cin]; y: the [B][2D][2H][2W][Cout] output volume (pixel stride ldy), of which this
 * call writes the voxels (2d + off_d, 2h + off_h, 2w + off_w); wpacked: the phase's sub-kernel packed [kd][q][t][m][c]
 * with taps ordered input-offset-major (sdhip_conv_pack_weights per depth tap); stats: as sdhip_conv2d_fwd (the eight
 * calls add into the same sums).  bf16 / f32, 16-byte aligned pixels. */
int sdhip_conv2d_fwd_phase(const void* x, const void* wpacked, void* y, double* stats, int stats_ld, int stats_nrep,
                           int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int kh, int kw,
                           int D, int kd, int groups, int off_d, int off_h, int off_w, int dtype, void* stream);
/* prezeroed != 0: the caller already zeroed dw_packed / dbias (one arena memset per step instead of one per call).
 * dW (packed f32, zeroed here) = sum over pixels of dy (x) pro(x); dbias[m] = sum dy (optional). */
int sdhip_conv2d_wgrad(const void* x, const void* dy, float* dw_packed, float* dbias,
                       const float* in_scale, const float* in_shift,
                       int B, int H, int W, int Cin, int ldx,
                       int Ho, int Wo, int Cout, int lddy,
                       int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                       int D, int Do, int kd, int sd, int pad_d,
                       int in_relu, int groups, int prezeroed, int dtype, void* stream);
/* The weight gradients of n layers in as few launches as possible: layers that run on the same kernel instantiation share
 * ONE grid (<= 16 layers per grid; the layer table travels in the kernel arguments, so a captured step needs no device
 * table).  items: HOST array, fields as the arguments of sdhip_conv2d_wgrad; every dw_packed / dbias must already be zero
 * (prezeroed semantics).  Results equal n calls of sdhip_conv2d_wgrad up to the order of the f32 atomic adds.  Nothing in
 * a training step reads a weight gradient before the optimizer (torch_implementation.py:389,724), so the step queues its
 * ~200 per-layer launches and issues them here after the backward pass (ops.StepContext.join).
 * max_workgroups > 0: every grid stays within that many workgroups (one round), leaving CUs to the kernels of another
 * stream — the decoder's weight gradients run beside the latency-bound DenseNet backward chain that way; 0: fill the chip. */
typedef struct SdhipWgradItem {
  const void* x; const void* dy; float* dw_packed; float* dbias; const float* in_scale; const float* in_shift;
  int B, H, W, Cin, ldx, Ho, Wo, Cout, lddy, kh, kw, stride, dil, pad_t, pad_l, D, Do, kd, sd, pad_d, in_relu, groups;
} SdhipWgradItem;
int sdhip_conv2d_wgrad_group(const SdhipWgradItem* items, int n, int max_workgroups, int dtype, void* stream);

/* 1x1 convolution (+ bias, + activation) over the channel concatenation [x0 | x1] without materialising it; segment i is
 * read at pixel (h >> us_i, w >> us_i) of a (B, c_i, H >> us_i, W >> us_i) map, i.e. nearest-neighbour upsampled by
 * 2^us_i on the fly.  Replaces `conv1d_2(torch.cat((F.interpolate(y, scale_factor=8), xleft2), 1))` and the other
 * concat -> 1x1 -> ReLU sites of models/dsnet_t2.py:1206-1216,1262-1291,927-933 (the x8 map of :1211 is never written).
 * wpacked: sdhip_conv_pack_weights of the (Cout, c0 + c1, 1, 1) weight; c0 a multiple of 8; bf16 only. */
int sdhip_conv1x1_cat_fwd(const void* x0, int ld0, int c0, int us0, const void* x1, int ld1, int c1, int us1,
                          const void* wpacked, void* y, int ldy, const float* bias,
                          int B, int H, int W, int Cout, int act, int dtype, void* stream);
/* Pack many weights in one launch: desc = ndesc rows of 8 int64 {src ptr, dst ptr, M, K, T, stride_m, stride_k, flip}
 * in device memory (same layout rules as sdhip_conv_pack_weights). */
int sdhip_conv_pack_batch(const long* desc, int ndesc, int dtype, void* stream);
/* One launch unpacks (adds) every weight gradient of the step into the flat gradient buffer:
 * desc[i] = {acc ptr (f32, packed as written by sdhip_conv2d_wgrad), grad ptr, M, K, T, stride_m, stride_k, flip};
 * grad(m,k,t) += acc[k/CK][flip ? T-1-t : t][m][k%CK].  Replaces ~200 per-layer sdhip_conv_unpack_wgrad launches of a
 * training step (the optimizer reads the gradients only after the whole backward pass: util/torch_implementation.py:389,724). */
int sdhip_conv_unpack_batch(const long* desc, int ndesc, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * BatchNorm2d in training mode (+ ReLU / sigmoid / skip add), decomposed so the
 * statistics ride on the producing conv and the normalisation on the consumer.
 * Replaces nn.BatchNorm2d as used at models/dsnet_t2.py:16-117,
 * models/densenet.py:25-128, models/aspp.py:7-32; statistics semantics as in
 * sync_batchnorm/batchnorm.py:114-126.  Tensors are [npix][C] row views with a
 * pixel stride; rows of statistics group g are the g-th npix/groups rows.  Statistics tensors are
 * f64 [groups][2][stats_ld] (stats_ld <= 0: = C), so a channel slice of a wider slab can be addressed.
 * ------------------------------------------------------------------------- */
/* stats[g][0][c] += sum x, stats[g][1][c] += sum x^2 (f64). */
int sdhip_channel_stats(const void* x, int ldx, double* stats, int stats_ld, int stats_nrep, long npix, int C, int groups,
                        int zero_first, int dtype, void* stream);
/* Train (stats != NULL): mean = S1/count, var = S2/count - mean^2 (biased), scale = gamma*invstd,
 * shift = beta - mean*scale; running_mean/var updated in place group by group with `momentum`
 * (running_var unbiased).  Eval (stats == NULL): scale/shift from the running statistics. */
int sdhip_bn_finalize(const double* stats, int stats_ld, int stats_nrep, const float* gamma, const float* beta,
                      float* running_mean, float* running_var,
                      float* scale, float* shift, float* mean_out, float* invstd_out,
                      int C, int groups, double count, float eps, float momentum, void* stream);
/* (dscale,dshift)[g][c] -> dgamma[c], dbeta[c] and dstats[g][2][c] (gradient w.r.t. S1, S2; zero in eval). */
int sdhip_bn_finalize_bwd(const float* dscale, const float* dshift, int nrep, const float* gamma,
                          const float* mean, const float* invstd,
                          float* dgamma, float* dbeta, double* dstats, int stats_ld, int accumulate_flags /* bit0: dstats +=, bit1: dgamma/dbeta += */,
                          int C, int groups, double count, int train, void* stream);
/* sdhip_bn_finalize_bwd (train mode, dstats accumulated) of a BatchNorm over the first Cf channels of a DenseNet slab fused
 * with sdhip_stats_fix of the 32 slab channels [cs, cs + 32) that the next layer of the backward walk needs:
 * gout = gin + dS[g][0][c] + 2 x dS[g][1][c] with dS INCLUDING this call's contribution for those channels (which is not
 * written back: nothing reads it again).  dscale/dshift: replicas [nrep][groups][Cf] of sdhip_affine_act_bwd;
 * accumulate_params: dgamma/dbeta += (flat gradient buffer) instead of =.  Rows 16-byte aligned. */
int sdhip_stats_fix_fin(const void* gin, int ldgi, const void* x, int ldx, void* gout, int ldgo, long npix,
                        double* dS, int ldc, int cs, const float* dscale, const float* dshift, int nrep,
                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        int accumulate_params, float param_scale, int Cf, int groups, double count, int dtype, void* stream);
/* y = act(x*scale[g][c] + shift[g][c]) (+ res). scale/shift may be NULL (identity). */
int sdhip_affine_act(const void* x, int ldx, void* y, int ldy, const void* res, int ldr,
                     const float* scale, const float* shift, long npix, int C, int groups, int act,
                     int dtype, void* stream);
/* act: 0 none, 1 ReLU, 2 sigmoid, 4 sigmoid with x holding the sigmoid OUTPUT.
 * dscale/dshift are replicated [nrep][groups][C] (see stats_nrep above; sdhip_bn_finalize_bwd sums them).
 * gx (+)= gy*act'*scale (if gx != NULL; += when accumulate != 0); dscale = sum gy*act'*x, dshift = sum gy*act' (if non-NULL; zeroed here). */
int sdhip_affine_act_bwd(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                         const float* scale, const float* shift, float* dscale, float* dshift, int nrep,
                         long npix, int C, int groups, int act, int accumulate, int prezeroed, int dtype, void* stream);
/* out[row][c] += sum_r ws[r][row][c] over the 2*groups statistics rows (folds conv-epilogue replicas into a slab). */
int sdhip_stats_replica_sum(const double* ws, double* out, int nrep, int groups, int C, int ldw, int ldo, void* stream);
/* sdhip_stats_replica_sum of the Cn newest channels (replicas ws[nrep][groups][2][ldw]) into the statistics slab
 * S[groups][2][ldc] at channel c_new0, followed by sdhip_bn_finalize of the first C >= c_new0 + Cn channels from S — one
 * launch instead of two on the DenseNet's layer-to-layer chain (models/densenet.py:41-45: norm1 of layer l+1 reads every
 * channel up to layer l's output). */
int sdhip_bn_fold_finalize(const double* ws, int nrep, int ldw, int c_new0, int Cn, double* S, int ldc,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float* scale, float* shift, float* mean_out, float* invstd_out, int C, int groups,
                           double count, float eps, float momentum, void* stream);
/* gout = gin + dstats[g][0][c] + 2*x*dstats[g][1][c]: the gradient that flows through the batch statistics. */
int sdhip_stats_fix(const void* gin, int ldgi, const void* x, int ldx, void* gout, int ldgo,
                    const double* dstats, int stats_ld, long npix, int C, int groups, int dtype, void* stream);
/* Two-phase BatchNorm backward, second phase: gx = gy * act'(x*scale+shift) * scale + dstats[g][0] + 2*x*dstats[g][1] in one
 * pass (first phase: sdhip_affine_act_bwd with gx == NULL, then sdhip_bn_finalize_bwd) — the backward of
 * nn.BatchNorm2d(+ReLU) in train mode, models/dsnet_t2.py:16-117.  act: 0 none, 1 relu, 2 sigmoid. */
int sdhip_bn_bwd_apply(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                       const float* scale, const float* shift, const double* dstats, int stats_ld,
                       long npix, int C, int groups, int act, int dtype, void* stream);
/* Consumer-side finalize (no separate per-channel launch between a reduction and the pass that uses it):
 * affine_act_bn   : y = act(BatchNorm_train(x)) (+ res) straight from the statistics of x (as written by the conv epilogue /
 *                   sdhip_channel_stats); also writes scale/shift/mean/invstd [G][C] for the backward pass and updates
 *                   running_mean / running_var (groups applied in order, momentum as nn.BatchNorm2d) — the forward of
 *                   convbn/deconvbn + ReLU (+ skip add), models/dsnet_t2.py:16-117.
 * bn_bwd_apply_fin: sdhip_bn_bwd_apply with dstats derived in the kernel from the (dscale, dshift) replica sums of
 *                   sdhip_affine_act_bwd; writes (accumulate_params: adds to) dgamma / dbeta summed over groups. */
int sdhip_affine_act_bn(const void* x, int ldx, void* y, int ldy, const void* res, int ldr,
                        const double* stats, int stats_ld, int nrep, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float* scale, float* shift, float* mean_out, float* invstd_out,
                        long npix, int C, int groups, double count, float eps, float momentum, int act, int dtype, void* stream);
int sdhip_bn_bwd_apply_fin(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                           const float* scale, const float* shift, const float* dscale, const float* dshift, int nrep,
                           const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                           int accumulate_params, float param_scale, long npix, int C, int groups, double count, int act, int dtype, void* stream);
/* bn_bwd_apply_fin with the two reductions supplied as f64 [nrep][groups][2][C] = (sum(gm*x), sum(gm)) — the layout in
 * which the epilogue of sdhip_conv2d_fwd_bnbwd adds them. */
int sdhip_bn_bwd_apply_fin_d(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                             const float* scale, const float* shift, const double* sums, int nrep,
                             const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                             int accumulate_params, float param_scale, long npix, int C, int groups, double count, int act, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Pooling / resizing / broadcast product on NHWC tensors (HBM bound).
 * Replace nn.MaxPool2d(3,2,1) (models/densenet.py:158), F.avg_pool2d / nn.AvgPool2d(p,p)
 * (models/densenet.py:232; models/dsnet_t2.py:1983-2022), F.interpolate nearest / bilinear
 * (models/dsnet_t2.py:927-936,1204-1275,2037-2081; models/aspp.py:88) and the attention
 * product s2_d*at_s (models/dsnet_t2.py:1258).  Index maths follow ATen's CPU kernels.
 * ------------------------------------------------------------------------- */
/* 3x3 / stride 2 / pad 1 max pool; idx (u8, [B*Ho*Wo][C]) records the winning tap (first max wins). */
int sdhip_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, unsigned char* idx,
                         int B, int H, int W, int C, int dtype, void* stream);
int sdhip_maxpool3s2_bwd(const void* gy, int ldg, const unsigned char* idx, void* gx, int ldgx,
                         int B, int H, int W, int C, int dtype, void* stream);
/* k x k / stride k average pool, no padding, floor mode (output H/k x W/k). */
int sdhip_avgpool_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int k,
                      int dtype, void* stream);
int sdhip_avgpool_bwd(const void* gy, int ldg, void* gx, int ldgx, int B, int H, int W, int C, int k,
                      int dtype, void* stream);
/* mode 0 nearest, 1 bilinear align_corners=False, 2 bilinear align_corners=True.
 * scale_h/scale_w > 0 override in/out (F.interpolate(scale_factor=f) uses 1/f). */
int sdhip_resize_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                     int mode, float scale_h, float scale_w, int dtype, void* stream);
/* tmp: caller-provided f32 workspace of B*Ho*W*C elements. */
int sdhip_resize_bwd(const void* gy, int ldg, void* gx, int ldgx, float* tmp, int B, int H, int W, int C,
                     int Ho, int Wo, int mode, float scale_h, float scale_w, int dtype, void* stream);
/* y[p,c] = a[p,c] * m[p] and its gradients (ga = g*m, gm[p] = sum_c g*a). */
int sdhip_mul_bcast_fwd(const void* a, int lda, const void* m, int ldm, void* y, int ldy, long npix, int C,
                        int dtype, void* stream);
int sdhip_mul_bcast_bwd(const void* g, int ldg, const void* a, int lda, const void* m, int ldm,
                        void* ga, int ldga, void* gm, int ldgm, long npix, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Tail of the training step.
 * ------------------------------------------------------------------------- */
/* Adam over ONE flat f32 buffer (torch.optim.Adam semantics, torch_implementation.py:718-724: lr 0.0015,
 * eps 1e-7).  beta_pow = {beta1^t, beta2^t} lives on the device and is advanced by the call, so the step can
 * be replayed from a hipGraph.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int sdhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* beta_pow,
                    long n, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    void* stream);
/* loss += weight * mean_p sum_c -target[p,c]*log_softmax(logits[p,:])_c  and (optionally) its gradient w.r.t. the
 * logits: categoricalCrossEntropy(F.log_softmax(y,1), gt) of util/utilTorchLoss.py:373-378.  target is f32. */
int sdhip_ce_loss(const void* logits, int ldy, const float* target, int ldt, void* grad, int ldg, double* loss,
                  long npix, int C, float weight, int dtype, void* stream);
/* loss += weight * mean |pred - target| (nn.L1Loss, losses/multiLosses.py:141) and its gradient.  mask_nonpositive != 0:
 * elements whose target is <= 0 contribute zero but stay in the mean — L1(pred*zeros, disp*zeros) with zeros = disp > 0,
 * the cityscapes / kitti rule of losses/multiLosses.py:134-141. */
int sdhip_l1_loss(const void* pred, const float* target, void* grad, double* loss, long n, float weight,
                  int mask_nonpositive, int dtype, void* stream);
/* y = x * mask / (1-p), mask from a counter-based hash of (*seed, layer_id, element index): applying the same call to
 * the gradient is the backward pass (nn.Dropout(0.5) of models/aspp.py:79,95).  x/y: dense buffers of n elements. */
int sdhip_dropout(const void* x, void* y, const long* seed, long layer_id, long n, float p, int dtype, void* stream);
/* Lovasz-softmax (util/lovasz_losses.py:153-199: classes='present', per_image=False) on
 * softmax(logits) with labels = argmax(target one-hot), as called at losses/multiLosses.py:70-72.
 * ignore_void != 0 (the cityscapes / kitti rule): a pixel whose target row has no positive entry is void and removed
 * from the loss (`ignore=19` with the 20th one-hot channel dropped, losses/multiLosses.py:19-21; flatten_probas,
 * util/lovasz_losses.py:202-216).  ignore_void == 0 (roses / garden, `ignore=None`, losses/multiLosses.py:11-17): such a
 * pixel has label argmax = class 0 and counts.
 * loss += weight * mean_{present c} dot(sort_desc |fg_c - p_c|, lovasz_grad(fg sorted)); grad (if non-NULL) is
 * ACCUMULATED (+=) with the gradient w.r.t. the logits.  workspace: sdhip_lovasz_workspace_bytes(npix, C) bytes. */
long sdhip_lovasz_workspace_bytes(long npix, int C);
int sdhip_lovasz_softmax(const void* logits, int ldy, const float* target, int ldt, void* grad, int ldg,
                         double* loss, long npix, int C, float weight, void* workspace, long workspace_bytes,
                         int ignore_void, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * PSMNet pieces (models_psmnet/stackhourglass.py:110-119,138-155; submodule.py:56-64).
 * ------------------------------------------------------------------------- */
/* scatter != 0: dst[n, d*sd, h*s, w*s, :] = src[n,d,h,w,:] into a zeroed (here) dense tensor of extent
 * ((D-1)*sd+1, (H-1)*s+1, (W-1)*s+1); scatter == 0: the gather back (dst dense-small, src stuffed).
 * Stride-s transposed convolutions and strided data gradients run as stride-1 convolutions over the stuffed tensor. */
int sdhip_stuff(const void* src, int ld_src, void* dst, int ld_dst, int N, int D, int H, int W, int C,
                int sd, int s, int scatter, int dtype, void* stream);
/* vol[b,i,h,w,0:C] = left[b,h,w,:], vol[b,i,h,w,C:2C] = right[b,h,w-i,:] for w >= i, else 0 (vol dense [B][D][H][W][2C]). */
int sdhip_cost_volume_fwd(const void* left, const void* right, int ld, void* vol, int B, int D, int H, int W, int C,
                          int dtype, void* stream);
int sdhip_cost_volume_bwd(const void* gvol, void* gleft, void* gright, int ld, int B, int D, int H, int W, int C,
                          int dtype, void* stream);
/* pred[b,h,w] = sum_d softmax_d(trilinear_upsample(cost)[b,d,h,w]) * d, cost [B][D4][H4][W4] -> (Dout,H,W),
 * align_corners=False (models_psmnet/stackhourglass.py:138-155, submodule.py:56-64); the (B,Dout,H,W) tensor is never
 * materialised.  stats (optional, only with Dout = 4 * D4): f32 [B*H*W][2] = (log-sum-exp, pred) per pixel, which spares the
 * backward pass the two softmax passes.  bwd: workspace of sdhip_softargmin_bwd_workspace_floats() floats; with
 * Dout = 4 * D4 it holds the per-pixel gradients of the D4 levels ([B][D4][H][W]), gathered into gcost without atomics. */
int sdhip_softargmin_fwd(const void* cost, void* pred, float* stats, int B, int D4, int H4, int W4, int Dout, int H, int W,
                         int dtype, void* stream);
long sdhip_softargmin_bwd_workspace_floats(int B, int D4, int H4, int W4, int Dout, int H, int W);
int sdhip_softargmin_bwd(const void* cost, const void* gpred, const float* stats, void* gcost, float* workspace,
                         long workspace_floats, int B, int D4, int H4, int W4, int Dout, int H, int W, int dtype, void* stream);
/* y = log_softmax(x) over the channel axis (F.log_softmax(.., dim=1), models/dsnet_t2.py:216,270) and its backward
 * gx = gy - exp(y) * sum_c gy. */
int sdhip_log_softmax_fwd(const void* x, int ldx, void* y, int ldy, long npix, int C, int dtype, void* stream);
int sdhip_log_softmax_bwd(const void* gy, int ldg, const void* y, int ldy, void* gx, int ldgx, long npix, int C,
                          int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Height-driven attention (HANet_Conv.forward, models_hanet/HANet.py:74-128) — the data-sized pieces; the tiny
 * (B, C, L) convolutions / BatchNorm1d / sigmoid / resize in between run through the conv / BN / resize entry points
 * on (B, C, L, 1) images.
 * rowpool_max: nn.AdaptiveMaxPool2d((OH, 1)) (HANet.py:50-55,84): y[b,i,c] = max over rows [floor(i*H/OH),
 *   ceil((i+1)*H/OH)) and all columns of x[b,:,:,c]; idx[b,i,c] = h*W+w of the first maximum (int32, dense [B][OH][C]).
 * mul_rows: torch.mul(out, attention.unsqueeze(3)) (HANet.py:112): y[b,h,w,c] = a[b,h,w,c] * att[b,h,c].
 * dropout_channels: nn.Dropout2d(p) on a (B, C, L) tensor (HANet.py:27-28,90-91): whole (b,c) rows dropped, rest scaled by
 *   1/(1-p); mask from a counter hash of (*seed, layer_id, b*C+c) — the same call on the gradient is the backward.
 * ------------------------------------------------------------------------- */
int sdhip_rowpool_max_fwd(const void* x, int ldx, void* y, int ldy, int* idx, int B, int H, int W, int C, int OH,
                          int dtype, void* stream);
int sdhip_rowpool_max_bwd(const void* gy, int ldg, const int* idx, void* gx, int ldgx, int B, int H, int W, int C, int OH,
                          int dtype, void* stream);
int sdhip_mul_rows_fwd(const void* a, int lda, const void* att, int ldt, void* y, int ldy, int B, int H, int W, int C,
                       int dtype, void* stream);
int sdhip_mul_rows_bwd(const void* g, int ldg, const void* a, int lda, const void* att, int ldt, void* ga, int ldga,
                       void* gatt, int ldgt, int B, int H, int W, int C, int dtype, void* stream);
int sdhip_dropout_channels(const void* x, int ldx, void* y, int ldy, const long* seed, long layer_id, int B, int L, int C,
                           float p, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * On-device step metrics: one pass over the step's outputs replaces the host-side block
 * of losses/multiLosses.py:116-125,146-154 — SegAccuracyNp (util/utilTorchLoss.py:221-236), GetSegMetricsNp
 * (:251-303), unnormalizedErrorNP (:363-370) and GetDispMetricsNp (:318-343) — which copies three full-resolution
 * maps to the host and runs numpy / sklearn on them every step.
 *   seg         (B*hw pixels, L logits, pixel stride lds) of `dtype`, or NULL
 *   seg_target  one-hot f32, Ct = L or L+1 channels (channel L = "ignore", mask `gt_seg != labels`), pixel stride ldt
 *   disp        dense B*hw disparities of `dtype`, or NULL;  disp_target dense f32
 *   mask_invalid  1: multiply prediction and target by (target > 0) first (`zeros` of lossDisp_fn for kitti/cityscapes)
 * ACCUMULATES (the caller zeroes once per reporting interval) into `nrep` replicas — workgroup g adds into replica
 * g % nrep, so that the closing atomics do not all serialise on one cache line; the caller sums the replicas.  Replica r
 * starts at counts + r*rep_stride (rep_stride >= L*L + SDHIP_METRIC_COUNTS) and sums + r*SDHIP_METRIC_SUM_STRIDE:
 *   counts[0 .. L*L)   confusion matrix, counts[L*gt + pred] over the pixels whose gt class != L
 *   counts[L*L + k], k < SDHIP_METRIC_COUNTS:
 *     0..3  image 0: TP, FP, FN, TN of (logit[1] > 0) against target[1]   (precision / recall / f1, average="micro")
 *     4,5   image 0: matching values and size of the branch mask (target[1] == 1 or raw logit[1] == 1)   (Bf1)
 *     6,7   whole batch: #(|pred - gt| * max_disp > 3 and gt > 0), #(gt > 0)                          (err, val_pxl)
 *     8     image 0: #(target[1] == 1)
 *     9     image 0: pixels scored by the disparity group (denominator of the RMSE / SqRel means)
 *   sums[k], k < SDHIP_METRIC_SUMS (f64): image 0: sum (gt-pred)^2, sum (gt-pred)^2/gt, and the same two over the
 *     pixels with target[1] == 1 (dispRMSE, dispSqRel, branch variants).
 * ------------------------------------------------------------------------- */
#define SDHIP_METRIC_COUNTS 10
#define SDHIP_METRIC_SUMS 4
#define SDHIP_METRIC_SUM_STRIDE 32   /* doubles per replica of `sums` (one 256-byte line) */
int sdhip_step_metrics(const void* seg, int lds, const float* seg_target, int ldt, int Ct, const void* disp,
                       const float* disp_target, long* counts, double* sums, int nrep, int rep_stride, int B, long hw,
                       int L, float max_disp, int mask_invalid, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * On-device sample preparation: everything CustomDataset.__getitem__ does after the PNG
 * containers are decoded (util/utilTorchDataLoader.py:133-274) — PFM decode + flipud (util/utilIOPfm.py:66-101),
 * depth -> disparity (:173-181), output activation (:188-197), one-hot target (:199-211; cityscapes table
 * util/utilCityscape.py:173-186), crop window (RandomCrop :435-463), (x/255 - mean)/std in float64 (:247-248) and the
 * HWC -> tensor conversion of ToTensor (:608-630) — in one launch per sample, writing one batch slot.
 *   left,right  uint8 H x W x img_cs (first 3 channels used), row pitch img_pitch bytes          -> out_left/out_right:
 *               out_h x out_w pixels of `dtype`, pixel stride ld_img (>= 3), value ((v/255 - mean[c]) / std[c])
 *   seg         uint8 H x W x seg_cs; channel seg_channel is classified by seg_mode:
 *               SDHIP_SEG_THRESHOLD (roses: class = v > seg_threshold), SDHIP_SEG_ID_PLUS_ONE (garden: class = v - 1),
 *               SDHIP_SEG_LUT (class = lut[v], a 256-byte device table; 255 there must already be mapped to n_seg - 1)
 *               -> out_seg f32 one-hot, n_seg channels, pixel stride ld_seg
 *   depth       SDHIP_DEPTH_PFM: the PFM payload (f32 rows, bottom row first, depth_pitch bytes apart, byte-swapped when
 *               depth_big_endian): disparity = depth > 0 ? fb / depth : 0;  SDHIP_DEPTH_U16: uint16 rows, disparity = v/256
 *               then activation: LINEAR none; SIGMOID min(d,max_d)/max_d; TANH d' = min(d,max_d), d' != 0 ? 2d'/max_d-1 : -1
 *               -> out_disp f32 dense out_h x out_w
 * row_roll (sliceandSwitch, RandomCrop :455-467): output row r is crop row (r + row_roll) mod out_h; 0 = off.
 * Any of the three groups may be NULL.  All pointers are device pointers; mean/std are HOST float[3].
 * ------------------------------------------------------------------------- */
#define SDHIP_SEG_THRESHOLD 0
#define SDHIP_SEG_ID_PLUS_ONE 1
#define SDHIP_SEG_LUT 2
#define SDHIP_DEPTH_PFM 0
#define SDHIP_DEPTH_U16 1
#define SDHIP_ACT_LINEAR 0
#define SDHIP_ACT_SIGMOID 1
#define SDHIP_ACT_TANH 2
int sdhip_prepare_sample(const unsigned char* left, const unsigned char* right, long img_pitch, int img_cs,
                         const unsigned char* seg, long seg_pitch, int seg_cs, int seg_channel, int seg_mode,
                         int seg_threshold, const unsigned char* lut, const void* depth, long depth_pitch,
                         int depth_mode, int depth_big_endian, int H, int W, int crop_top, int crop_left,
                         int out_h, int out_w, int row_roll, float fb, float max_d, int activation, const float* mean,
                         const float* stdv, void* out_left, void* out_right, int ld_img, float* out_seg,
                         int ld_seg, int n_seg, float* out_disp, int dtype, void* stream);

/* Horizontal flip of ONE prepared stereo sample, in place — RandomCrop(flipHorizontal=True) of
 * util/utilTorchDataLoader.py:476-499 (cityscapes): the two images are swapped and mirrored; every pixel of the disparity
 * and one-hot maps moves to column max(int(c - disp), 0) of its row (the largest source column wins a contested target,
 * untouched targets keep their content), the last 10 / 20 columns are cleared, pixels without disparity go to the void
 * channel (the last of n_seg), then both maps are mirrored.  left/right: H x W pixels of `dtype`, pixel stride ld_img;
 * seg: f32 one-hot, pixel stride ld_seg; disp: dense f32.  workspace: sdhip_flip_sample_workspace_bytes(H, W, n_seg). */
/* augment_DoubleLeftImg of RandomCrop (util/utilTorchDataLoader.py:469-474), in place on ONE prepared sample: left is
 * mirrored, right becomes the mirrored left, the one-hot map is mirrored, the disparity is 0.0001 everywhere. */
int sdhip_double_left_sample(void* left, void* right, int ld_img, float* seg, int ld_seg, int n_seg, float* disp, int H, int W,
                             int dtype, void* stream);
long sdhip_flip_sample_workspace_bytes(int H, int W, int n_seg);
int sdhip_flip_sample(void* left, void* right, int ld_img, float* seg, int ld_seg, int n_seg, float* disp, int H, int W,
                      void* workspace, long workspace_bytes, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDHIP_H_ */
