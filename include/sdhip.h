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

#ifdef __cplusplus
}
#endif
#endif /* SDHIP_H_ */
