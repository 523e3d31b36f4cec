// Training-step tail for gfx950: fused Adam over one flat parameter buffer, and the per-pixel
// losses of the timed step (categorical cross-entropy on log-softmax, L1), forward + gradient in one pass.
//
// Replaces torch.optim.Adam(lr=0.0015, eps=1e-7) (torch_implementation.py:718-724),
// categoricalCrossEntropy (util/utilTorchLoss.py:373-378, fed by F.log_softmax at
// losses/multiLosses.py:42) and nn.L1Loss (losses/multiLosses.py:141).
#include "sdhip_common.h"
#include "rows_lds.h"

namespace {

// state[0] = beta1^t, state[1] = beta2^t (device resident so the step is graph-replayable)
__global__ void adam_tick_kernel(float* state, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { state[0] *= b1; state[1] *= b2; }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float gscale, const float* __restrict__ state) {
  const float bc1 = 1.f - state[0], bc2 = 1.f - state[1];
  const float step = lr / bc1, ibc2 = 1.f / sqrtf(bc2);
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gv[e] * gscale + wd * pv[e];
        mv[e] = b1 * mv[e] + (1.f - b1) * gr;
        vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
        pv[e] -= step * mv[e] / (sqrtf(vv[e]) * ibc2 + eps);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv; *reinterpret_cast<f32x4*>(m + i) = mv; *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long j = i; j < n; ++j) {
        const float gr = g[j] * gscale + wd * p[j];
        m[j] = b1 * m[j] + (1.f - b1) * gr;
        v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
        p[j] -= step * m[j] / (sqrtf(v[j]) * ibc2 + eps);
      }
    }
  }
}

// block reduce of one float, result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

// loss += weight/npix * sum_p sum_c -t[p,c] * log_softmax(y[p,:])_c ;  gy[p,c] = weight/npix * (softmax_c * sum_c t - t_c)
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                 T* __restrict__ gy, int ldg, double* __restrict__ loss, long npix, int C, float wnorm) {
  __shared__ float sh[4];
  float part = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const T* yp = y + p * ldy;
    const float* tp = t + p * ldt;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, Elem<T>::ld(yp + c));
    float se = 0.f, ts = 0.f, dot = 0.f;
    for (int c = 0; c < C; ++c) { const float z = Elem<T>::ld(yp + c) - mx; se += __expf(z); ts += tp[c]; dot += tp[c] * z; }
    const float lse = __logf(se);
    part += ts * lse - dot;  // sum_c -t_c (z_c - lse)
    if (gy) {
      const float inv = 1.f / se;
      for (int c = 0; c < C; ++c) {
        const float sm = __expf(Elem<T>::ld(yp + c) - mx) * inv;
        Elem<T>::st(gy + p * ldg + c, wnorm * (sm * ts - tp[c]));
      }
    }
  }
  const float tot = block_sum(part, sh);
  if (threadIdx.x == 0) atomicAdd(loss, (double)tot * (double)wnorm);
}

// The same for many classes (5 <= C <= 64; 19 Cityscapes classes): the workgroup's 256 pixel rows travel through LDS (rows_lds.h)
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                      T* __restrict__ gy, int ldg, double* __restrict__ loss, long npix, int C, float wnorm,
                                                      int off_t, int off_g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  __shared__ float sh[4];
  T* const ly = reinterpret_cast<T*>(rsm);
  float* const lt = reinterpret_cast<float*>(rsm + off_t);
  T* const lg = reinterpret_cast<T*>(rsm + off_g);
  const int tid = threadIdx.x;
  const long ntiles = (npix + 255) / 256;
  float part = 0.f;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long p0 = tile * 256;
    const int n = (int)min(256L, npix - p0);
    rows_to_lds(y + p0 * ldy, ly, n * ldy * (int)sizeof(T), tid);
    rows_to_lds(t + p0 * ldt, lt, n * ldt * 4, tid);
    __syncthreads();
    if (tid < n) {
      const T* yp = ly + tid * ldy;
      const float* tp = lt + tid * ldt;
      float mx = -INFINITY;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, Elem<T>::ld(yp + c));
      float se = 0.f, ts = 0.f, dot = 0.f;
      for (int c = 0; c < C; ++c) { const float z = Elem<T>::ld(yp + c) - mx; se += __expf(z); ts += tp[c]; dot += tp[c] * z; }
      part += ts * __logf(se) - dot;
      if (gy) {
        const float inv = 1.f / se;
        T* gp = lg + tid * ldg;
        for (int c = 0; c < C; ++c) Elem<T>::st(gp + c, wnorm * (__expf(Elem<T>::ld(yp + c) - mx) * inv * ts - tp[c]));
        for (int c = C; c < ldg; ++c) Elem<T>::st(gp + c, 0.f);      // (pad channels of the pixel stride: defined, never data)
      }
    }
    __syncthreads();
    if (gy) rows_from_lds(gy + p0 * ldg, lg, n * ldg * (int)sizeof(T), tid);
    __syncthreads();
  }
  const float tot = block_sum(part, sh);
  if (threadIdx.x == 0) atomicAdd(loss, (double)tot * (double)wnorm);
}

// loss += weight/n * sum |a - b| ; ga = weight/n * sign(a - b).  mask_nonpositive: elements whose target is <= 0 count as
// zero difference but stay in the mean (L1(pred*zeros, disp*zeros), zeros = disp > 0: losses/multiLosses.py:138-141)
template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(const T* __restrict__ a, const float* __restrict__ b, T* __restrict__ ga,
                                                 double* __restrict__ loss, long n, float wnorm, int mask_nonpositive) {
  __shared__ float sh[4];
  float part = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float bi = b[i];
    const float d = (mask_nonpositive && !(bi > 0.f)) ? 0.f : Elem<T>::ld(a + i) - bi;
    part += fabsf(d);
    if (ga) Elem<T>::st(ga + i, d > 0.f ? wnorm : (d < 0.f ? -wnorm : 0.f));
  }
  const float tot = block_sum(part, sh);
  if (threadIdx.x == 0) atomicAdd(loss, (double)tot * (double)wnorm);
}

inline dim3 grid_for(long items) {
  long b = (items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

// the loss kernels end with ONE f64 atomic per workgroup on the same scalar: two workgroups per CU instead of eight
inline dim3 grid_loss(long items) {
  long b = (items + 255) / 256;
  if (b > 512) b = 512;       // (19 classes: 2048 workgroups measured SLOWER, 394 -> 625 us — the per-thread rows thrash the L1)
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

extern "C" int sdhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* beta_pow,
                               long n, float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                               void* stream) {
  SDHIP_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && beta_pow && n > 0, "adam_step: bad arguments");
  SDHIP_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                  "adam_step: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, beta_pow, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel, grid_for((n + 3) / 4), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n, lr, beta1,
                     beta2, eps, weight_decay, grad_scale, beta_pow);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_ce_loss(const void* logits, int ldy, const float* target, int ldt, void* grad, int ldg, double* loss,
                             long npix, int C, float weight, int dtype, void* stream) {
  SDHIP_CHECK_ARG(logits && target && loss && npix > 0 && C > 0 && ldy >= C && ldt >= C && (!grad || ldg >= C), "ce_loss: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "ce_loss: unknown dtype %d", dtype);
  const float wn = weight / (float)npix;
  hipStream_t s = (hipStream_t)stream;
  const int es = dtype == SDHIP_F32 ? 4 : 2;
  const size_t by = rows_lds_bytes(ldy, es), bt = rows_lds_bytes(ldt, 4), bg = grad ? rows_lds_bytes(ldg, es) : 0;
  if (C > 4 && C <= 64 && by + bt + bg <= 60 * 1024 && ((uintptr_t)logits & 3) == 0 && (!grad || ((uintptr_t)grad & 3) == 0)) {
    long b = (npix + 255) / 256;
    if (b > 768) b = 768;
    if (dtype == SDHIP_F32)
      hipLaunchKernelGGL(ce_rows_kernel<float>, dim3((unsigned)b), dim3(256), by + bt + bg, s, (const float*)logits, ldy, target, ldt, (float*)grad, ldg, loss, npix, C, wn, (int)by, (int)(by + bt));
    else
      hipLaunchKernelGGL(ce_rows_kernel<bf16_t>, dim3((unsigned)b), dim3(256), by + bt + bg, s, (const bf16_t*)logits, ldy, target, ldt, (bf16_t*)grad, ldg, loss, npix, C, wn, (int)by, (int)(by + bt));
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(ce_kernel<float>, grid_loss(npix), dim3(256), 0, s, (const float*)logits, ldy, target, ldt, (float*)grad, ldg, loss, npix, C, wn);
  else
    hipLaunchKernelGGL(ce_kernel<bf16_t>, grid_loss(npix), dim3(256), 0, s, (const bf16_t*)logits, ldy, target, ldt, (bf16_t*)grad, ldg, loss, npix, C, wn);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_l1_loss(const void* pred, const float* target, void* grad, double* loss, long n, float weight,
                             int mask_nonpositive, int dtype, void* stream) {
  SDHIP_CHECK_ARG(pred && target && loss && n > 0, "l1_loss: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "l1_loss: unknown dtype %d", dtype);
  const float wn = weight / (float)n;
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(l1_kernel<float>, grid_loss(n), dim3(256), 0, (hipStream_t)stream, (const float*)pred, target, (float*)grad, loss, n, wn, mask_nonpositive);
  else
    hipLaunchKernelGGL(l1_kernel<bf16_t>, grid_loss(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, target, (bf16_t*)grad, loss, n, wn, mask_nonpositive);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// ---- dropout (nn.Dropout in ASPP, models/aspp.py:79,95): counter-based hash RNG, so the backward pass regenerates
// the mask from (seed, layer id, element index) instead of storing it; the seed lives in device memory and is advanced
// once per step by the training harness, which keeps the step graph-replayable.
namespace {
__device__ __forceinline__ unsigned int hash32(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned int)x;
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, const long* __restrict__ seed,
                                                      long layer, long n, float p, float scale) {
  const unsigned long long s = (unsigned long long)seed[0] * 0x9E3779B97F4A7C15ULL + (unsigned long long)layer * 0xD1B54A32D192ED03ULL;
  const unsigned int thr = (unsigned int)((double)p * 4294967296.0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const bool keep = hash32(s + (unsigned long long)i) >= thr;
    Elem<T>::st(y + i, keep ? Elem<T>::ld(x + i) * scale : 0.f);
  }
}
}  // namespace

extern "C" int sdhip_dropout(const void* x, void* y, const long* seed, long layer_id, long n, float p, int dtype, void* stream) {
  SDHIP_CHECK_ARG(x && y && seed && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "dropout: unknown dtype %d", dtype);
  const float scale = 1.f / (1.f - p);
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, seed, layer_id, n, p, scale);
  else
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, grid_for(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, seed, layer_id, n, p, scale);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// ---- log_softmax over the channel axis of an NHWC tensor (F.log_softmax(x, dim=1) of models/dsnet_t2.py:216,270)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, long npix, int C) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const T* xp = x + p * ldx;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, Elem<T>::ld(xp + c));
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(Elem<T>::ld(xp + c) - mx);
    const float lse = mx + __logf(se);
    for (int c = 0; c < C; ++c) Elem<T>::st(y + p * ldy + c, Elem<T>::ld(xp + c) - lse);
  }
}
// gx = gy - exp(y) * sum_c gy   (y = log_softmax output)
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const T* __restrict__ gy, int ldg, const T* __restrict__ y, int ldy,
                                                              T* __restrict__ gx, int ldgx, long npix, int C) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += Elem<T>::ld(gy + p * ldg + c);
    for (int c = 0; c < C; ++c)
      Elem<T>::st(gx + p * ldgx + c, Elem<T>::ld(gy + p * ldg + c) - __expf(Elem<T>::ld(y + p * ldy + c)) * s);
  }
}
}  // namespace

extern "C" int sdhip_log_softmax_fwd(const void* x, int ldx, void* y, int ldy, long npix, int C, int dtype, void* stream) {
  SDHIP_CHECK_ARG(x && y && npix > 0 && C > 0 && ldx >= C && ldy >= C, "log_softmax_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "log_softmax_fwd: unknown dtype %d", dtype);
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(log_softmax_fwd_kernel<float>, grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, npix, C);
  else hipLaunchKernelGGL(log_softmax_fwd_kernel<bf16_t>, grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, npix, C);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_log_softmax_bwd(const void* gy, int ldg, const void* y, int ldy, void* gx, int ldgx, long npix, int C,
                                     int dtype, void* stream) {
  SDHIP_CHECK_ARG(gy && y && gx && npix > 0 && C > 0 && ldg >= C && ldy >= C && ldgx >= C, "log_softmax_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "log_softmax_bwd: unknown dtype %d", dtype);
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(log_softmax_bwd_kernel<float>, grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const float*)gy, ldg, (const float*)y, ldy, (float*)gx, ldgx, npix, C);
  else hipLaunchKernelGGL(log_softmax_bwd_kernel<bf16_t>, grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gy, ldg, (const bf16_t*)y, ldy, (bf16_t*)gx, ldgx, npix, C);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
