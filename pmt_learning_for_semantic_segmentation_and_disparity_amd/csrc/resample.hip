// Pooling, resizing and small broadcast ops on NHWC tensors for gfx950 (all HBM bound:
// one 16-byte channel chunk per lane, coalesced along the channel axis).
//
// Replaces, on the reference's hot path:
//   nn.MaxPool2d(3, stride 2, pad 1)            models/densenet.py:158
//   F.avg_pool2d(2,2) / nn.AvgPool2d(p, p)      models/densenet.py:232, models/dsnet_t2.py:1983-2022
//   F.interpolate nearest / bilinear            models/dsnet_t2.py:927-936,1204-1275,2037-2081; models/aspp.py:88
//   s2_d * at_s (1-channel attention broadcast) models/dsnet_t2.py:1258
// Index arithmetic follows ATen's CPU kernels so that results agree with the reference to rounding.
#include "sdhip_common.h"

namespace {

struct Img { int B, H, W, C, ld; };

// flat work item -> (pixel, channel unit)
template <int N>
__device__ __forceinline__ bool item(long i, long npix, int units, long& pix, int& c0) {
  if (i >= npix * units) return false;
  pix = i / units;
  c0 = (int)(i - pix * units) * N;
  return true;
}

// ---------------------------------------------------------------- max pool 3x3 / stride 2 / pad 1
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void maxpool3s2_fwd(const T* __restrict__ x, Img in, T* __restrict__ y, Img out,
                                                      unsigned char* __restrict__ idx) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = in.C / N;
  const long npix = (long)out.B * out.H * out.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    const int ow = (int)(pix % out.W), oh = (int)((pix / out.W) % out.H);
    const long b = pix / ((long)out.W * out.H);
    float best[N]; int bi[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * 2 - 1 + kh;
      if (h < 0 || h >= in.H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int w = ow * 2 - 1 + kw;
        if (w < 0 || w >= in.W) continue;
        float v[N];
        Unit<T, VEC>::load(x + ((b * in.H + h) * in.W + w) * in.ld + c0, v);
#pragma unroll
        for (int e = 0; e < N; ++e)
          if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }  // first max wins, NaN propagates (ATen)
      }
    }
    Unit<T, VEC>::store(y + pix * out.ld + c0, best);
#pragma unroll
    for (int e = 0; e < N; ++e) idx[pix * in.C + c0 + e] = (unsigned char)bi[e];
  }
}

// gather form: every input pixel looks at the <= 2x2 windows that contain it
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void maxpool3s2_bwd(const T* __restrict__ gy, Img out, const unsigned char* __restrict__ idx,
                                                      T* __restrict__ gx, Img in) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = in.C / N;
  const long npix = (long)in.B * in.H * in.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    const int w = (int)(pix % in.W), h = (int)((pix / in.W) % in.H);
    const long b = pix / ((long)in.W * in.H);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {   // windows 2*oh-1 .. 2*oh+1 that contain row h
      if (oh < 0 || oh >= out.H) continue;
      const int kh = h - (oh * 2 - 1);
      if (kh < 0 || kh > 2) continue;
      for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
        if (ow < 0 || ow >= out.W) continue;
        const int kw = w - (ow * 2 - 1);
        if (kw < 0 || kw > 2) continue;
        const long opix = (b * out.H + oh) * out.W + ow;
        float g[N];
        Unit<T, VEC>::load(gy + opix * out.ld + c0, g);
#pragma unroll
        for (int e = 0; e < N; ++e)
          if (idx[opix * in.C + c0 + e] == kh * 3 + kw) acc[e] += g[e];
      }
    }
    Unit<T, VEC>::store(gx + pix * in.ld + c0, acc);
  }
}

// ---------------------------------------------------------------- average pool k x k, stride k, no padding
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void avgpool_fwd(const T* __restrict__ x, Img in, T* __restrict__ y, Img out, int k) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = in.C / N;
  const long npix = (long)out.B * out.H * out.W;
  const float inv = 1.f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    const int ow = (int)(pix % out.W), oh = (int)((pix / out.W) % out.H);
    const long b = pix / ((long)out.W * out.H);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        float v[N];
        Unit<T, VEC>::load(x + ((b * in.H + oh * k + kh) * in.W + ow * k + kw) * in.ld + c0, v);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += v[e];
      }
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] *= inv;
    Unit<T, VEC>::store(y + pix * out.ld + c0, acc);
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void avgpool_bwd(const T* __restrict__ gy, Img out, T* __restrict__ gx, Img in, int k) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = in.C / N;
  const long npix = (long)in.B * in.H * in.W;
  const float inv = 1.f / (float)(k * k);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    const int w = (int)(pix % in.W), h = (int)((pix / in.W) % in.H);
    const long b = pix / ((long)in.W * in.H);
    const int oh = h / k, ow = w / k;
    float g[N];
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = 0.f;
    if (oh < out.H && ow < out.W) {
      Unit<T, VEC>::load(gy + ((b * out.H + oh) * out.W + ow) * out.ld + c0, g);
#pragma unroll
      for (int e = 0; e < N; ++e) g[e] *= inv;
    }
    Unit<T, VEC>::store(gx + pix * in.ld + c0, g);
  }
}

// ---------------------------------------------------------------- resize (ATen area_pixel_* index maths, f32)
struct Axis { int in, out; float scale; int mode; };  // mode 0 nearest, 1 bilinear (align_corners=False), 2 bilinear (True)

__device__ __forceinline__ int nearest_src(const Axis& a, int d) {
  const int s = (int)floorf((float)d * a.scale);
  return s < a.in - 1 ? s : a.in - 1;
}
__device__ __forceinline__ void linear_src(const Axis& a, int d, int& i0, int& i1, float& l1) {
  float s;
  if (a.mode == 2) s = (float)d * a.scale;
  else { s = ((float)d + 0.5f) * a.scale - 0.5f; if (s < 0.f) s = 0.f; }
  i0 = (int)s;
  if (i0 > a.in - 1) i0 = a.in - 1;
  i1 = i0 + (i0 < a.in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
  if (l1 > 1.f) l1 = 1.f;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void resize_fwd(const T* __restrict__ x, Img in, T* __restrict__ y, Img out, Axis ah, Axis aw) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = in.C / N;
  const long npix = (long)out.B * out.H * out.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    const int ow = (int)(pix % out.W), oh = (int)((pix / out.W) % out.H);
    const long b = pix / ((long)out.W * out.H);
    const T* xb = x + b * in.H * in.W * in.ld + c0;
    float r[N];
    if (ah.mode == 0) {
      Unit<T, VEC>::load(xb + ((long)nearest_src(ah, oh) * in.W + nearest_src(aw, ow)) * in.ld, r);
    } else {
      int h0, h1, w0, w1; float lh, lw;
      linear_src(ah, oh, h0, h1, lh);
      linear_src(aw, ow, w0, w1, lw);
      float a[N], bb[N], c[N], d[N];
      Unit<T, VEC>::load(xb + ((long)h0 * in.W + w0) * in.ld, a);
      Unit<T, VEC>::load(xb + ((long)h0 * in.W + w1) * in.ld, bb);
      Unit<T, VEC>::load(xb + ((long)h1 * in.W + w0) * in.ld, c);
      Unit<T, VEC>::load(xb + ((long)h1 * in.W + w1) * in.ld, d);
      const float h0l = 1.f - lh, w0l = 1.f - lw;
#pragma unroll
      for (int e = 0; e < N; ++e) r[e] = h0l * (w0l * a[e] + lw * bb[e]) + lh * (w0l * c[e] + lw * d[e]);
    }
    Unit<T, VEC>::store(y + pix * out.ld + c0, r);
  }
}

// 1-D gather backward along one axis (the other axis is carried in `outer`):
//   gx[o, i, r, c] = sum over d with weight(d -> i) * gy[o, d, r, c]
// layout: [outer][axis][inner pixels][C]; a source index i receives from destinations whose
// (nearest or linear) source set contains i; candidates are bounded by the scale.
// `split` (1, 4 or 16, a power of two) lanes share one output item and take every split-th candidate: the pooled
// pyramid branches are upsampled 16-64x, so an item has up to ~130 candidates while there are only a few thousand items —
// one lane per item walked them in a serial, latency-bound chain (28 us per pass on tensors of a few MB).
template <typename T, bool VEC, typename TO>
__global__ __launch_bounds__(256) void resize_bwd_axis(const T* __restrict__ gy, int ldg, TO* __restrict__ gx, int ldx,
                                                       long outer, int inner, int C, Axis a, int split) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = C / N;
  const long npix = outer * a.in * inner;
  const int sub = threadIdx.x & (split - 1);
  const int per_block = 256 / split;
  for (long it = (long)blockIdx.x * per_block + threadIdx.x / split;; it += (long)gridDim.x * per_block) {
    long pix; int c0;
    if (!item<N>(it, npix, units, pix, c0)) break;       // uniform over the lanes of an item
    const int r = (int)(pix % inner);
    const int i = (int)((pix / inner) % a.in);
    const long o = pix / ((long)inner * a.in);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    // destination candidates: d*scale within (i-1-eps, i+1+eps)
    const float inv = a.scale > 0.f ? 1.f / a.scale : 0.f;
    int lo = (int)floorf(((float)i - 1.f) * inv) - 2, hi = (int)ceilf(((float)i + 1.f) * inv) + 2;
    if (a.scale <= 0.f) { lo = 0; hi = a.out - 1; }
    if (lo < 0) lo = 0;
    if (hi > a.out - 1) hi = a.out - 1;
    for (int d = lo + sub; d <= hi; d += split) {
      float wgt = 0.f;
      if (a.mode == 0) {
        wgt = nearest_src(a, d) == i ? 1.f : 0.f;
      } else {
        int i0, i1; float l1;
        linear_src(a, d, i0, i1, l1);
        if (i0 == i) wgt += 1.f - l1;
        if (i1 == i) wgt += l1;
      }
      if (wgt != 0.f) {
        float g[N];
        Unit<T, VEC>::load(gy + ((o * a.out + d) * inner + r) * ldg + c0, g);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
      }
    }
    for (int o2 = split >> 1; o2 > 0; o2 >>= 1)
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] += __shfl_xor(acc[e], o2, 64);
    if (sub != 0) continue;
    TO* dst = gx + pix * ldx + c0;
    if constexpr (sizeof(TO) == sizeof(T)) {
      Unit<T, VEC>::store(reinterpret_cast<T*>(dst), acc);
    } else {  // f32 intermediate between the two passes of a bf16 tensor
#pragma unroll
      for (int e = 0; e < N; ++e) Elem<TO>::st(dst + e, acc[e]);
    }
  }
}

// ---------------------------------------------------------------- y = a * m  (m has one channel)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void mul_bcast_fwd(const T* __restrict__ a, int lda, const T* __restrict__ m, int ldm,
                                                     T* __restrict__ y, int ldy, long npix, int C) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = C / N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x;; i += (long)gridDim.x * 256) {
    long pix; int c0;
    if (!item<N>(i, npix, units, pix, c0)) break;
    float v[N];
    Unit<T, VEC>::load(a + pix * lda + c0, v);
    const float mv = Elem<T>::ld(m + pix * ldm);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] *= mv;
    Unit<T, VEC>::store(y + pix * ldy + c0, v);
  }
}

// ga = g * m ; gm = sum_c g * a   (one thread per pixel walks the channels: C is small, 64 here)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void mul_bcast_bwd(const T* __restrict__ g, int ldg, const T* __restrict__ a, int lda,
                                                     const T* __restrict__ m, int ldm, T* __restrict__ ga, int ldga,
                                                     T* __restrict__ gm, int ldgm, long npix, int C) {
  constexpr int N = Unit<T, VEC>::N;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
    const float mv = Elem<T>::ld(m + pix * ldm);
    float s = 0.f;
    for (int c0 = 0; c0 < C; c0 += N) {
      float gv[N], av[N];
      Unit<T, VEC>::load(g + pix * ldg + c0, gv);
      Unit<T, VEC>::load(a + pix * lda + c0, av);
#pragma unroll
      for (int e = 0; e < N; ++e) { s = fmaf(gv[e], av[e], s); gv[e] *= mv; }
      Unit<T, VEC>::store(ga + pix * ldga + c0, gv);
    }
    Elem<T>::st(gm + pix * ldgm, s);
  }
}

template <typename T>
bool vec_ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  const int n = Chunk<T>::N;
  if (C % n) return false;
  for (int l : lds) if (l % n) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

inline dim3 grid_for(long items) {
  long b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

Axis make_axis(int in, int out, int mode, float scale_override) {
  Axis a; a.in = in; a.out = out; a.mode = mode;
  if (mode == 2) a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  else a.scale = scale_override > 0.f ? scale_override : (float)in / (float)out;
  return a;
}

}  // namespace

static int check_img(const char* who, int B, int H, int W, int C, int ld, int dtype) {
  SDHIP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && ld >= C, "%s: bad tensor B=%d H=%d W=%d C=%d ld=%d", who, B, H, W, C, ld);
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "%s: unknown dtype %d", who, dtype);
  return 0;
}

extern "C" int sdhip_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, unsigned char* idx,
                                    int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check_img("maxpool3s2_fwd", B, H, W, C, ldx, dtype)) return rc;
  SDHIP_CHECK_ARG(x && y && idx && ldy >= C, "maxpool3s2_fwd: bad pointers");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  Img in{B, H, W, C, ldx}, out{B, Ho, Wo, C, ldy};
  const long np = (long)B * Ho * Wo;
#define VOK(T) vec_ok<T>(C, {ldx, ldy}, {x, y})
#define IV(n) (np * (C / (n)))
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((maxpool3s2_fwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, idx);
    else hipLaunchKernelGGL((maxpool3s2_fwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, idx);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((maxpool3s2_fwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, idx);
    else hipLaunchKernelGGL((maxpool3s2_fwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, idx);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_maxpool3s2_bwd(const void* gy, int ldg, const unsigned char* idx, void* gx, int ldgx,
                                    int B, int H, int W, int C, int dtype, void* stream) {
  if (int rc = check_img("maxpool3s2_bwd", B, H, W, C, ldgx, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && gx && idx && ldg >= C, "maxpool3s2_bwd: bad pointers");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  Img in{B, H, W, C, ldgx}, out{B, Ho, Wo, C, ldg};
  const long np = (long)B * H * W;
#define VOK(T) vec_ok<T>(C, {ldg, ldgx}, {gy, gx})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((maxpool3s2_bwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)gy, out, idx, (float*)gx, in);
    else hipLaunchKernelGGL((maxpool3s2_bwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)gy, out, idx, (float*)gx, in);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((maxpool3s2_bwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gy, out, idx, (bf16_t*)gx, in);
    else hipLaunchKernelGGL((maxpool3s2_bwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gy, out, idx, (bf16_t*)gx, in);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_avgpool_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int k,
                                 int dtype, void* stream) {
  if (int rc = check_img("avgpool_fwd", B, H, W, C, ldx, dtype)) return rc;
  SDHIP_CHECK_ARG(x && y && ldy >= C && k >= 1 && H >= k && W >= k, "avgpool_fwd: bad arguments (k=%d H=%d W=%d)", k, H, W);
  Img in{B, H, W, C, ldx}, out{B, H / k, W / k, C, ldy};
  const long np = (long)B * out.H * out.W;
#define VOK(T) vec_ok<T>(C, {ldx, ldy}, {x, y})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((avgpool_fwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, k);
    else hipLaunchKernelGGL((avgpool_fwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, k);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((avgpool_fwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, k);
    else hipLaunchKernelGGL((avgpool_fwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, k);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_avgpool_bwd(const void* gy, int ldg, void* gx, int ldgx, int B, int H, int W, int C, int k,
                                 int dtype, void* stream) {
  if (int rc = check_img("avgpool_bwd", B, H, W, C, ldgx, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && gx && ldg >= C && k >= 1 && H >= k && W >= k, "avgpool_bwd: bad arguments");
  Img in{B, H, W, C, ldgx}, out{B, H / k, W / k, C, ldg};
  const long np = (long)B * H * W;
#define VOK(T) vec_ok<T>(C, {ldg, ldgx}, {gy, gx})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((avgpool_bwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)gy, out, (float*)gx, in, k);
    else hipLaunchKernelGGL((avgpool_bwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)gy, out, (float*)gx, in, k);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((avgpool_bwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gy, out, (bf16_t*)gx, in, k);
    else hipLaunchKernelGGL((avgpool_bwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gy, out, (bf16_t*)gx, in, k);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_resize_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                                int mode, float scale_h, float scale_w, int dtype, void* stream) {
  if (int rc = check_img("resize_fwd", B, H, W, C, ldx, dtype)) return rc;
  SDHIP_CHECK_ARG(x && y && ldy >= C && Ho > 0 && Wo > 0 && mode >= 0 && mode <= 2, "resize_fwd: bad arguments");
  Img in{B, H, W, C, ldx}, out{B, Ho, Wo, C, ldy};
  const Axis ah = make_axis(H, Ho, mode, scale_h), aw = make_axis(W, Wo, mode, scale_w);
  const long np = (long)B * Ho * Wo;
#define VOK(T) vec_ok<T>(C, {ldx, ldy}, {x, y})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((resize_fwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, ah, aw);
    else hipLaunchKernelGGL((resize_fwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)x, in, (float*)y, out, ah, aw);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((resize_fwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, ah, aw);
    else hipLaunchKernelGGL((resize_fwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, in, (bf16_t*)y, out, ah, aw);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

/* gx (B,H,W,C) from gy (B,Ho,Wo,C): separable, W axis first into the f32 workspace tmp (B*Ho*W*C floats), then H. */
extern "C" int sdhip_resize_bwd(const void* gy, int ldg, void* gx, int ldgx, float* tmp, int B, int H, int W, int C,
                                int Ho, int Wo, int mode, float scale_h, float scale_w, int dtype, void* stream) {
  if (int rc = check_img("resize_bwd", B, H, W, C, ldgx, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && gx && tmp && ldg >= C && Ho > 0 && Wo > 0 && mode >= 0 && mode <= 2, "resize_bwd: bad arguments");
  const Axis ah = make_axis(H, Ho, mode, scale_h), aw = make_axis(W, Wo, mode, scale_w);
  hipStream_t s = (hipStream_t)stream;
  // pass 1: along W.  layout [outer = B*Ho][axis = W][inner = 1][C]; output f32 tmp with ld = C
  const long np1 = (long)B * Ho * W;
  // pass 2: along H.  layout [outer = B][axis = H][inner = W][C]; input tmp (f32), output gx (T)
  const long np2 = (long)B * H * W;
  auto lanes = [](const Axis& a) {      // candidates per item ~ 2 / scale + 5 (see the kernel's [lo, hi] window)
    const float span = a.scale > 0.f ? 2.f / a.scale + 5.f : (float)a.out;
    return span >= 32.f ? 16 : span >= 12.f ? 4 : 1;
  };
  const int sw = lanes(aw), sh = lanes(ah);
  if (dtype == SDHIP_F32) {
    const bool v1 = vec_ok<float>(C, {ldg}, {gy, tmp}), v2 = vec_ok<float>(C, {ldgx}, {gx, tmp});
    if (v1) hipLaunchKernelGGL((resize_bwd_axis<float, true, float>), grid_for(np1 * (C / 4) * sw), dim3(256), 0, s, (const float*)gy, ldg, tmp, C, (long)B * Ho, 1, C, aw, sw);
    else hipLaunchKernelGGL((resize_bwd_axis<float, false, float>), grid_for(np1 * C * sw), dim3(256), 0, s, (const float*)gy, ldg, tmp, C, (long)B * Ho, 1, C, aw, sw);
    if (v2) hipLaunchKernelGGL((resize_bwd_axis<float, true, float>), grid_for(np2 * (C / 4) * sh), dim3(256), 0, s, (const float*)tmp, C, (float*)gx, ldgx, (long)B, W, C, ah, sh);
    else hipLaunchKernelGGL((resize_bwd_axis<float, false, float>), grid_for(np2 * C * sh), dim3(256), 0, s, (const float*)tmp, C, (float*)gx, ldgx, (long)B, W, C, ah, sh);
  } else {
    const bool v1 = vec_ok<bf16_t>(C, {ldg}, {gy});
    if (v1) hipLaunchKernelGGL((resize_bwd_axis<bf16_t, true, float>), grid_for(np1 * (C / 8) * sw), dim3(256), 0, s, (const bf16_t*)gy, ldg, tmp, C, (long)B * Ho, 1, C, aw, sw);
    else hipLaunchKernelGGL((resize_bwd_axis<bf16_t, false, float>), grid_for(np1 * C * sw), dim3(256), 0, s, (const bf16_t*)gy, ldg, tmp, C, (long)B * Ho, 1, C, aw, sw);
    // second pass reads f32, writes bf16: scalar units keep it simple (the tensor is the small, pre-upsampling one)
    hipLaunchKernelGGL((resize_bwd_axis<float, false, bf16_t>), grid_for(np2 * C * sh), dim3(256), 0, s, (const float*)tmp, C, (bf16_t*)gx, ldgx, (long)B, W, C, ah, sh);
  }
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_mul_bcast_fwd(const void* a, int lda, const void* m, int ldm, void* y, int ldy, long npix, int C,
                                   int dtype, void* stream) {
  SDHIP_CHECK_ARG(a && m && y && npix > 0 && C > 0 && lda >= C && ldy >= C && ldm >= 1, "mul_bcast_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "mul_bcast_fwd: unknown dtype %d", dtype);
  const long np = npix;
#define VOK(T) vec_ok<T>(C, {lda, ldy}, {a, y})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((mul_bcast_fwd<float, true>), grid_for(IV(4)), dim3(256), 0, (hipStream_t)stream, (const float*)a, lda, (const float*)m, ldm, (float*)y, ldy, npix, C);
    else hipLaunchKernelGGL((mul_bcast_fwd<float, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const float*)a, lda, (const float*)m, ldm, (float*)y, ldy, npix, C);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((mul_bcast_fwd<bf16_t, true>), grid_for(IV(8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)m, ldm, (bf16_t*)y, ldy, npix, C);
    else hipLaunchKernelGGL((mul_bcast_fwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)m, ldm, (bf16_t*)y, ldy, npix, C);
  }
#undef VOK
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_mul_bcast_bwd(const void* g, int ldg, const void* a, int lda, const void* m, int ldm,
                                   void* ga, int ldga, void* gm, int ldgm, long npix, int C, int dtype, void* stream) {
  SDHIP_CHECK_ARG(g && a && m && ga && gm && npix > 0 && C > 0 && ldg >= C && lda >= C && ldga >= C, "mul_bcast_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "mul_bcast_bwd: unknown dtype %d", dtype);
#define VOK(T) vec_ok<T>(C, {ldg, lda, ldga}, {g, a, ga})
  if (dtype == SDHIP_F32) {
    if (VOK(float)) hipLaunchKernelGGL((mul_bcast_bwd<float, true>), grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const float*)g, ldg, (const float*)a, lda, (const float*)m, ldm, (float*)ga, ldga, (float*)gm, ldgm, npix, C);
    else hipLaunchKernelGGL((mul_bcast_bwd<float, false>), grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const float*)g, ldg, (const float*)a, lda, (const float*)m, ldm, (float*)ga, ldga, (float*)gm, ldgm, npix, C);
  } else {
    if (VOK(bf16_t)) hipLaunchKernelGGL((mul_bcast_bwd<bf16_t, true>), grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, ldg, (const bf16_t*)a, lda, (const bf16_t*)m, ldm, (bf16_t*)ga, ldga, (bf16_t*)gm, ldgm, npix, C);
    else hipLaunchKernelGGL((mul_bcast_bwd<bf16_t, false>), grid_for(npix), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, ldg, (const bf16_t*)a, lda, (const bf16_t*)m, ldm, (bf16_t*)ga, ldga, (bf16_t*)gm, ldgm, npix, C);
  }
#undef VOK
#undef IV
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
