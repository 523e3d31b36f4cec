// Height-driven attention (HANet) data kernels for gfx950.
//
// Replaces, inside `HANet_Conv.forward` (models_hanet/HANet.py:74-128):
//   rowpool      nn.AdaptiveMaxPool2d((L, 1)) over the feature map (:50-55,84)      -> sdhip_rowpool_max_fwd / _bwd
//   Dropout2d    nn.Dropout2d(p) on the (B, C, L) row descriptor (:27-28,90-91)    -> sdhip_dropout_channels
//   re-weighting torch.mul(out, attention.unsqueeze(3)) (:112)                     -> sdhip_mul_rows_fwd / _bwd
// The 1-D convolutions / BatchNorm1d / sigmoid / linear resize of the tiny (B, C, L) tensors in between run through the
// library's conv / BatchNorm / resize kernels on (B, C, L, 1) images.  All tensors are NHWC with a pixel stride.
#include "sdhip_common.h"

namespace {

// one workgroup per (image b, output row bin i): max over rows [floor(i*H/OH), ceil((i+1)*H/OH)) x all columns.
// First maximum in scan order wins (ATen's adaptive_max_pool2d compares with `>` / NaN), its flat index h*W+w is kept.
template <typename T>
__global__ __launch_bounds__(256) void rowpool_max_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                              int* __restrict__ idx, int H, int W, int C, int OH) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int b = blockIdx.y, i = blockIdx.x;
  const int h0 = (int)(((long)i * H) / OH), h1 = (int)((((long)i + 1) * H + OH - 1) / OH);
  const int lanes_c = C < 256 ? C : 256;             // threads across channels
  const int wl = 256 / lanes_c;                      // column lanes
  const int tc = threadIdx.x % lanes_c, tw = threadIdx.x / lanes_c;
  for (int c0 = 0; c0 < C; c0 += lanes_c) {
    const int c = c0 + tc;
    float best = -INFINITY;
    int bi = 0x7fffffff;                               // no element seen
    if (tw < wl && c < C && tw < W) {
      bi = h0 * W + tw;
      for (int h = h0; h < h1; ++h)
        for (int w = tw; w < W; w += wl) {
          const float v = Elem<T>::ld(x + (((long)b * H + h) * W + w) * ldx + c);
          if (v > best || v != v) { best = v; bi = h * W + w; }   // aten/native/AdaptiveMaxPooling2d: (val > maxval) || isnan(val)
        }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    if (tw == 0 && c < C) {
      float m = sv[tc]; int mi = si[tc];
      for (int k = 1; k < wl; ++k) {   // the earliest index among equal maxima = the first one in scan order
        const float v = sv[k * lanes_c + tc]; const int vi = si[k * lanes_c + tc];
        if (vi == 0x7fffffff) continue;
        if (v > m || (v == m && vi < mi) || (v != v && m == m)) { m = v; mi = vi; }
      }
      Elem<T>::st(y + ((long)b * OH + i) * ldy + c, m);
      idx[((long)b * OH + i) * C + c] = mi;
    }
    __syncthreads();
  }
}

// one thread per (image, channel) walks its OH bins in order and adds into the (pre-zeroed) input gradient: adjacent
// bins overlap by a row when H is not a multiple of OH, so two bins may share an argmax
template <typename T>
__global__ void rowpool_max_bwd_kernel(const T* __restrict__ gy, int ldg, const int* __restrict__ idx, T* __restrict__ gx, int ldgx,
                                       int H, int W, int C, int OH, long total) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const long b = t / C;
    for (int i = 0; i < OH; ++i) {
      const long bi = b * OH + i;
      T* dst = gx + (b * H * W + idx[bi * C + c]) * ldgx + c;
      Elem<T>::st(dst, Elem<T>::ld(dst) + Elem<T>::ld(gy + bi * ldg + c));
    }
  }
}

// y[b,h,w,c] = a[b,h,w,c] * att[b,h,c]
template <typename T>
__global__ void mul_rows_fwd_kernel(const T* __restrict__ a, int lda, const T* __restrict__ att, int ldt, T* __restrict__ y, int ldy,
                                    int W, int C, long total) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const long pix = t / C;              // (b*H + h)*W + w
    const long row = pix / W;            // b*H + h
    Elem<T>::st(y + pix * ldy + c, Elem<T>::ld(a + pix * lda + c) * Elem<T>::ld(att + row * ldt + c));
  }
}

// ga = g * att;  gatt[b,h,c] = sum_w g * a      (one workgroup per row (b,h))
template <typename T>
__global__ __launch_bounds__(256) void mul_rows_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ a, int lda,
                                                           const T* __restrict__ att, int ldt, T* __restrict__ ga, int ldga,
                                                           T* __restrict__ gatt, int ldgt, int W, int C) {
  __shared__ float red[256];
  const long row = blockIdx.x;
  const int lanes_c = C < 256 ? C : 256, wl = 256 / lanes_c;
  const int tc = threadIdx.x % lanes_c, tw = threadIdx.x / lanes_c;
  for (int c0 = 0; c0 < C; c0 += lanes_c) {
    const int c = c0 + tc;
    float s = 0.f;
    if (tw < wl && c < C) {
      const float at = Elem<T>::ld(att + row * ldt + c);
      for (int w = tw; w < W; w += wl) {
        const long pix = row * W + w;
        const float gv = Elem<T>::ld(g + pix * ldg + c);
        s = fmaf(gv, Elem<T>::ld(a + pix * lda + c), s);
        Elem<T>::st(ga + pix * ldga + c, gv * at);
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (tw == 0 && c < C) {
      float tot = 0.f;
      for (int k = 0; k < wl; ++k) tot += red[k * lanes_c + tc];
      Elem<T>::st(gatt + row * ldgt + c, tot);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ unsigned hash32(unsigned long long v) {
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL; v ^= v >> 33;
  return (unsigned)v;
}

// y[b,l,c] = x[b,l,c] * keep(b,c) / (1-p): whole channels of one sample are dropped (nn.Dropout2d on a (B,C,L) tensor)
template <typename T>
__global__ void dropout_channels_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, const long* __restrict__ seed,
                                        long layer, int L, int C, float p, long total) {
  const float scale = 1.f / (1.f - p);
  const unsigned thr = (unsigned)(p * 4294967296.0);
  const unsigned long long s0 = (unsigned long long)*seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)layer * 0xD1B54A32D192ED03ULL;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const long bl = t / C;
    const long b = bl / L;
    const bool keep = hash32(s0 + (unsigned long long)(b * C + c)) >= thr;
    Elem<T>::st(y + bl * ldy + c, keep ? Elem<T>::ld(x + bl * ldx + c) * scale : 0.f);
  }
}

int blocks_for(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int sdhip_rowpool_max_fwd(const void* x, int ldx, void* y, int ldy, int* idx, int B, int H, int W, int C, int OH,
                                     int dtype, void* stream) {
  SDHIP_CHECK_ARG(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && OH > 0, "rowpool_max_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "rowpool_max_fwd: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG((long)H * W < (1L << 31), "rowpool_max_fwd: image too large");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(rowpool_max_fwd_kernel<float>, dim3(OH, B), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, idx, H, W, C, OH);
  else hipLaunchKernelGGL(rowpool_max_fwd_kernel<bf16_t>, dim3(OH, B), dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, idx, H, W, C, OH);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_rowpool_max_bwd(const void* gy, int ldg, const int* idx, void* gx, int ldgx, int B, int H, int W, int C, int OH,
                                     int dtype, void* stream) {
  SDHIP_CHECK_ARG(gy && idx && gx && B > 0 && H > 0 && W > 0 && C > 0 && OH > 0, "rowpool_max_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "rowpool_max_bwd: unknown dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const size_t es = dtype == SDHIP_BF16 ? 2 : 4;
  if (sdhip_zero_async(gx, (size_t)B * H * W * ldgx * es, s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "rowpool_max_bwd: memset failed");
  const long total = (long)B * C;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(rowpool_max_bwd_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, s, (const float*)gy, ldg, idx, (float*)gx, ldgx, H, W, C, OH, total);
  else hipLaunchKernelGGL(rowpool_max_bwd_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)gy, ldg, idx, (bf16_t*)gx, ldgx, H, W, C, OH, total);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_mul_rows_fwd(const void* a, int lda, const void* att, int ldt, void* y, int ldy, int B, int H, int W, int C,
                                  int dtype, void* stream) {
  SDHIP_CHECK_ARG(a && att && y && B > 0 && H > 0 && W > 0 && C > 0, "mul_rows_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "mul_rows_fwd: unknown dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * H * W * C;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(mul_rows_fwd_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, s, (const float*)a, lda, (const float*)att, ldt, (float*)y, ldy, W, C, total);
  else hipLaunchKernelGGL(mul_rows_fwd_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)att, ldt, (bf16_t*)y, ldy, W, C, total);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_mul_rows_bwd(const void* g, int ldg, const void* a, int lda, const void* att, int ldt, void* ga, int ldga,
                                  void* gatt, int ldgt, int B, int H, int W, int C, int dtype, void* stream) {
  SDHIP_CHECK_ARG(g && a && att && ga && gatt && B > 0 && H > 0 && W > 0 && C > 0, "mul_rows_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "mul_rows_bwd: unknown dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(mul_rows_bwd_kernel<float>, dim3(B * H), dim3(256), 0, s, (const float*)g, ldg, (const float*)a, lda, (const float*)att, ldt, (float*)ga, ldga, (float*)gatt, ldgt, W, C);
  else hipLaunchKernelGGL(mul_rows_bwd_kernel<bf16_t>, dim3(B * H), dim3(256), 0, s, (const bf16_t*)g, ldg, (const bf16_t*)a, lda, (const bf16_t*)att, ldt, (bf16_t*)ga, ldga, (bf16_t*)gatt, ldgt, W, C);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_dropout_channels(const void* x, int ldx, void* y, int ldy, const long* seed, long layer_id, int B, int L, int C,
                                      float p, int dtype, void* stream) {
  SDHIP_CHECK_ARG(x && y && seed && B > 0 && L > 0 && C > 0 && p >= 0.f && p < 1.f, "dropout_channels: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "dropout_channels: unknown dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * L * C;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(dropout_channels_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, seed, layer_id, L, C, p, total);
  else hipLaunchKernelGGL(dropout_channels_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, seed, layer_id, L, C, p, total);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
