// Spatial correlation sampler (kernel_size=1, stride=1, padding=0) for gfx950.
// Replaces the third-party `SpatialCorrelationSampler` op used at
// models/dsnet_t2.py:1078-1087,1188-1193,1233-1234 (1-D, patch (1,17)) and
// models/dsnet_t2.py:129-133,221-223 (2-D, patch (17,17)).
//
// The op is HBM/latency bound (AI ~ 4 FLOP/B): one wave64 owns one pixel, its
// lanes own 16-byte channel chunks, the inner product over channels is finished
// with a wave shuffle reduction.  NHWC, so every pixel's channels are one
// contiguous, coalesced read.
#include "sdhip_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

// dot product of one 16-byte chunk pair
template <typename T>
__device__ __forceinline__ float chunk_dot(const u32x4& a, const u32x4& b) {
  float fa[Chunk<T>::N], fb[Chunk<T>::N];
  Chunk<T>::unpack(a, fa);
  Chunk<T>::unpack(b, fb);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < Chunk<T>::N; ++i) s = fmaf(fa[i], fb[i], s);
  return s;
}

// VEC=true : C and ld are multiples of the 16-byte chunk, pointers 16-B aligned.
template <typename T, bool VEC>
__global__ __launch_bounds__(kWavesPerBlock * 64) void corr_fwd_kernel(
    const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out,
    int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);  // B*H*W < 2^31 (host-checked)
  if (pix >= B * H * W) return;  // wave-uniform
  const int w = pix % W;
  const int h = (pix / W) % H;
  const int b = pix / (W * H);
  const T* a_ptr = in1 + (long)pix * ld_in;
  T* o_ptr = out + (long)pix * ld_out;
  constexpr int V = VEC ? Chunk<T>::N : 1;
  const int rh = PH / 2, rw = PW / 2;

  for (int ph = 0; ph < PH; ++ph) {
    const int h2 = h + (ph - rh) * dil;
    for (int pw = 0; pw < PW; ++pw) {
      const int w2 = w + (pw - rw) * dil;
      float acc = 0.f;
      if (h2 >= 0 && h2 < H && w2 >= 0 && w2 < W) {  // wave-uniform
        const T* b_ptr = in2 + (long)((b * H + h2) * W + w2) * ld_in;
        for (int c = lane * V; c < C; c += 64 * V) {
          if constexpr (VEC) {
            u32x4 av = *reinterpret_cast<const u32x4*>(a_ptr + c);
            u32x4 bv = *reinterpret_cast<const u32x4*>(b_ptr + c);
            acc += chunk_dot<T>(av, bv);
          } else {
            acc = fmaf(Elem<T>::ld(a_ptr + c), Elem<T>::ld(b_ptr + c), acc);
          }
        }
        acc = wave_sum(acc);
      }
      if (lane == 0) Elem<T>::st(o_ptr + ph * PW + pw, acc);
    }
  }
}

// One wave per pixel computes both input gradients of that pixel:
//   gin1[pix,c] = sum_p gout[pix,p]       * in2[pix + d(p), c]
//   gin2[pix,c] = sum_p gout[pix - d(p),p] * in1[pix - d(p), c]
template <typename T, bool VEC>
__global__ __launch_bounds__(kWavesPerBlock * 64) void corr_bwd_kernel(
    const T* __restrict__ in1, const T* __restrict__ in2, const T* __restrict__ gout,
    T* __restrict__ gin1, T* __restrict__ gin2,
    int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pix >= B * H * W) return;
  const int w = pix % W;
  const int h = (pix / W) % H;
  const int b = pix / (W * H);
  constexpr int V = VEC ? Chunk<T>::N : 1;
  const int rh = PH / 2, rw = PW / 2;

  for (int c = lane * V; c < C; c += 64 * V) {
    float g1[V], g2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { g1[i] = 0.f; g2[i] = 0.f; }
    for (int ph = 0; ph < PH; ++ph) {
      for (int pw = 0; pw < PW; ++pw) {
        const int dy = (ph - rh) * dil, dx = (pw - rw) * dil;
        const int p = ph * PW + pw;
        // gin1: partner pixel in in2 at +d
        {
          const int h2 = h + dy, w2 = w + dx;
          if (h2 >= 0 && h2 < H && w2 >= 0 && w2 < W) {
            const float g = Elem<T>::ld(gout + (long)pix * ld_out + p);
            const T* src = in2 + (long)((b * H + h2) * W + w2) * ld_in + c;
            if constexpr (VEC) {
              float f[V];
              Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(src), f);
#pragma unroll
              for (int i = 0; i < V; ++i) g1[i] = fmaf(g, f[i], g1[i]);
            } else {
              g1[0] = fmaf(g, Elem<T>::ld(src), g1[0]);
            }
          }
        }
        // gin2: partner pixel in in1 at -d
        {
          const int h1 = h - dy, w1 = w - dx;
          if (h1 >= 0 && h1 < H && w1 >= 0 && w1 < W) {
            const long q = (b * H + h1) * W + w1;
            const float g = Elem<T>::ld(gout + q * ld_out + p);
            const T* src = in1 + q * ld_in + c;
            if constexpr (VEC) {
              float f[V];
              Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(src), f);
#pragma unroll
              for (int i = 0; i < V; ++i) g2[i] = fmaf(g, f[i], g2[i]);
            } else {
              g2[0] = fmaf(g, Elem<T>::ld(src), g2[0]);
            }
          }
        }
      }
    }
    if constexpr (VEC) {
      *reinterpret_cast<u32x4*>(gin1 + (long)pix * ld_in + c) = Chunk<T>::pack(g1);
      *reinterpret_cast<u32x4*>(gin2 + (long)pix * ld_in + c) = Chunk<T>::pack(g2);
    } else {
      Elem<T>::st(gin1 + (long)pix * ld_in + c, g1[0]);
      Elem<T>::st(gin2 + (long)pix * ld_in + c, g2[0]);
    }
  }
}

template <typename T>
bool vec_ok(const void* a, const void* b, const void* c, const void* d, int C, int ld) {
  const int n = Chunk<T>::N;
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
  return (C % n == 0) && (ld % n == 0) && al(a) && al(b) && al(c) && al(d);
}

int check_common(int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out, int dtype) {
  SDHIP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "corr: empty tensor B=%d H=%d W=%d C=%d", B, H, W, C);
  SDHIP_CHECK_ARG(PH > 0 && PW > 0 && (PH & 1) && (PW & 1), "corr: patch size must be odd, got (%d,%d)", PH, PW);
  SDHIP_CHECK_ARG(dil >= 1, "corr: dilation_patch must be >= 1");
  SDHIP_CHECK_ARG(ld_in >= C && ld_out >= PH * PW, "corr: pixel stride smaller than channel count");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "corr: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG((long)B * H * W < (1L << 31), "corr: more than 2^31 pixels");
  return 0;
}

}  // namespace

extern "C" int sdhip_corr_fwd(const void* in1, const void* in2, void* out, int B, int H, int W, int C,
                              int ld_in, int PH, int PW, int dil_patch, int ld_out, int dtype, void* stream) {
  if (int rc = check_common(B, H, W, C, ld_in, PH, PW, dil_patch, ld_out, dtype)) return rc;
  SDHIP_CHECK_ARG(in1 && in2 && out, "corr_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  dim3 grid(sdhip_cdiv(npix, kWavesPerBlock)), block(kWavesPerBlock * 64);
#define LAUNCH(T, VEC) hipLaunchKernelGGL((corr_fwd_kernel<T, VEC>), grid, block, 0, s, (const T*)in1, (const T*)in2, \
                                          (T*)out, B, H, W, C, ld_in, PH, PW, dil_patch, ld_out)
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(in1, in2, nullptr, nullptr, C, ld_in)) LAUNCH(float, true); else LAUNCH(float, false);
  } else {
    if (vec_ok<bf16_t>(in1, in2, nullptr, nullptr, C, ld_in)) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
  }
#undef LAUNCH
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_corr_bwd(const void* in1, const void* in2, const void* gout, void* gin1, void* gin2,
                              int B, int H, int W, int C, int ld_in, int PH, int PW, int dil_patch,
                              int ld_out, int dtype, void* stream) {
  if (int rc = check_common(B, H, W, C, ld_in, PH, PW, dil_patch, ld_out, dtype)) return rc;
  SDHIP_CHECK_ARG(in1 && in2 && gout && gin1 && gin2, "corr_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  dim3 grid(sdhip_cdiv(npix, kWavesPerBlock)), block(kWavesPerBlock * 64);
#define LAUNCH(T, VEC) hipLaunchKernelGGL((corr_bwd_kernel<T, VEC>), grid, block, 0, s, (const T*)in1, (const T*)in2, \
                                          (const T*)gout, (T*)gin1, (T*)gin2, B, H, W, C, ld_in, PH, PW, dil_patch, ld_out)
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(in1, in2, gin1, gin2, C, ld_in)) LAUNCH(float, true); else LAUNCH(float, false);
  } else {
    if (vec_ok<bf16_t>(in1, in2, gin1, gin2, C, ld_in)) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
  }
#undef LAUNCH
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
