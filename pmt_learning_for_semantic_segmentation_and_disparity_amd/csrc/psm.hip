// PSMNet-specific pieces for gfx950 (models_psmnet/stackhourglass.py, submodule.py), all HBM bound:
//
//   * cost volume (stackhourglass.py:110-119): the reference zero-fills a (B,64,48,H/4,W/4) tensor and runs a
//     48-iteration Python loop of strided slice copies; here one kernel writes the [B][D][H][W][2C] volume;
//   * soft-argmin head (stackhourglass.py:138-155, submodule.py:56-64): trilinear x4 upsample -> softmax over the 192
//     disparities -> sum_d p_d * d.  The reference materialises the (B,192,H,W) tensor four times per head; here one
//     kernel evaluates the 192 interpolated costs of a pixel on the fly (online softmax), forward and backward;
//   * zero insertion / strided gather: stride-s transposed convolutions (ConvTranspose3d of the hourglass,
//     stackhourglass.py:25-29) and the data gradient of stride-s convolutions run as stride-1 convolutions over a
//     zero-stuffed tensor.
#include "sdhip_common.h"

namespace {

struct Vol { int N, D, H, W, C, ld; };   // [N][D][H][W] pixels of C channels, pixel stride ld

inline dim3 grid_for(long items) {
  long b = (items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

template <typename T>
bool vec_ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  const int n = Chunk<T>::N;
  if (C % n) return false;
  for (int l : lds) if (l % n) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

// dense[n,d,h,w,:] <-> stuffed[n, d*sd, h*s, w*s, :]   (SCATTER: dense -> stuffed, stuffed pre-zeroed; else gather)
template <typename T, bool VEC, bool SCATTER>
__global__ __launch_bounds__(256) void stuff_kernel(const T* __restrict__ src, T* __restrict__ dst, Vol dense, Vol stuffed, int sd, int s) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = dense.C / N;
  const long npix = (long)dense.N * dense.D * dense.H * dense.W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix * units; i += (long)gridDim.x * 256) {
    const long pix = i / units;
    const int c0 = (int)(i - pix * units) * N;
    const int w = (int)(pix % dense.W);
    long r = pix / dense.W;
    const int h = (int)(r % dense.H); r /= dense.H;
    const int d = (int)(r % dense.D);
    const long n = r / dense.D;
    const long spix = ((n * stuffed.D + (long)d * sd) * stuffed.H + (long)h * s) * stuffed.W + (long)w * s;
    float f[N];
    if (SCATTER) {
      Unit<T, VEC>::load(src + pix * dense.ld + c0, f);
      Unit<T, VEC>::store(dst + spix * stuffed.ld + c0, f);
    } else {
      Unit<T, VEC>::load(src + spix * stuffed.ld + c0, f);
      Unit<T, VEC>::store(dst + pix * dense.ld + c0, f);
    }
  }
}

// ---------------------------------------------------------------- cost volume
// vol[b,i,h,w,0:C] = L[b,h,w,:], vol[b,i,h,w,C:2C] = R[b,h,w-i,:]   for w >= i, else 0
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void cost_volume_fwd(const T* __restrict__ L, const T* __restrict__ R, int ldl, T* __restrict__ vol,
                                                       int B, int D, int H, int W, int C) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = 2 * C / N;
  const long nvox = (long)B * D * H * W;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < nvox * units; it += (long)gridDim.x * 256) {
    const long vox = it / units;
    const int c0 = (int)(it - vox * units) * N;
    const int w = (int)(vox % W);
    long r = vox / W;
    const int h = (int)(r % H); r /= H;
    const int i = (int)(r % D);
    const long b = r / D;
    float f[N];
#pragma unroll
    for (int e = 0; e < N; ++e) f[e] = 0.f;
    if (w >= i) {
      if (c0 < C) Unit<T, VEC>::load(L + ((b * H + h) * W + w) * ldl + c0, f);
      else Unit<T, VEC>::load(R + ((b * H + h) * W + (w - i)) * ldl + (c0 - C), f);
    }
    Unit<T, VEC>::store(vol + vox * (2 * C) + c0, f);
  }
}

// gL[b,h,w,:] = sum_{i <= w} g[b,i,h,w,0:C];  gR[b,h,w,:] = sum_{i: w+i < W} g[b,i,h,w+i,C:2C]
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void cost_volume_bwd(const T* __restrict__ g, T* __restrict__ gL, T* __restrict__ gR, int ldl,
                                                       int B, int D, int H, int W, int C) {
  constexpr int N = Unit<T, VEC>::N;
  const int units = C / N;
  const long npix = (long)B * H * W;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < npix * units; it += (long)gridDim.x * 256) {
    const long pix = it / units;
    const int c0 = (int)(it - pix * units) * N;
    const int w = (int)(pix % W);
    const long bh = pix / W;                // b*H + h
    const long b = bh / H;
    const int h = (int)(bh - b * H);
    float aL[N], aR[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { aL[e] = 0.f; aR[e] = 0.f; }
    for (int i = 0; i < D; ++i) {
      const long base = ((b * D + i) * H + h) * W;
      float f[N];
      if (i <= w) {
        Unit<T, VEC>::load(g + (base + w) * (2 * C) + c0, f);
#pragma unroll
        for (int e = 0; e < N; ++e) aL[e] += f[e];
      }
      if (w + i < W) {
        Unit<T, VEC>::load(g + (base + w + i) * (2 * C) + C + c0, f);
#pragma unroll
        for (int e = 0; e < N; ++e) aR[e] += f[e];
      }
    }
    Unit<T, VEC>::store(gL + pix * ldl + c0, aL);
    Unit<T, VEC>::store(gR + pix * ldl + c0, aR);
  }
}

// ---------------------------------------------------------------- soft-argmin head
struct Lin { int i0, i1; float l1; };
__device__ __forceinline__ Lin lin_src(int d, float scale, int in) {   // ATen area_pixel source index, align_corners=False
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  Lin r;
  r.i0 = (int)s;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = fminf(fmaxf(s - (float)r.i0, 0.f), 1.f);
  return r;
}

constexpr int kMaxD4 = 56;   // low-resolution disparity levels kept per lane in LDS (PSMNet: maxdisp/4 = 48)

// pred[b,h,w] = sum_d softmax_d(trilinear(cost)[b,d,h,w]) * d ;  cost: [B][D4][H4][W4] (1 channel)
template <typename T>
__global__ __launch_bounds__(256) void softargmin_fwd(const T* __restrict__ cost, T* __restrict__ pred,
                                                      int B, int D4, int H4, int W4, int Dout, int H, int W) {
  __shared__ float cs[kMaxD4][256];
  const float sd = (float)D4 / (float)Dout, shh = (float)H4 / (float)H, sw = (float)W4 / (float)W;
  const long npix = (long)B * H * W;
  for (long p0 = (long)blockIdx.x * 256; p0 < npix; p0 += (long)gridDim.x * 256) {
    const long p = p0 + threadIdx.x;
    const bool live = p < npix;
    const int w = live ? (int)(p % W) : 0;
    const int h = live ? (int)((p / W) % H) : 0;
    const long b = live ? p / ((long)W * H) : 0;
    const Lin lh = lin_src(h, shh, H4), lw = lin_src(w, sw, W4);
    const float w00 = (1.f - lh.l1) * (1.f - lw.l1), w01 = (1.f - lh.l1) * lw.l1, w10 = lh.l1 * (1.f - lw.l1), w11 = lh.l1 * lw.l1;
    for (int d4 = 0; d4 < D4; ++d4) {
      const T* c = cost + ((b * D4 + d4) * H4) * (long)W4;
      cs[d4][threadIdx.x] = w00 * Elem<T>::ld(c + lh.i0 * W4 + lw.i0) + w01 * Elem<T>::ld(c + lh.i0 * W4 + lw.i1) +
                            w10 * Elem<T>::ld(c + lh.i1 * W4 + lw.i0) + w11 * Elem<T>::ld(c + lh.i1 * W4 + lw.i1);
    }
    float m = -INFINITY, s = 0.f, e = 0.f;
    for (int d = 0; d < Dout; ++d) {
      const Lin ld = lin_src(d, sd, D4);
      const float v = (1.f - ld.l1) * cs[ld.i0][threadIdx.x] + ld.l1 * cs[ld.i1][threadIdx.x];
      if (v > m) { const float k = __expf(m - v); s *= k; e *= k; m = v; }
      const float x = __expf(v - m);
      s += x; e = fmaf(x, (float)d, e);
    }
    if (live) Elem<T>::st(pred + p, e / s);
  }
}

// gcost (f32, pre-zeroed, [B][D4][H4][W4]) += d pred / d cost * g
// One workgroup = an 8 x 32 tile of full-resolution pixels (one per lane).  Their gradients with respect to the
// low-resolution cost levels land on a small footprint of low-resolution cells ((8*sh+3) x (32*sw+3) cells, 5 x 11 at
// PSMNet's 1/4 scale): they are summed in LDS (ds_add_f32) and flushed once, contiguous lanes along w4 — ~50x fewer
// global float atomics than one scattered 4-byte atomic per (pixel, level, bilinear neighbour), which cost 2 ms per call.
constexpr int kSaTH = 8, kSaTW = 32, kSaCells = 96;
template <typename T>
__global__ __launch_bounds__(256) void softargmin_bwd(const T* __restrict__ cost, const T* __restrict__ g, float* __restrict__ gcost,
                                                      int B, int D4, int H4, int W4, int Dout, int H, int W) {
  extern __shared__ float sa_dyn[];
  float (*cs)[256] = reinterpret_cast<float (*)[256]>(sa_dyn);            // [kMaxD4][256] interpolated low-res column per lane
  float* acc = sa_dyn + kMaxD4 * 256;                                     // [D4][kSaCells] gradient footprint of the tile
  const float sd = (float)D4 / (float)Dout, shh = (float)H4 / (float)H, sw = (float)W4 / (float)W;
  const int tiles_w = (W + kSaTW - 1) / kSaTW, tiles_h = (H + kSaTH - 1) / kSaTH;
  const long ntiles = (long)B * tiles_h * tiles_w;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long b = tile / ((long)tiles_h * tiles_w);
    const int tr = (int)(tile - b * tiles_h * tiles_w);
    const int h0 = (tr / tiles_w) * kSaTH, w0 = (tr % tiles_w) * kSaTW;
    const int h1 = min(h0 + kSaTH, H) - 1, w1 = min(w0 + kSaTW, W) - 1;
    // footprint of the tile on the low-resolution grid (source indices are monotone in the pixel coordinate)
    const int hb = lin_src(h0, shh, H4).i0, he = lin_src(h1, shh, H4).i1;
    const int wb = lin_src(w0, sw, W4).i0, we = lin_src(w1, sw, W4).i1;
    const int ch = he - hb + 1, cw = we - wb + 1, ncell = ch * cw;
    const bool lds_ok = ncell <= kSaCells;                                 // uniform; odd scale factors fall back to global atomics
    if (lds_ok) for (int i = threadIdx.x; i < D4 * ncell; i += 256) acc[i] = 0.f;
    const int h = h0 + threadIdx.x / kSaTW, w = w0 + threadIdx.x % kSaTW;
    const bool live = h < H && w < W;
    const long p = (b * H + (live ? h : h0)) * (long)W + (live ? w : w0);
    const Lin lh = lin_src(live ? h : h0, shh, H4), lw = lin_src(live ? w : w0, sw, W4);
    const float w00 = (1.f - lh.l1) * (1.f - lw.l1), w01 = (1.f - lh.l1) * lw.l1, w10 = lh.l1 * (1.f - lw.l1), w11 = lh.l1 * lw.l1;
    for (int d4 = 0; d4 < D4; ++d4) {
      const T* c = cost + ((b * D4 + d4) * H4) * (long)W4;
      cs[d4][threadIdx.x] = w00 * Elem<T>::ld(c + lh.i0 * W4 + lw.i0) + w01 * Elem<T>::ld(c + lh.i0 * W4 + lw.i1) +
                            w10 * Elem<T>::ld(c + lh.i1 * W4 + lw.i0) + w11 * Elem<T>::ld(c + lh.i1 * W4 + lw.i1);
    }
    float m = -INFINITY, s = 0.f, e = 0.f;
    for (int d = 0; d < Dout; ++d) {
      const Lin ld = lin_src(d, sd, D4);
      const float v = (1.f - ld.l1) * cs[ld.i0][threadIdx.x] + ld.l1 * cs[ld.i1][threadIdx.x];
      if (v > m) { const float k = __expf(m - v); s *= k; e *= k; m = v; }
      const float x = __expf(v - m);
      s += x; e = fmaf(x, (float)d, e);
    }
    const float predv = e / s, inv = 1.f / s;
    const float gv = live ? Elem<T>::ld(g + p) : 0.f;
    __syncthreads();   // acc is zeroed
    const int c00 = (lh.i0 - hb) * cw + (lw.i0 - wb), c01 = (lh.i0 - hb) * cw + (lw.i1 - wb);
    const int c10 = (lh.i1 - hb) * cw + (lw.i0 - wb), c11 = (lh.i1 - hb) * cw + (lw.i1 - wb);
    auto scatter = [&](int level, float a) {
      if (!live || a == 0.f) return;
      if (lds_ok) {
        float* al = acc + level * ncell;
        atomicAdd(al + c00, w00 * a); atomicAdd(al + c01, w01 * a); atomicAdd(al + c10, w10 * a); atomicAdd(al + c11, w11 * a);
      } else {
        float* gc = gcost + ((b * D4 + level) * H4) * (long)W4;
        atomicAdd(gc + lh.i0 * W4 + lw.i0, w00 * a); atomicAdd(gc + lh.i0 * W4 + lw.i1, w01 * a);
        atomicAdd(gc + lh.i1 * W4 + lw.i0, w10 * a); atomicAdd(gc + lh.i1 * W4 + lw.i1, w11 * a);
      }
    };
    // d pred / d v_d = p_d (d - pred); fold the depth interpolation back onto the D4 levels.  The gradients of levels
    // `cur` and `cur + 1` are accumulated in registers (i0 is monotone in d) and scattered when a level is complete.
    float acc0 = 0.f, acc1 = 0.f;
    int cur = 0;
    for (int d = 0; d < Dout; ++d) {
      const Lin ld = lin_src(d, sd, D4);
      const float v = (1.f - ld.l1) * cs[ld.i0][threadIdx.x] + ld.l1 * cs[ld.i1][threadIdx.x];
      const float dv = gv * __expf(v - m) * inv * ((float)d - predv);
      while (cur < ld.i0) { scatter(cur, acc0); acc0 = acc1; acc1 = 0.f; ++cur; }
      acc0 += (1.f - ld.l1) * dv;
      if (ld.i1 != ld.i0) acc1 += ld.l1 * dv; else acc0 += ld.l1 * dv;
    }
    for (int k = 0; k < 2; ++k) {   // flush the last two levels
      if (cur < D4) scatter(cur, acc0);
      acc0 = acc1; acc1 = 0.f; ++cur;
    }
    __syncthreads();
    if (lds_ok) {
      for (int i = threadIdx.x; i < D4 * ncell; i += 256) {
        const float v = acc[i];
        if (v != 0.f) {
          const int level = i / ncell, cell = i - level * ncell;
          const int hh = hb + cell / cw, ww = wb + cell % cw;
          atomicAdd(gcost + (((b * D4 + level) * H4 + hh) * (long)W4 + ww), v);
        }
      }
    }
    __syncthreads();   // acc / cs are reused by the next tile
  }
}

// ---------------------------------------------------------------- soft-argmin, x4 form (Dout = 4 * D4: PSMNet's 1/4-scale volume)
// With an exact factor 4 the align_corners=False depth interpolation has four fixed weight pairs: for d = 4k + r
//   v(4k)   = c[k] + 0.375 (c[k-1] - c[k]),  v(4k+1) = c[k] + 0.125 (c[k-1] - c[k]),
//   v(4k+2) = c[k] + 0.125 (c[k+1] - c[k]),  v(4k+3) = c[k] + 0.375 (c[k+1] - c[k]),      c[-1] := c[0], c[D4] := c[D4-1]
// (ATen's area_pixel source index (d + 0.5) / 4 - 0.5, clamped at 0), so the 192-level loops need no index arithmetic and
// no branches: one pass for the maximum, one for the two sums.  The general kernels above spend most of their time in
// lin_src() and the branchy online softmax (238 us forward, 1.34 ms backward per head at B = 8, 256 x 512).
template <typename T>
__device__ __forceinline__ void sa_column(const T* __restrict__ cost, float (*cs)[256], long b, int D4, int H4, int W4,
                                          const Lin& lh, const Lin& lw) {
  const float w00 = (1.f - lh.l1) * (1.f - lw.l1), w01 = (1.f - lh.l1) * lw.l1, w10 = lh.l1 * (1.f - lw.l1), w11 = lh.l1 * lw.l1;
  const int o00 = lh.i0 * W4 + lw.i0, o01 = lh.i0 * W4 + lw.i1, o10 = lh.i1 * W4 + lw.i0, o11 = lh.i1 * W4 + lw.i1;
  const T* c = cost + (b * D4) * (long)H4 * W4;
  const int plane = H4 * W4;
#pragma unroll 4
  for (int d4 = 0; d4 < D4; ++d4, c += plane)
    cs[d4][threadIdx.x] = w00 * Elem<T>::ld(c + o00) + w01 * Elem<T>::ld(c + o01) + w10 * Elem<T>::ld(c + o10) + w11 * Elem<T>::ld(c + o11);
}

// (log-sum-exp, expectation) of the 4 * D4 interpolated levels of the lane's column cs[.][tid]
__device__ __forceinline__ void sa_reduce4(float (*cs)[256], int D4, float& lse, float& pred) {
  const int t = threadIdx.x;
  float m = -INFINITY;
  {
    float cm = cs[0][t], c0 = cm;
    for (int k = 0; k < D4; ++k) {
      const float cp = cs[k + 1 < D4 ? k + 1 : k][t];
      const float dm = cm - c0, dp = cp - c0;
      m = fmaxf(m, fmaxf(fmaxf(fmaf(0.375f, dm, c0), fmaf(0.125f, dm, c0)), fmaxf(fmaf(0.125f, dp, c0), fmaf(0.375f, dp, c0))));
      cm = c0; c0 = cp;
    }
  }
  float s = 0.f, e = 0.f;
  {
    float cm = cs[0][t], c0 = cm;
    for (int k = 0; k < D4; ++k) {
      const float cp = cs[k + 1 < D4 ? k + 1 : k][t];
      const float dm = cm - c0, dp = cp - c0, base = c0 - m;
      const float x0 = __expf(fmaf(0.375f, dm, base)), x1 = __expf(fmaf(0.125f, dm, base));
      const float x2 = __expf(fmaf(0.125f, dp, base)), x3 = __expf(fmaf(0.375f, dp, base));
      const float d0 = (float)(4 * k);
      s += (x0 + x1) + (x2 + x3);
      e += fmaf(x0, d0, x1 * (d0 + 1.f)) + fmaf(x2, d0 + 2.f, x3 * (d0 + 3.f));
      cm = c0; c0 = cp;
    }
  }
  lse = m + __logf(s);
  pred = e / s;
}

template <typename T>
__global__ __launch_bounds__(256) void softargmin4_fwd(const T* __restrict__ cost, T* __restrict__ pred, float* __restrict__ stats,
                                                       int B, int D4, int H4, int W4, int H, int W) {
  __shared__ float cs[kMaxD4][256];
  const float shh = (float)H4 / (float)H, sw = (float)W4 / (float)W;
  const long npix = (long)B * H * W;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < npix;
  const int w = live ? (int)(p % W) : 0;
  const int h = live ? (int)((p / W) % H) : 0;
  const long b = live ? p / ((long)W * H) : 0;
  sa_column<T>(cost, cs, b, D4, H4, W4, lin_src(h, shh, H4), lin_src(w, sw, W4));
  float lse, pv;
  sa_reduce4(cs, D4, lse, pv);
  if (live) {
    Elem<T>::st(pred + p, pv);
    if (stats) { stats[2 * p] = lse; stats[2 * p + 1] = pv; }
  }
}

// Backward, stage 1: per full-resolution pixel, the gradient folded back onto the D4 low-resolution LEVELS of its own column,
//   t[b, k, h, w] = sum_d Wd[d, k] * g * p_d * (d - pred),
// written [B][D4][H][W] in the activations' own type (lanes along w: coalesced; bf16 halves the 200 MB this intermediate
// takes at B = 8 — its 64 contributions per cell are summed in f32 by stage 2, and the result is rounded to bf16 anyway).
// Stage 2 gathers it through the adjoint of the bilinear map.
template <typename T>
__global__ __launch_bounds__(256) void softargmin4_bwd_levels(const T* __restrict__ cost, const T* __restrict__ g,
                                                              const float* __restrict__ stats, T* __restrict__ tl,
                                                              int B, int D4, int H4, int W4, int H, int W) {
  __shared__ float cs[kMaxD4][256];
  const float shh = (float)H4 / (float)H, sw = (float)W4 / (float)W;
  const long npix = (long)B * H * W;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < npix;
  const int w = live ? (int)(p % W) : 0;
  const int h = live ? (int)((p / W) % H) : 0;
  const long b = live ? p / ((long)W * H) : 0;
  sa_column<T>(cost, cs, b, D4, H4, W4, lin_src(h, shh, H4), lin_src(w, sw, W4));
  float lse, pv;
  if (stats) { lse = stats[2 * (live ? p : 0)]; pv = stats[2 * (live ? p : 0) + 1]; }
  else sa_reduce4(cs, D4, lse, pv);
  const float gv = live ? Elem<T>::ld(g + p) : 0.f;
  const int t = threadIdx.x;
  T* dst = tl + ((b * D4) * H + h) * (long)W + w;
  const long plane = (long)H * W;
  float cm = cs[0][t], c0 = cm;
  float tm = 0.f, t0 = 0.f;      // gradients of levels k - 1 and k gathered so far
  for (int k = 0; k < D4; ++k) {
    const float cp = cs[k + 1 < D4 ? k + 1 : k][t];
    const float dm = cm - c0, dp = cp - c0, base = c0 - lse;
    const float d0 = (float)(4 * k) - pv;
    const float g0 = gv * __expf(fmaf(0.375f, dm, base)) * d0, g1 = gv * __expf(fmaf(0.125f, dm, base)) * (d0 + 1.f);
    const float g2 = gv * __expf(fmaf(0.125f, dp, base)) * (d0 + 2.f), g3 = gv * __expf(fmaf(0.375f, dp, base)) * (d0 + 3.f);
    const float lo = fmaf(0.375f, g0, 0.125f * g1), hi = fmaf(0.125f, g2, 0.375f * g3);
    float mid = fmaf(0.625f, g0 + g3, 0.875f * (g1 + g2));
    float tp = hi;
    if (k == 0) mid += lo; else tm += lo;                       // c[-1] is c[0]
    if (k == D4 - 1) { mid += hi; tp = 0.f; }                   // c[D4] is c[D4-1]
    t0 += mid;
    if (k > 0 && live) Elem<T>::st(dst + (long)(k - 1) * plane, tm);   // level k - 1 is complete
    tm = t0; t0 = tp;
    cm = c0; c0 = cp;
  }
  if (live) Elem<T>::st(dst + (long)(D4 - 1) * plane, tm);
}

// Backward, stage 2: gcost[b, k, h4, w4] = sum over the pixels (h, w) whose bilinear footprint contains (h4, w4) of
// Wh[h, h4] * Ww[w, w4] * t[b, k, h, w] — one lane per low-resolution cell, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void softargmin_bwd_cells(const T* __restrict__ tl, T* __restrict__ gcost,
                                                            int B, int D4, int H4, int W4, int H, int W) {
  const float shh = (float)H4 / (float)H, sw = (float)W4 / (float)W;
  // a workgroup = 8 x 32 cells of one (b, k) plane: neighbouring cells share half of their footprints, which then meet in
  // the workgroup's L1 instead of being fetched again by a workgroup far away in time
  const int tw = (W4 + 31) >> 5, th = (H4 + 7) >> 3;
  int t = blockIdx.x;
  const int tx = t % tw; t /= tw;
  const int ty = t % th;
  const long bk = t / th;                                        // b * D4 + k
  const int w4 = tx * 32 + (threadIdx.x & 31), h4 = ty * 8 + (threadIdx.x >> 5);
  if (w4 >= W4 || h4 >= H4) return;
  const long i = (bk * H4 + h4) * (long)W4 + w4;
  // pixels whose source coordinate lies within one cell of (h4, w4); the exact weights below are zero for the rest
  const int hlo = max(0, (int)floorf(((float)h4 - 0.5f) / shh - 0.5f)), hhi = min(H - 1, (int)ceilf(((float)h4 + 1.5f) / shh - 0.5f));
  const int wlo = max(0, (int)floorf(((float)w4 - 0.5f) / sw - 0.5f)), whi = min(W - 1, (int)ceilf(((float)w4 + 1.5f) / sw - 0.5f));
  const T* src = tl + bk * (long)H * W;
  float acc = 0.f;
  constexpr int NC = 12;                                         // candidates per axis held in registers (x4 scale: 10)
  auto weight = [](int pix, float scale, int in, int cell) {
    const Lin l = lin_src(pix, scale, in);
    return (l.i0 == cell ? 1.f - l.l1 : 0.f) + (l.i1 == cell ? l.l1 : 0.f);
  };
  if (hhi - hlo < NC && whi - wlo < NC) {
    // the column weights are the same for every row: computed once (the weight search was most of this kernel's time)
    float ww[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) ww[j] = wlo + j <= whi ? weight(wlo + j, sw, W4, w4) : 0.f;
#pragma unroll
    for (int r = 0; r < NC; ++r) {
      const int h = hlo + r;
      const float wh = h <= hhi ? weight(h, shh, H4, h4) : 0.f;
      if (wh != 0.f) {
        const T* row = src + (long)h * W + wlo;
        float rs = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
          if (ww[j] != 0.f) rs = fmaf(ww[j], Elem<T>::ld(row + j), rs);
        acc = fmaf(wh, rs, acc);
      }
    }
  } else {
    for (int h = hlo; h <= hhi; ++h) {
      const float wh = weight(h, shh, H4, h4);
      if (wh == 0.f) continue;
      float row = 0.f;
      for (int w = wlo; w <= whi; ++w) row = fmaf(weight(w, sw, W4, w4), Elem<T>::ld(src + (long)h * W + w), row);
      acc = fmaf(wh, row, acc);
    }
  }
  Elem<T>::st(gcost + i, acc);
}

template <typename T>
__global__ void f32_to_T_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) Elem<T>::st(dst + i, src[i]);
}

}  // namespace

extern "C" int sdhip_stuff(const void* src, int lds_, void* dst, int ldd, int N, int D, int H, int W, int C,
                           int sd, int s, int scatter, int dtype, void* stream) {
  SDHIP_CHECK_ARG(src && dst && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && sd >= 1 && s >= 1, "stuff: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "stuff: unknown dtype %d", dtype);
  const int Ds = (D - 1) * sd + 1, Hs = (H - 1) * s + 1, Ws = (W - 1) * s + 1;
  hipStream_t st = (hipStream_t)stream;
  // scatter: src dense (ld lds_), dst stuffed (ld ldd, zeroed here);  gather: src stuffed (ld lds_), dst dense (ld ldd)
  Vol dense{N, D, H, W, C, scatter ? lds_ : ldd}, stuffed{N, Ds, Hs, Ws, C, scatter ? ldd : lds_};
  SDHIP_CHECK_ARG(dense.ld >= C && stuffed.ld >= C, "stuff: pixel stride smaller than channel count");
  const long np = (long)N * D * H * W;
  const int es = dtype == SDHIP_BF16 ? 2 : 4;
  if (scatter) {
    SDHIP_CHECK_ARG(ldd == C, "stuff: the zero-stuffed output must be dense");
    if (sdhip_zero_async(dst, (size_t)N * Ds * Hs * Ws * C * es, st) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "stuff: memset failed");
  }
#define GO(T, V) do { if (scatter) hipLaunchKernelGGL((stuff_kernel<T, V, true>), grid_for(np * (V ? C / Chunk<T>::N : C)), dim3(256), 0, st, (const T*)src, (T*)dst, dense, stuffed, sd, s); \
                      else hipLaunchKernelGGL((stuff_kernel<T, V, false>), grid_for(np * (V ? C / Chunk<T>::N : C)), dim3(256), 0, st, (const T*)src, (T*)dst, dense, stuffed, sd, s); } while (0)
  if (dtype == SDHIP_F32) { if (vec_ok<float>(C, {lds_, ldd}, {src, dst})) GO(float, true); else GO(float, false); }
  else { if (vec_ok<bf16_t>(C, {lds_, ldd}, {src, dst})) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_cost_volume_fwd(const void* left, const void* right, int ld, void* vol, int B, int D, int H, int W, int C,
                                     int dtype, void* stream) {
  SDHIP_CHECK_ARG(left && right && vol && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && ld >= C, "cost_volume_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "cost_volume_fwd: unknown dtype %d", dtype);
  const long nv = (long)B * D * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(C, {ld}, {left, right, vol})) hipLaunchKernelGGL((cost_volume_fwd<float, true>), grid_for(nv * (2 * C / 4)), dim3(256), 0, st, (const float*)left, (const float*)right, ld, (float*)vol, B, D, H, W, C);
    else hipLaunchKernelGGL((cost_volume_fwd<float, false>), grid_for(nv * 2 * C), dim3(256), 0, st, (const float*)left, (const float*)right, ld, (float*)vol, B, D, H, W, C);
  } else {
    if (vec_ok<bf16_t>(C, {ld}, {left, right, vol})) hipLaunchKernelGGL((cost_volume_fwd<bf16_t, true>), grid_for(nv * (2 * C / 8)), dim3(256), 0, st, (const bf16_t*)left, (const bf16_t*)right, ld, (bf16_t*)vol, B, D, H, W, C);
    else hipLaunchKernelGGL((cost_volume_fwd<bf16_t, false>), grid_for(nv * 2 * C), dim3(256), 0, st, (const bf16_t*)left, (const bf16_t*)right, ld, (bf16_t*)vol, B, D, H, W, C);
  }
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_cost_volume_bwd(const void* gvol, void* gleft, void* gright, int ld, int B, int D, int H, int W, int C,
                                     int dtype, void* stream) {
  SDHIP_CHECK_ARG(gvol && gleft && gright && B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && ld >= C, "cost_volume_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "cost_volume_bwd: unknown dtype %d", dtype);
  const long np = (long)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(C, {ld}, {gvol, gleft, gright})) hipLaunchKernelGGL((cost_volume_bwd<float, true>), grid_for(np * (C / 4)), dim3(256), 0, st, (const float*)gvol, (float*)gleft, (float*)gright, ld, B, D, H, W, C);
    else hipLaunchKernelGGL((cost_volume_bwd<float, false>), grid_for(np * C), dim3(256), 0, st, (const float*)gvol, (float*)gleft, (float*)gright, ld, B, D, H, W, C);
  } else {
    if (vec_ok<bf16_t>(C, {ld}, {gvol, gleft, gright})) hipLaunchKernelGGL((cost_volume_bwd<bf16_t, true>), grid_for(np * (C / 8)), dim3(256), 0, st, (const bf16_t*)gvol, (bf16_t*)gleft, (bf16_t*)gright, ld, B, D, H, W, C);
    else hipLaunchKernelGGL((cost_volume_bwd<bf16_t, false>), grid_for(np * C), dim3(256), 0, st, (const bf16_t*)gvol, (bf16_t*)gleft, (bf16_t*)gright, ld, B, D, H, W, C);
  }
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_softargmin_fwd(const void* cost, void* pred, float* stats, int B, int D4, int H4, int W4, int Dout, int H, int W,
                                    int dtype, void* stream) {
  SDHIP_CHECK_ARG(cost && pred && B > 0 && D4 > 0 && D4 <= kMaxD4 && H4 > 0 && W4 > 0 && Dout > 0 && H > 0 && W > 0,
                  "softargmin_fwd: bad arguments (at most %d low-resolution disparity levels)", kMaxD4);
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "softargmin_fwd: unknown dtype %d", dtype);
  const long np = (long)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (Dout == 4 * D4) {
    const dim3 grid((unsigned)((np + 255) / 256));
    if (dtype == SDHIP_F32) hipLaunchKernelGGL(softargmin4_fwd<float>, grid, dim3(256), 0, st, (const float*)cost, (float*)pred, stats, B, D4, H4, W4, H, W);
    else hipLaunchKernelGGL(softargmin4_fwd<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)cost, (bf16_t*)pred, stats, B, D4, H4, W4, H, W);
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  SDHIP_CHECK_ARG(!stats, "softargmin_fwd: per-pixel statistics are produced for Dout = 4 * D4 only");
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(softargmin_fwd<float>, grid_for(np), dim3(256), 0, st, (const float*)cost, (float*)pred, B, D4, H4, W4, Dout, H, W);
  else hipLaunchKernelGGL(softargmin_fwd<bf16_t>, grid_for(np), dim3(256), 0, st, (const bf16_t*)cost, (bf16_t*)pred, B, D4, H4, W4, Dout, H, W);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" long sdhip_softargmin_bwd_workspace_floats(int B, int D4, int H4, int W4, int Dout, int H, int W) {
  const long levels = (long)B * D4 * H * W, cells = (long)B * D4 * H4 * W4;
  return Dout == 4 * D4 ? levels : cells;
}

extern "C" int sdhip_softargmin_bwd(const void* cost, const void* gpred, const float* stats, void* gcost, float* workspace,
                                    long workspace_floats, int B, int D4, int H4, int W4, int Dout, int H, int W, int dtype, void* stream) {
  SDHIP_CHECK_ARG(cost && gpred && gcost && workspace && B > 0 && D4 > 0 && D4 <= kMaxD4 && H4 > 0 && W4 > 0 && Dout > 0 && H > 0 && W > 0,
                  "softargmin_bwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "softargmin_bwd: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG(workspace_floats >= sdhip_softargmin_bwd_workspace_floats(B, D4, H4, W4, Dout, H, W),
                  "softargmin_bwd: workspace too small (%ld floats)", workspace_floats);
  hipStream_t st = (hipStream_t)stream;
  const long nv = (long)B * D4 * H4 * W4, np = (long)B * H * W;
  if (Dout == 4 * D4) {
    // two stages, no atomics: per-pixel gradients of the D4 levels (f32, [B][D4][H][W]), then the bilinear adjoint per cell
    const dim3 g1((unsigned)((np + 255) / 256)), g2((unsigned)((long)B * D4 * sdhip_cdiv(H4, 8) * sdhip_cdiv(W4, 32)));
    (void)nv;
    if (dtype == SDHIP_F32) {
      hipLaunchKernelGGL(softargmin4_bwd_levels<float>, g1, dim3(256), 0, st, (const float*)cost, (const float*)gpred, stats, workspace, B, D4, H4, W4, H, W);
      hipLaunchKernelGGL(softargmin_bwd_cells<float>, g2, dim3(256), 0, st, (const float*)workspace, (float*)gcost, B, D4, H4, W4, H, W);
    } else {
      hipLaunchKernelGGL(softargmin4_bwd_levels<bf16_t>, g1, dim3(256), 0, st, (const bf16_t*)cost, (const bf16_t*)gpred, stats, (bf16_t*)workspace, B, D4, H4, W4, H, W);
      hipLaunchKernelGGL(softargmin_bwd_cells<bf16_t>, g2, dim3(256), 0, st, (const bf16_t*)workspace, (bf16_t*)gcost, B, D4, H4, W4, H, W);
    }
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  float* gcost_f32 = workspace;
  if (sdhip_zero_async(gcost_f32, nv * sizeof(float), st) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "softargmin_bwd: memset failed");
  const size_t sa_lds = ((size_t)kMaxD4 * 256 + (size_t)kMaxD4 * kSaCells) * sizeof(float);
  static bool sa_attr = false;
  if (!sa_attr) {
    if (hipFuncSetAttribute((const void*)softargmin_bwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)softargmin_bwd<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sa_lds) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "softargmin_bwd: cannot raise dynamic LDS limit");
    sa_attr = true;
  }
  const long sa_tiles = (long)B * sdhip_cdiv(H, kSaTH) * sdhip_cdiv(W, kSaTW);
  const int sa_grid = (int)(sa_tiles < 2048 ? sa_tiles : 2048);
  if (dtype == SDHIP_F32) {
    hipLaunchKernelGGL(softargmin_bwd<float>, dim3(sa_grid), dim3(256), sa_lds, st, (const float*)cost, (const float*)gpred, gcost_f32, B, D4, H4, W4, Dout, H, W);
    hipLaunchKernelGGL(f32_to_T_kernel<float>, grid_for(nv), dim3(256), 0, st, gcost_f32, (float*)gcost, nv);
  } else {
    hipLaunchKernelGGL(softargmin_bwd<bf16_t>, dim3(sa_grid), dim3(256), sa_lds, st, (const bf16_t*)cost, (const bf16_t*)gpred, gcost_f32, B, D4, H4, W4, Dout, H, W);
    hipLaunchKernelGGL(f32_to_T_kernel<bf16_t>, grid_for(nv), dim3(256), 0, st, gcost_f32, (bf16_t*)gcost, nv);
  }
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
