// Thin convolutions: Cin <= 8 (one 16-byte chunk per pixel) -> ONE output channel, stride 1 — the image side-branches
// `convbn(cin, 1, 5, 1, 'same', 2)` of models/dsnet_t2.py:1105-1112 at full resolution.  16 useful MFMA rows out of
// 256 and a halo tile per 256 pixels make the matrix-core kernels latency-bound here (66 us forward, 168 us weight gradient
// for 17 MB of input); on the vector ALUs the whole layer is 200 FMAs per pixel and streams at HBM speed.
#pragma once
#include "conv_common.h"

namespace {

constexpr int kThinMaxT = 49;

struct ThinArgs {
  const void* x; const void* wp; void* y; const float* bias; double* stats;
  int B, H, W, Ho, Wo, kh, kw, dil, pad_t, pad_l;
  int Cin, ldx, ldy, Mpad, act, bpg, stats_ld, nrep;
  long rep_stride;
};

// one thread per output pixel; packed weights [t][Mpad][CK]: row (t, m = 0) holds the Cin weights of tap t
template <typename T>
__global__ __launch_bounds__(256) void conv_thin_fwd_kernel(const ThinArgs p) {
  constexpr int V = Chunk<T>::N, CK = 8 * V;
  __shared__ float wsm[kThinMaxT][8];
  __shared__ float red[2][4];
  const int Tn = p.kh * p.kw;
  for (int i = threadIdx.x; i < Tn * 8; i += 256) {
    const int t = i >> 3, c = i & 7;
    wsm[t][c] = (c < p.Cin && c < V * 1) ? Elem<T>::ld((const T*)p.wp + ((long)t * p.Mpad) * CK + c) : 0.f;
  }
  __syncthreads();
  const int ow = blockIdx.x * 256 + threadIdx.x, oh = blockIdx.y, b = blockIdx.z;
  const bool live = ow < p.Wo;
  const T* xb = (const T*)p.x + (long)b * p.H * p.W * p.ldx;
  float acc = 0.f;
  if (live) {
    for (int khi = 0; khi < p.kh; ++khi) {
      const int ih = oh - p.pad_t + khi * p.dil;
      if (ih < 0 || ih >= p.H) continue;               // uniform per workgroup row
      for (int kwi = 0; kwi < p.kw; ++kwi) {
        const int iw = ow - p.pad_l + kwi * p.dil;
        if (iw < 0 || iw >= p.W) continue;
        float f[V];
        Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(xb + ((long)ih * p.W + iw) * p.ldx), f);
        const float* wt = wsm[khi * p.kw + kwi];
#pragma unroll
        for (int c = 0; c < (V < 8 ? V : 8); ++c) acc = fmaf(f[c], wt[c], acc);
      }
    }
    if (p.bias) acc += p.bias[0];
    if (p.act == 1) acc = fmaxf(acc, 0.f);
    else if (p.act == 2) acc = 1.f / (1.f + __expf(-acc));
    T* dst = (T*)p.y + (((long)b * p.Ho + oh) * p.Wo + ow) * p.ldy;
    Elem<T>::st(dst, acc);
    acc = Elem<T>::rnd(acc);
  }
  if (p.stats) {   // uniform: BatchNorm statistics of the stored values (one channel)
    float s1 = live ? acc : 0.f, s2 = live ? acc * acc : 0.f;
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) {
      const float tot = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
      const int grp = b / p.bpg;
      atomicAdd(p.stats + (long)((blockIdx.x + blockIdx.y + blockIdx.z) % p.nrep) * p.rep_stride + ((long)grp * 2 + threadIdx.x) * p.stats_ld, (double)tot);
    }
  }
}

struct ThinWgArgs {
  const void* x; const void* dy; float* dwp; float* dbias;
  int B, H, W, Ho, Wo, kh, kw, dil, pad_t, pad_l;
  int Cin, ldx, lddy, Mpad;
};

// dW[t][0][c] += sum over pixels dY[p] * X[p + tap t][c]: every lane keeps all T x 8 partial sums in registers while it
// strides over its pixels; one wave-level and one LDS reduction, T x 8 atomics per workgroup.
template <typename T, int MAXT>
__global__ __launch_bounds__(256) void conv_thin_wgrad_kernel(const ThinWgArgs p) {
  constexpr int V = Chunk<T>::N, CK = 8 * V, NC = V < 8 ? V : 8;
  __shared__ float red[4][MAXT * 8 + 1];
  float acc[MAXT][NC];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[t][c] = 0.f;
  float bsum = 0.f;
  const long npix = (long)p.B * p.Ho * p.Wo;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
    const int ow = (int)(pix % p.Wo);
    const long r = pix / p.Wo;
    const int oh = (int)(r % p.Ho);
    const long b = r / p.Ho;
    const float g = Elem<T>::ld((const T*)p.dy + pix * p.lddy);
    bsum += g;
    const T* xb = (const T*)p.x + b * p.H * p.W * p.ldx;
    // taps in groups of 5: the group's loads are issued back to back from clamped (always valid) addresses and masked
    // afterwards — a branch per tap would make every load a separate round trip
    constexpr int GRP = 5;
#pragma unroll
    for (int t0 = 0; t0 < MAXT; t0 += GRP) {
      u32x4 raw[GRP];
      bool ok[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int t = t0 + j;
        ok[j] = false;
        if (t < MAXT && t < p.kh * p.kw) {    // uniform
          const int khi = t / p.kw, kwi = t - khi * p.kw;
          const int ih = oh - p.pad_t + khi * p.dil, iw = ow - p.pad_l + kwi * p.dil;
          ok[j] = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
          const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
          raw[j] = *reinterpret_cast<const u32x4*>(xb + ((long)ihc * p.W + iwc) * p.ldx);
        }
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        const int t = t0 + j;
        if (t < MAXT && t < p.kh * p.kw) {
          float f[V];
          Chunk<T>::unpack(raw[j], f);
          const float gm = ok[j] ? g : 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[t][c] = fmaf(gm, f[c], acc[t][c]);
        }
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float s = wave_sum(acc[t][c]);
      if (lane == 0) red[wave][t * 8 + c] = s;
    }
  bsum = wave_sum(bsum);
  if (lane == 0) red[wave][MAXT * 8] = bsum;
  __syncthreads();
  const int Tn = p.kh * p.kw;
  for (int i = threadIdx.x; i < Tn * 8; i += 256) {
    const int t = i >> 3, c = i & 7;
    if (c < NC && c < p.Cin) {
      const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      atomicAdd(p.dwp + ((long)t * p.Mpad) * CK + c, s);
    }
  }
  if (p.dbias && threadIdx.x == 0)
    atomicAdd(p.dbias, red[0][MAXT * 8] + red[1][MAXT * 8] + red[2][MAXT * 8] + red[3][MAXT * 8]);
}

}  // namespace
