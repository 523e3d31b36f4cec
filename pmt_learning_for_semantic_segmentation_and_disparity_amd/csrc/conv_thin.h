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
        for (int c = 0; c < (V < 8 ? V : 8); ++c) acc = fmaf(c < p.Cin ? f[c] : 0.f, wt[c], acc);   // pad channels of the pixel stride are not data (may be NaN)
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

// LDS-tiled bf16 variant of the above for full-resolution maps.  The register version keeps T x 8 sums per lane (200
// VGPRs for 5x5: two waves per SIMD) and walks its pixels one by one, every tap a separate L1 round trip: 129 us on
// 8 x 256 x 512 where the data is 19 MB (this kernel: 33 us).  Here a workgroup stages an 8 x 64 pixel tile of dY and its X halo in LDS and the
// LANES take the (tap, channel pair) outputs: lane = (half of the tile rows, tap, channel pair) runs along its rows with
// one 4-byte LDS read and two FMAs per pixel (the dY values arrive 8 at a time as one broadcast read).
template <int MAXT>
__global__ __launch_bounds__(256) void conv_thin_wgrad_tiled_kernel(const ThinWgArgs p, int tiles_h, int tiles_w, int IH, int IW, int P) {
  typedef bf16_t T;
  constexpr int TH = 8, TW = 64, CK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);                                   // [IH][P] pixels x 8 channels
  T* ys = reinterpret_cast<T*>(smem + (size_t)IH * P * 16);                     // [TH][TW]
  __shared__ float red[2][MAXT * 8];
  __shared__ float bsh;
  const int tid = threadIdx.x;
  const int T_ = p.kh * p.kw;
  const int half = tid >= 4 * MAXT ? 1 : 0;
  const int slot = tid - half * 4 * MAXT;
  const int t = slot >> 2, pair = slot & 3;
  const bool active = tid < 8 * MAXT && t < T_;
  const int ky = active ? t / p.kw : 0, kx = active ? t - ky * p.kw : 0;
  float acc0 = 0.f, acc1 = 0.f, bs = 0.f;
  if (tid == 0) bsh = 0.f;
  const int ntiles = p.B * tiles_h * tiles_w;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_h * tiles_w);
    const int tr = tile - b * tiles_h * tiles_w;
    const int ty = tr / tiles_w, tx = tr - ty * tiles_w;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 - p.pad_t, iw0 = ow0 - p.pad_l;
    __syncthreads();                                   // the previous tile has been consumed
    const T* xb = (const T*)p.x + (long)b * p.H * p.W * p.ldx;
    for (int i = tid; i < IH * P; i += 256) {
      const int r = i / P, c = i - r * P;
      const int gh = ih0 + r, gw = iw0 + c;
      const bool ok = c < IW && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
      xs[i] = ok ? *reinterpret_cast<const u32x4*>(xb + ((long)gh * p.W + gw) * p.ldx) : u32x4{0u, 0u, 0u, 0u};
    }
    const T* yb = (const T*)p.dy + (long)b * p.Ho * p.Wo * p.lddy;
    for (int i = tid; i < TH * TW; i += 256) {
      const int r = i / TW, c = i - r * TW;
      const int oh = oh0 + r, ow = ow0 + c;
      ys[i] = (oh < p.Ho && ow < p.Wo) ? yb[((long)oh * p.Wo + ow) * p.lddy] : (T)0;
    }
    __syncthreads();
    if (active) {
      for (int r = half * (TH / 2); r < (half + 1) * (TH / 2); ++r) {
        const unsigned* xrow = reinterpret_cast<const unsigned*>(xs + (r + ky * p.dil) * P + kx * p.dil) + pair;
        const u32x4* yrow = reinterpret_cast<const u32x4*>(ys + r * TW);
#pragma unroll 2
        for (int w0 = 0; w0 < TW / 8; ++w0) {
          float g[8];
          Chunk<T>::unpack(yrow[w0], g);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned xv = xrow[(w0 * 8 + j) * 4];
            acc0 = fmaf(g[j], bflo(xv), acc0);
            acc1 = fmaf(g[j], bfhi(xv), acc1);
          }
        }
      }
    } else if (tid >= 8 * MAXT && p.dbias) {           // the spare lanes add up the dY tile for the bias gradient
      for (int i = tid - 8 * MAXT; i < TH * TW; i += 256 - 8 * MAXT) bs += bf2f(ys[i]);
    }
  }
  if (active) { red[half][t * 8 + pair * 2] = acc0; red[half][t * 8 + pair * 2 + 1] = acc1; }
  __syncthreads();
  if (p.dbias && tid >= 8 * MAXT) atomicAdd(&bsh, bs);
  if (tid < T_ * 8) {
    const int tt = tid >> 3, c = tid & 7;
    if (c < p.Cin) atomicAdd(p.dwp + ((long)tt * p.Mpad) * CK + c, red[0][tid] + red[1][tid]);
  }
  __syncthreads();
  if (p.dbias && tid == 0) atomicAdd(p.dbias, bsh);
}


// ---- ONE input channel -> up to 64 output channels, <= 32 taps, stride 1, bf16 --------------------------------------
// The data gradient of a convolution with a single output map (the disparity head `ConvTranspose2dSame(64, 1, 5)` and the
// 1x1 attention gates `conv2dSame(64, 1, 1)`, models/dsnet_t2.py: dispoutConv / conv1d_at_*; the last 3x3x3 of PSMNet's
// classif1-3, models/stackhourglass.py:90-102): y[p][m] = sum_t x[p + off_t] * w[m][t].  On the halo-tile kernels the one
// real channel rides in a 32-channel k-step (1/32 of every MFMA) — 159 us for a layer whose only real work is writing
// 134 MB.  Here the TAPS are the reduction axis: one v_mfma_f32_16x16x32_bf16 per (16 pixels x 16 output channels) with
// k = tap index (kd*kh*kw <= 32); the pixel operand is gathered from a scalar halo tile in LDS.  Volumes: one workgroup
// walks `dpw` consecutive output slices of its 8x32 tile (weights and their fragments staged once, dpw + kd - 1 halo
// slices resident).
struct FanArgs {
  const void* x; const void* wp; void* y;
  int B, H, W, Ho, Wo, kh, kw, dil, pad_t, pad_l;
  int ldx, Cout, Mpad, ldy;
  int D, Do, kd, pad_d, dpw, zsegs;     // volume depth / output depth / depth taps / front padding / slices per workgroup / ceil(Do/dpw)
};

constexpr int kFanHalo = 4096;          // scalar halo elements in LDS

__global__ __launch_bounds__(256) void conv_fanout_kernel(const FanArgs p) {
  constexpr int TH = 8, TW = 32, CK = 64;
  __shared__ __attribute__((aligned(16))) bf16_t wl[64][32];      // [output channel][tap], zero beyond the kernel / Cout
  __shared__ bf16_t xt[kFanHalo + 8];                              // halo tiles of the single input map, slice after slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int T2 = p.kh * p.kw, Tn = T2 * p.kd;
  const int tiles_w = (p.Wo + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_w, oh0 = ty * TH, ow0 = (blockIdx.x - ty * tiles_w) * TW;
  const int b = blockIdx.y / p.zsegs, z0 = (blockIdx.y - b * p.zsegs) * p.dpw;
  const int nz = min(p.dpw, p.Do - z0);
  const int IH = TH + (p.kh - 1) * p.dil, IW = TW + (p.kw - 1) * p.dil, IS = IH * IW;
  const bf16_t* xb = (const bf16_t*)p.x + (long)b * p.D * p.H * p.W * p.ldx;
  for (int i = tid; i < 64 * 32; i += 256) {
    const int m = i >> 5, t = i & 31;
    wl[m][t] = (m < p.Cout && t < Tn) ? ((const bf16_t*)p.wp)[((long)t * p.Mpad + m) * CK] : (bf16_t)0;   // packed [kd][1][T2][Mpad][64], channel 0
  }
  const int nsl = nz + p.kd - 1;
  for (int i = tid; i < nsl * IS; i += 256) {
    const int sl = i / IS, r = i - sl * IS;
    const int ih = r / IW, iw = r - ih * IW;
    const int gz = z0 - p.pad_d + sl, gh = oh0 - p.pad_t + ih, gw = ow0 - p.pad_l + iw;
    xt[i] = (gz >= 0 && gz < p.D && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) ? xb[(((long)gz * p.H + gh) * p.W + gw) * p.ldx] : (bf16_t)0;
  }
  __syncthreads();
  // tap t of this lane's k-slice (8 taps: 8*lg .. 8*lg+7): offset inside the halo tiles, taps past the kernel read slot 0
  // (their weights are zero)
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = 8 * lg + j;
    const int kdi = t / T2, t2 = t - kdi * T2;
    const int khi = t2 / p.kw, kwi = t2 - khi * p.kw;
    toff[j] = t < Tn ? kdi * IS + khi * p.dil * IW + kwi * p.dil : 0;
  }
  u32x4 af[4];                                                     // weights: lane (l15 = output channel of the tile, lg = k-slice)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(&wl[mi * 16 + l15][8 * lg]);
  for (int z = 0; z < nz; ++z) {
    bf16_t* yb = (bf16_t*)p.y + ((long)b * p.Do + z0 + z) * p.Ho * p.Wo * p.ldy;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {                               // wave: tile rows 2*wave, 2*wave + 1; two 16-pixel column blocks each
      const int r = 2 * wave + (ni >> 1), c = (ni & 1) * 16 + l15;
      const int base = z * IS + r * IW + c;
      unsigned short v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = xt[base + toff[j]];
      const u32x4 bfrag = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
      const int oh = oh0 + r, ow = ow0 + c;
      const bool valid = oh < p.Ho && ow < p.Wo;
      bf16_t* dst = yb + ((long)oh * p.Wo + ow) * p.ldy;
      f32x4 acc[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mi * 16 < p.Mpad) Mma<bf16_t>::run(acc[mi], af[mi], bfrag);   // uniform
      }
      // the (up to) four 32-byte pieces of a pixel's line in consecutive stores: they merge in L2 (see conv_band.h)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (mi * 16 < p.Mpad) {
          const int co = mi * 16 + 4 * lg;
          if (valid && co + 3 < p.Cout) {
            *reinterpret_cast<u32x2*>(dst + co) = u32x2{pack2bf(acc[mi][0], acc[mi][1]), pack2bf(acc[mi][2], acc[mi][3])};
          } else if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < p.Cout) dst[co + e] = f2bf(acc[mi][e]);
          }
        }
      }
    }
  }
}

inline bool fanout_ok(int Cin, int Cout, int T, int stride, int kd, int sd, int ldy, const void* y) {
  return Cin == 1 && Cout <= 64 && Cout >= 8 && T * kd <= 32 && stride == 1 && sd == 1 && ldy % 4 == 0 && ((uintptr_t)y & 7) == 0;
}

// slices per workgroup: as many as the halo buffer holds, at most 8, and no fewer workgroups than ~4 per CU
inline int fanout_dpw(const FanArgs& a) {
  const int IS = (8 + (a.kh - 1) * a.dil) * (32 + (a.kw - 1) * a.dil);
  int dpw = kFanHalo / IS - (a.kd - 1);
  if (dpw > 8) dpw = 8;
  if (dpw > a.Do) dpw = a.Do;
  const long tiles = (long)sdhip_cdiv(a.Ho, 8) * sdhip_cdiv(a.Wo, 32) * a.B;
  while (dpw > 1 && tiles * sdhip_cdiv(a.Do, dpw) < 1024) --dpw;
  return dpw;
}

inline int launch_fanout(FanArgs a, hipStream_t s) {
  a.dpw = fanout_dpw(a);
  if (a.dpw < 1) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv fan-out: halo of a %dx%dx%d kernel with dilation %d exceeds the tile buffer", a.kd, a.kh, a.kw, a.dil);
  a.zsegs = sdhip_cdiv(a.Do, a.dpw);
  dim3 grid(sdhip_cdiv(a.Ho, 8) * sdhip_cdiv(a.Wo, 32), a.B * a.zsegs);
  hipLaunchKernelGGL(conv_fanout_kernel, grid, dim3(256), 0, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// ---- 16..64 input channels -> ONE output map, <= 32 taps, stride 1, bf16 (the mirror image of conv_fanout_kernel) --------
// The last 3x3x3 of PSMNet's classif1-3 (models_psmnet/stackhourglass.py:90-102) and the single-map heads of the 2-D
// networks (models/dsnet_t2.py: dispoutConv): on the halo-tile kernels the one real output channel owns a 16-row MFMA tile
// (360 us for reading 377 MB once).  Here the TAPS are the MFMA rows: for every pixel q of the halo of an 8x32 output tile
//     P[t][q] = sum_ci w[t][ci] * x[q][ci]              (A = weights in registers, B = 16-byte pieces of x straight from
// global memory, one v_mfma_f32_16x16x32_bf16 per 16 pixels, 16 taps and 32 channels), P goes to LDS tap-major, and
// output pixel o is the sum over taps of P[t][o + off_t] — a fixed order, so the result does not depend on timing.
// Volumes (KD = 3): the workgroup walks input slices; slice s feeds depth tap k of output slice s - k + pad_d, three
// running sums per thread; the next slice's pieces travel through registers behind the sums.
struct FaninArgs {
  const void* x; const void* wp; void* y; const float* bias;
  int B, H, W, Ho, Wo, kh, kw, pad_t, pad_l;
  int D, Do, pad_d;
  int Cin, ldx, ldy, Mpad, act;
  int dpw, zsegs, npxp, pitch;           // output slices per workgroup / ceil(Do / dpw) / halo pixels rounded up to 16 / floats per tap row of P
};

constexpr int kFaninMaxNT = 7;           // 16-pixel halo blocks per wave: halo <= 448 pixels

template <int KD, int NK, int KHW = 0>            // KHW = 3: kh = kw = 3 known at compile time (the tap sum unrolls to constant LDS offsets)
__global__ __launch_bounds__(256) void conv_fanin_kernel(const FaninArgs p) {
  constexpr int TH = 8, TW = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char fan_smem[];
  float* const P = reinterpret_cast<float*>(fan_smem);              // [tap][pitch]: pitch % 16 == 4, the four tap rows a wave writes at once fall on different banks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int T2 = p.kh * p.kw, Tn = T2 * KD;
  const int tiles_w = (p.Wo + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_w, oh0 = ty * TH, ow0 = (blockIdx.x - ty * tiles_w) * TW;
  const int b = blockIdx.y / p.zsegs, z0 = (blockIdx.y - b * p.zsegs) * p.dpw;
  const int nz = min(p.dpw, p.Do - z0);
  const int IW = TW + p.kw - 1, NPX = (TH + p.kh - 1) * IW;
  const int nblk = p.npxp >> 4;
  const bf16_t* xb = (const bf16_t*)p.x + (long)b * p.D * p.H * p.W * p.ldx;

  // weights: lane (l15 = tap of the 16-row tile, lg = 8-channel slice of the 32-channel k-step)
  u32x4 af[2][NK];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int t = mt * 16 + l15, ch = ks * 32 + 8 * lg;
      af[mt][ks] = (t < Tn && ch < p.Cin) ? *reinterpret_cast<const u32x4*>((const bf16_t*)p.wp + ((long)t * p.Mpad) * 64 + ch) : u32x4{0u, 0u, 0u, 0u};
    }
  // this lane's halo pixel of each of its wave's blocks (block = wave + 4 i): element offset inside a slice, -1 = padding
  int xo[kFaninMaxNT];
#pragma unroll
  for (int i = 0; i < kFaninMaxNT; ++i) {
    const int q = (wave + 4 * i) * 16 + l15;
    const int ih = q / IW, iw = q - ih * IW;
    const int gh = oh0 - p.pad_t + ih, gw = ow0 - p.pad_l + iw;
    xo[i] = (q < NPX && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) ? (gh * p.W + gw) * p.ldx + 8 * lg : -1;
  }
  u32x4 xv[kFaninMaxNT][NK];
  auto fetch = [&](int s) {                                         // s: input slice (any integer; outside the volume = zeros)
    const bool in = s >= 0 && s < p.D;                              // uniform
    const bf16_t* xs = xb + (long)(in ? s : 0) * p.H * p.W * p.ldx;
#pragma unroll
    for (int i = 0; i < kFaninMaxNT; ++i)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
        xv[i][ks] = (in && wave + 4 * i < nblk && xo[i] >= 0 && ks * 32 + 8 * lg < p.Cin) ? *reinterpret_cast<const u32x4*>(xs + xo[i] + ks * 32)
                                                                                          : u32x4{0u, 0u, 0u, 0u};
  };
  const int r = tid >> 5, c = tid & 31;                             // this thread's output pixel of the tile
  const bool live = oh0 + r < p.Ho && ow0 + c < p.Wo;
  const float bias = p.bias ? p.bias[0] : 0.f;
  auto store = [&](int z, float v) {
    v += bias;
    if (p.act == 1) v = fmaxf(v, 0.f);
    else if (p.act == 2) v = 1.f / (1.f + __expf(-v));
    if (live) ((bf16_t*)p.y)[((((long)b * p.Do + z) * p.Ho + oh0 + r) * p.Wo + ow0 + c) * p.ldy] = f2bf(v);
  };

  float run1 = 0.f, run2 = 0.f;                                     // sums of output slices s + pad_d - 1, s + pad_d - 2 so far
  const int s_begin = z0 - p.pad_d, ns = nz + KD - 1;
  fetch(s_begin);
  for (int si = 0; si < ns; ++si) {
    const int s = s_begin + si;
    __syncthreads();                                                // the sums of the previous slice have been taken
#pragma unroll
    for (int i = 0; i < kFaninMaxNT; ++i) {
      if (wave + 4 * i < nblk) {                                    // uniform
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          Mma<bf16_t>::run(acc[0], af[0][ks], xv[i][ks]);
          Mma<bf16_t>::run(acc[1], af[1][ks], xv[i][ks]);
        }
        const int q = (wave + 4 * i) * 16 + l15;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = mt * 16 + 4 * lg + e;
            if (t < Tn) P[t * p.pitch + q] = acc[mt][e];
          }
      }
    }
    __syncthreads();
    if (si + 1 < ns) fetch(s + 1);                                  // in flight behind the sums
    float S[KD];
#pragma unroll
    for (int k = 0; k < KD; ++k) {
      float a = 0.f;
      const float* Pk = P + (k * T2) * p.pitch + r * IW + c;
      if constexpr (KHW == 3) {
#pragma unroll
        for (int t = 0; t < 9; ++t) a += Pk[t * p.pitch + (t / 3) * (TW + 2) + t % 3];
      } else {
        for (int khi = 0; khi < p.kh; ++khi)
          for (int kwi = 0; kwi < p.kw; ++kwi) a += Pk[(khi * p.kw + kwi) * p.pitch + khi * IW + kwi];
      }
      S[k] = a;
    }
    if constexpr (KD == 1) {
      store(s + p.pad_d, S[0]);
    } else {
      const int zf = s + p.pad_d - 2;                               // the slice whose last depth tap this was
      if (zf >= z0 && zf < z0 + nz) store(zf, run2 + S[2]);
      run2 = run1 + S[1];
      run1 = S[0];
    }
  }
}

inline bool fanin_ok(int Cin, int Cout, int kh, int kw, int stride, int dil, int kd, int sd, int ldx, const void* x) {
  if (!(Cout == 1 && Cin % 8 == 0 && Cin >= 16 && stride == 1 && dil == 1 && sd == 1 && kh * kw * kd <= 32 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0)) return false;
  if (kd == 3) { if (Cin > 32) return false; } else if (kd != 1 || Cin > 64) return false;
  return (8 + kh - 1) * (32 + kw - 1) <= kFaninMaxNT * 64;
}

inline int launch_fanin(FaninArgs a, int kd, hipStream_t s) {
  const int npx = (8 + a.kh - 1) * (32 + a.kw - 1);
  a.npxp = (npx + 15) & ~15;
  a.pitch = a.npxp + 4;
  const size_t lds = (size_t)a.kh * a.kw * kd * a.pitch * sizeof(float);
  if (lds > 64 * 1024) return 1;                                    // (the caller takes another kernel)
  int dpw = kd == 1 ? 1 : 12;
  if (dpw > a.Do) dpw = a.Do;
  const long tiles = (long)sdhip_cdiv(a.Ho, 8) * sdhip_cdiv(a.Wo, 32) * a.B;
  while (dpw > 4 && tiles * sdhip_cdiv(a.Do, dpw) < 1024) --dpw;
  a.dpw = dpw; a.zsegs = sdhip_cdiv(a.Do, dpw);
  if ((long)a.B * a.zsegs > 65535) return 1;
  dim3 grid(sdhip_cdiv(a.Ho, 8) * sdhip_cdiv(a.Wo, 32), a.B * a.zsegs);
  if (kd == 3 && a.kh == 3 && a.kw == 3) hipLaunchKernelGGL((conv_fanin_kernel<3, 1, 3>), grid, dim3(256), lds, s, a);
  else if (kd == 3) hipLaunchKernelGGL((conv_fanin_kernel<3, 1>), grid, dim3(256), lds, s, a);
  else if (a.Cin <= 32) hipLaunchKernelGGL((conv_fanin_kernel<1, 1>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv_fanin_kernel<1, 2>), grid, dim3(256), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

}  // namespace
