// Weight gradient of a convolution with ONE output map and 9..32 input channels, <= 32 taps, stride 1, bf16: the 3x3x3
// `Conv3d(32, 1)` that ends classif1-3 of PSMNet (models_psmnet/stackhourglass.py:90-102).
//
//   dW[t][ci] = sum_p x[p][ci] * dy[p + pad - off_t]
//
// On the 64-byte-row kernel (conv_wgrad_half.h) the one real output channel rides in a 16-row MFMA tile and every tap is a
// product of its own: 108 MFMAs per 32 pixels, 620 us per layer at 4 x 48 x 128 x 240 for a layer whose only real work is
// reading x once (377 MB).  Here the TAPS are the M axis (as in conv_fanout_kernel for the data gradient): A[t][p] is
// gathered from a scalar halo of dy in LDS, B[p][ci] = x, read with the transposing ds_read_b64_tr_b16 (the contraction runs
// over pixels) — 4 MFMAs per 32 pixels.  One workgroup (4 waves) walks `dpw` consecutive input slices of its 8x32-pixel
// tile (the next slice travels through registers behind the MFMAs), reduces its four waves' partial sums in LDS and flushes
// once with f32 atomics.
//
// LDS image of the x tile: pixel q = row * 32 + col at byte q * 64, its two 32-byte segments (channels 0-15 / 16-31)
// exchanged when bit 3 of the column is set — the layout conv_wgrad_half.h reads conflict-free.
#pragma once
#include "conv_wgrad_fast.h"

namespace {

struct SingleWgArgs {
  const void* x; const void* dy; float* dwp;
  int B, H, W, Ho, Wo, kh, kw, pad_t, pad_l;
  int D, Do, kd, pad_d;
  int Cin, ldx, lddy, Mpad;
  int dpw, zsegs;                              // input slices per workgroup / ceil(D / dpw)
};

constexpr int kSingleHalo = 4096;              // scalar dy halo elements in LDS

__global__ __launch_bounds__(256) void conv_single_wgrad_kernel(const SingleWgArgs p) {
  constexpr int TH = 8, TW = 32;
  __shared__ __attribute__((aligned(16))) unsigned char xs[TH * TW * 64];   // x tile; afterwards the cross-wave reduction
  __shared__ bf16_t dh[kSingleHalo + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int p4 = lane & 3, r4 = l15 >> 2;
  const int T2 = p.kh * p.kw, Tn = T2 * p.kd;
  const int tiles_w = (p.W + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_w, r0 = ty * TH, c0 = (blockIdx.x - ty * tiles_w) * TW;
  const int b = blockIdx.y / p.zsegs, z0 = (blockIdx.y - b * p.zsegs) * p.dpw;
  const int nz = min(p.dpw, p.D - z0);
  const int IH = TH + p.kh - 1, IW = TW + p.kw - 1, IS = IH * IW;
  const bf16_t* xb = (const bf16_t*)p.x + (long)b * p.D * p.H * p.W * p.ldx;
  const bf16_t* dyb = (const bf16_t*)p.dy + (long)b * p.Do * p.Ho * p.Wo * p.lddy;

  // ---- dy halo of all the slices this workgroup touches: slice s <-> dy slice z0 + pad_d - (kd-1) + s, likewise rows / columns ----
  const int nsl = nz + p.kd - 1;
  for (int i = tid; i < nsl * IS; i += 256) {
    const int sl = i / IS, r = i - sl * IS;
    const int ih = r / IW, iw = r - ih * IW;
    const int gz = z0 + p.pad_d - (p.kd - 1) + sl, gh = r0 + p.pad_t - (p.kh - 1) + ih, gw = c0 + p.pad_l - (p.kw - 1) + iw;
    dh[i] = (gz >= 0 && gz < p.Do && gh >= 0 && gh < p.Ho && gw >= 0 && gw < p.Wo) ? dyb[(((long)gz * p.Ho + gh) * p.Wo + gw) * p.lddy] : (bf16_t)0;
  }
  // tap rows of this lane's two A tiles (m = l15): offset of x pixel (slice 0, row 0, column 8 * lg) inside the halo; taps past the
  // kernel read slot 0 (their rows are never flushed)
  int aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int t = mt * 16 + l15;
    const int kdi = t / T2, t2 = t - kdi * T2;
    const int khi = t2 / p.kw, kwi = t2 - khi * p.kw;
    aoff[mt] = t < Tn ? (p.kd - 1 - kdi) * IS + (p.kh - 1 - khi) * IW + (p.kw - 1 - kwi) + 8 * lg : 0;
  }
  // x fragments: lane (p4, r4, lg) addresses pixels 8 * lg + r4 and + 4 of a tile row, 8-byte column p4 of channel tile nt
  int boff[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ca = 8 * lg + r4, cb = ca + 4;
    boff[nt][0] = ca * 64 + ((nt ^ ((ca >> 3) & 1)) << 5) + p4 * 8;
    boff[nt][1] = cb * 64 + ((nt ^ ((cb >> 3) & 1)) << 5) + p4 * 8;
  }
  // staging plan of a slice: 1024 chunks of 16 bytes, chunk id = tid + 256 i -> pixel id >> 2, slot id & 3 (lane-linear in LDS,
  // the swizzle applied to the SOURCE channel)
  const int cin8 = (p.Cin + 7) & ~7;
  long soff[4];                                 // element offset inside a slice, -1: zeros
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, q = id >> 2, s4 = id & 3;
    const int row = q >> 5, col = q & 31;
    const int ch = (((s4 >> 1) ^ ((col >> 3) & 1)) << 4) + (s4 & 1) * 8;
    const int gh = r0 + row, gw = c0 + col;
    soff[i] = (gh < p.H && gw < p.W && ch < cin8) ? ((long)gh * p.W + gw) * p.ldx + ch : -1;
  }
  auto fetch = [&](int z, u32x4 (&v)[4]) {
    const bf16_t* xz = xb + (long)(z0 + z) * p.H * p.W * p.ldx;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = soff[i] >= 0 ? *reinterpret_cast<const u32x4*>(xz + soff[i]) : u32x4{0u, 0u, 0u, 0u};
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 v[4];
  fetch(0, v);
  for (int z = 0; z < nz; ++z) {
    __syncthreads();                            // everybody is done with the previous slice's tile (first trip: the dy halo is complete)
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(xs + (tid + 256 * i) * 16) = v[i];
    __syncthreads();
    if (z + 1 < nz) fetch(z + 1, v);            // in flight behind the MFMAs
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {            // wave: tile rows 2 * wave, 2 * wave + 1, one 32-pixel k-step each
      const int row = 2 * wave + rr;
      u32x4 af[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16_t* src = dh + z * IS + row * IW + aoff[mt];
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = src[j];
        af[mt] = u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                       (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
      }
      const unsigned char* xr = xs + row * (TW * 64);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const u32x4 bf = tr_pair(xr + boff[nt][0], xr + boff[nt][1]);
        Mma<bf16_t>::run(acc[0][nt], af[0], bf);
        Mma<bf16_t>::run(acc[1][nt], af[1], bf);
      }
    }
  }
  // ---- four waves' partial sums -> one: [wave][tap 0..31][ci 0..31] f32 in the tile buffer (16 KB), then one atomic per value ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(xs);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 1024 + (mt * 16 + 4 * lg + r) * 32 + nt * 16 + l15] = acc[mt][nt][r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, t = id >> 5, ci = id & 31;
    if (t < Tn && ci < p.Cin) atomicAdd(p.dwp + ((long)t * p.Mpad) * 64 + ci, red[id] + red[1024 + id] + red[2048 + id] + red[3072 + id]);
  }
}

inline bool single_wgrad_ok(int Cin, int Cout, int T, int stride, int dil, int kd, int sd, int ldx, const void* x) {
  return Cout == 1 && Cin > 8 && Cin <= 32 && T * kd <= 32 && stride == 1 && dil == 1 && sd == 1 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0;
}

inline int launch_single_wgrad(SingleWgArgs a, hipStream_t s) {
  const int IS = (8 + a.kh - 1) * (32 + a.kw - 1);
  int dpw = kSingleHalo / IS - (a.kd - 1);
  if (dpw > 8) dpw = 8;
  if (dpw > a.D) dpw = a.D;
  const long tiles = (long)sdhip_cdiv(a.H, 8) * sdhip_cdiv(a.W, 32) * a.B;
  while (dpw > 1 && tiles * sdhip_cdiv(a.D, dpw) < 1024) --dpw;
  if (dpw < 1) return 1;                        // halo does not fit: the caller takes another kernel
  a.dpw = dpw; a.zsegs = sdhip_cdiv(a.D, dpw);
  if ((long)a.B * a.zsegs > 65535) return 1;
  dim3 grid(sdhip_cdiv(a.H, 8) * sdhip_cdiv(a.W, 32), a.B * a.zsegs);
  hipLaunchKernelGGL(conv_single_wgrad_kernel, grid, dim3(256), 0, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

}  // namespace
