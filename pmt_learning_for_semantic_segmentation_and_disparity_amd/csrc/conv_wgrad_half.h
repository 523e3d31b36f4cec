// Weight gradient of 3x3 (x3) stride-1 convolutions with <= 32 input AND <= 32 output channels (bf16): the 3-D stack of
// PSMNet (models_psmnet/stackhourglass.py:31-50,59-84: 17 layers of 32 -> 32 over (B,32,48,H/4,W/4) volumes) and the
// 32-channel full-resolution blocks of the 2-D networks (models/dsnet_t2.py:80-117).
//
// wgrad_fast_kernel stages 128-byte LDS rows (64 channels per pixel); with 32 channels half of every row — half of the
// LDS-DMA instructions, of the LDS capacity and of the L2 traffic of the sweep — is zero padding, and each of the three
// depth taps of a 3-D layer sweeps all tiles on its own (X and dY staged three times).  That kernel is bound by exactly
// this staging: per 4 x 32-pixel tile and depth tap it moves 56 KB into LDS for 144 MFMAs (measured 340 us per PSMNet
// layer = 0.5 PFLOP/s, ~12 TB/s of staged bytes).  Here
//   * an LDS row is 64 bytes = one pixel's 32 channels (two pixels per 128-byte line), halo pitch 36 instead of 48;
//   * a workgroup takes ALL depth taps of its output tile: the dY tile is staged (and its fragments read) once for
//     27 taps; three input slices of 14 KB each;
// so a tile-step stages 50 KB for 432 MFMAs — one sixth of the bytes per MFMA.
//
// LDS images ([pixel][32 channels], transposing fragment reads as in wgrad_fast_kernel: the contraction runs over pixels):
//   pixel q of an image lives at byte q * 64; its two 32-byte segments (channels 0-15 / 16-31) are exchanged when bit 3 of
//   the pixel's COLUMN is set.  A 32-lane half of ds_read_b64_tr_b16 touches pixels c..c+3 and c+8..c+11 of one image row,
//   32 bytes each: the first four fall on the four 16-bank groups, the second four on the same groups but — column bit 3
//   differs — on the other 8-bank half of each: conflict-free for every tap shift.  The key depends on the column only, so
//   a kernel-row / k-step / depth-slice shift is a compile-time offset of the read.
// Filled by LDS-DMA (lane-linear destination, swizzle on the source address), double-buffered, one barrier per tile.
#pragma once
#include "conv_wgrad_fast.h"

namespace {

template <int KD>
__device__ __forceinline__ void wgrad32_body(const WgfArgs& p, const int bx, const int gdx) {
  typedef bf16_t T;
  constexpr int TH = 4, TW = 32, IH = TH + 2, IWP = 36;
  constexpr int XPIX = IH * IWP;               // 216 halo pixels per input slice
  constexpr int XBLK = (XPIX + 15) / 16;       // 14 DMA blocks of 1 KB (16 pixels x 64 bytes)
  constexpr int XS = XBLK * 1024;              // bytes of one slice image
  constexpr int YBLK = TH * TW / 16;           // 8
  constexpr int NBLK = KD * XBLK + YBLK;       // 50 / 22 blocks per stage
  constexpr int SB = NBLK * 1024;              // stage bytes
  constexpr int NT = 9 * KD;                   // taps of the workgroup
  constexpr int MAXTW = (NT + 3) / 4;          // per wave quartet: 7 / 3
  constexpr int NI = (NBLK + 7) / 8;           // DMA blocks per wave and stage: 7 / 3
  constexpr int NKS = TH;                      // one k-step (32 pixels) per tile row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: it selects LDS-DMA destinations, which travel in M0)
  const int l15 = lane & 15, lg = lane >> 4;
  const int ci = wave & 1, tg = wave >> 1;     // input-channel tile, tap quartet
  const int p4 = lane & 3, r4 = l15 >> 2, pk = 8 * lg + r4;

  f32x4 acc[MAXTW][2];
#pragma unroll
  for (int j = 0; j < MAXTW; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses inside a stage (k-step / depth slice / kernel row are constant offsets) ----
  int a_lo[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) a_lo[mi] = KD * XS + pk * 64 + ((mi ^ (lg & 1)) << 5) + p4 * 8;   // + 256: pixel pk + 4 (same column bit 3)
  int b_lo[MAXTW], b_hi[MAXTW];
#pragma unroll
  for (int j = 0; j < MAXTW; ++j) {
    int tt = tg + 4 * j;
    if (tt >= NT) tt = 0;                      // idle slot: reads tap 0, its accumulator is never flushed
    const int kdi = tt / 9, t9 = tt - kdi * 9, khi = t9 / 3, kwi = t9 - khi * 3;
    const int c0 = pk + kwi, c1 = c0 + 4;
    b_lo[j] = kdi * XS + (khi * IWP + c0) * 64 + ((ci ^ ((c0 >> 3) & 1)) << 5) + p4 * 8;
    b_hi[j] = kdi * XS + (khi * IWP + c1) * 64 + ((ci ^ ((c1 >> 3) & 1)) << 5) + p4 * 8;
  }

  // ---- staging plan: block id = wave + 8 i; ids < KD*XBLK are halo blocks (slice id / XBLK), the rest dY blocks ----
  const int s4 = lane & 3, pl16 = lane >> 2;   // 16-byte slot inside the pixel's 64 bytes, pixel inside the block
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  int hr_[NI], hc_[NI], ch_[NI];               // halo row / column (or tile row / column) and first channel of the lane's chunk
  bool isx_[NI], live_[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = wave + 8 * i;
    isx_[i] = id < KD * XBLK;
    live_[i] = id < NBLK;
    const int blk = isx_[i] ? id % XBLK : id - KD * XBLK;
    const int q = blk * 16 + pl16;
    const int pitch = isx_[i] ? IWP : TW;
    hr_[i] = q / pitch; hc_[i] = q - hr_[i] * pitch;
    const int seg = (s4 >> 1) ^ ((hc_[i] >> 3) & 1);
    ch_[i] = seg * 16 + (s4 & 1) * 8;
    if (isx_[i] && q >= XPIX) live_[i] = false;                      // tail of the last halo block: never read
  }
  const int cin8 = (p.Cin + 7) & ~7, cout8 = (p.Cout + 7) & ~7;

  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int ntiles = p.B * p.Do * tiles_h * tiles_w;
  const int ntl = (ntiles - bx + gdx - 1) / gdx;

  auto issue = [&](int tile, unsigned char* buf) {
    const int img = tile / (tiles_h * tiles_w);
    const int tr = tile - img * tiles_h * tiles_w;
    const int ty = tr / tiles_w, tx = tr - ty * tiles_w;
    const int b = img / p.Do, dz = img - b * p.Do;
    const int oh0 = ty * TH, ow0 = tx * TW;
    const int ih0 = oh0 - p.pad_t, iw0 = ow0 - p.pad_l;
    const unsigned dst0 = lds0 + (unsigned)(buf - smem);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int id = wave + 8 * i;
      if (id >= NBLK) break;                                          // wave-uniform
      const T* src = (const T*)sdhip_zero16;
      if (isx_[i]) {
        const int din = dz + id / XBLK - p.pad_d;                     // input depth slice of this block's depth tap
        const int gh = ih0 + hr_[i], gw = iw0 + hc_[i];
        if (live_[i] && din >= 0 && din < p.D && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W && ch_[i] < cin8)
          src = (const T*)p.x + (((long)b * p.D + din) * p.H + gh) * (long)p.W * p.ldx + (long)gw * p.ldx + ch_[i];
      } else {
        const int oh = oh0 + hr_[i], ow = ow0 + hc_[i];
        if (oh < p.Ho && ow < p.Wo && ch_[i] < cout8)
          src = (const T*)p.dy + ((long)img * p.Ho + oh) * (long)p.Wo * p.lddy + (long)ow * p.lddy + ch_[i];
      }
      glds16(src, __builtin_amdgcn_readfirstlane(dst0 + id * 1024));
    }
  };

  auto compute = [&](const unsigned char* buf) {
    u32x4 af[NKS][2];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        af[ks][mi] = tr_pair(buf + a_lo[mi] + ks * (TW * 64), buf + a_lo[mi] + 256 + ks * (TW * 64));
    constexpr int NST = NKS * MAXTW, PD = 4;
    u32x4 bfr[PD];
    auto ldb = [&](int u) -> u32x4 {
      const int ks = u / MAXTW, j = u % MAXTW;
      return tr_pair(buf + b_lo[j] + ks * (IWP * 64), buf + b_hi[j] + ks * (IWP * 64));
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) bfr[u] = ldb(u);
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int ks = u / MAXTW, j = u % MAXTW;
      const u32x4 bf = bfr[u % PD];
      Mma<T>::run(acc[j][0], af[ks][0], bf);
      Mma<T>::run(acc[j][1], af[ks][1], bf);
      if (u + PD < NST) bfr[u % PD] = ldb(u + PD);
      __builtin_amdgcn_sched_barrier(0);       // keep the lookahead (see wgrad_fast_kernel)
    }
  };

  if (ntl > 0) issue(bx, smem);
  for (int it = 0; it < ntl; ++it) {
    unsigned char* buf = smem + (it & 1) * SB;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's DMA has landed
    __syncthreads();                                    // visible to all; everybody is done with the other buffer
    if (it + 1 < ntl) issue(bx + (it + 1) * gdx, smem + ((it + 1) & 1) * SB);
    compute(buf);
  }

  // ---- flush: f32 atomics into the packed gradient buffer [kd][nq = 1][9][Mpad][64] ----
  if (ci * 16 < p.Cin) {
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
      const int tt = tg + 4 * j;
      if (tt < NT) {
        const int kdi = tt / 9, t9 = tt - kdi * 9;
        float* dst = p.dwp + ((long)(kdi * 9 + t9) * p.Mpad) * 64;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          if (mi * 16 < p.Cout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int m = mi * 16 + 4 * lg + r;
              if (m < p.Mpad) atomicAdd(dst + (long)m * 64 + ci * 16 + l15, acc[j][mi][r]);
            }
          }
        }
      }
    }
  }
}

template <int KD>
__global__ __launch_bounds__(512) void wgrad32_kernel(const WgfArgs p) {
  wgrad32_body<KD>(p, (int)blockIdx.x, (int)gridDim.x);
}

template <int KD>
__global__ __launch_bounds__(512) void wgrad32_group_kernel(const WgfGroup g) {
  const int b = (int)blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.wg0[i + 1]) ++i;
  wgrad32_body<KD>(g.L[i], b - g.wg0[i], g.gx[i]);
}

// Plan (see plan_wgf): gy = 1 — a workgroup holds every tap of its tile.
template <int KD>
int plan_wg32(const WgfArgs& a, WgfPlan& pl) {
  auto kern = wgrad32_kernel<KD>;
  auto gkern = wgrad32_group_kernel<KD>;
  constexpr size_t lds = 2 * (size_t)(KD * 14 + 8) * 1024;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gkern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  const int ntiles = a.B * a.Do * sdhip_cdiv(a.Ho, 4) * sdhip_cdiv(a.Wo, 32);
  const int occ = lds * 2 <= 160 * 1024 ? 2 : 1;
  int gx = 256 * occ;
  if (gx > sdhip_cdiv(ntiles, 2)) gx = sdhip_cdiv(ntiles, 2);
  if (gx < 1) gx = 1;
  pl.single = (const void*)kern; pl.group = (const void*)gkern;
  pl.lds = lds; pl.ntiles = ntiles; pl.gy = 1; pl.gx = gx; pl.occ = occ;
  pl.t_tile_us = KD == 3 ? 1.0 : 0.4;
  pl.flush_us = (double)(9 * KD) * 32 * 64 * 4 / (sdhip_diag().tune_atomic_tbs * 1e6);
  pl.a = a;
  return 0;
}

}  // namespace
