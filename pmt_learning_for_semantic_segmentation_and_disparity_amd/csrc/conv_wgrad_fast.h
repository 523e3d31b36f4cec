// Weight gradient of the direct convolution, bf16 fast path for gfx950 (same contraction as conv_wgrad.hip's general
// kernel: dW[t][m][k] = sum over pixels of dY[pix][m] * Xeff[pix shifted by tap t][k]).
//
// What differs from the general kernel:
//  * X halo tile and dY tile are DOUBLE-BUFFERED in LDS and filled by LDS-DMA (global_load_lds): the loads of tile i+1
//    fly while tile i is multiplied, one barrier per tile.  When a BatchNorm prologue has to be applied on the way, both
//    operands go through registers instead, prefetched TWO tiles ahead in two register sets (see the sweep).
//  * LDS images use the linear-destination / swizzled-source layout of conv_fast.h (chunk slot = c ^ (row & 6), row
//    pitch of the halo tile a multiple of 8), so the byte address of a transposing fragment read is
//        per-lane base (one per tap, computed ONCE per kernel)  +  k-step offset (wave-uniform)
//    instead of ~12 VALU operations of swizzle arithmetic per fragment; the dY fragments of a whole tile are read once
//    and kept in registers while all taps walk over the tile.
//  * no per-tap branches inside the MFMA loop: tap slots past the kernel read tap 0 and feed accumulators that are
//    never flushed.
#pragma once
#include "conv_fast.h"
#include <math.h>

namespace {

struct WgfArgs {
  const void* x; const void* dy; float* dwp; float* dbias;
  const float* in_scale; const float* in_shift;
  int B, H, W, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l;
  int D, Do, kd, sd, pad_d;
  int Cin, ldx, Cout, Mpad, lddy;
  int in_relu, bpg;
  int tpb, ntg, nq;   // taps per workgroup, tap groups, channel chunks
  // Chunk packing for 1x1 convolutions with many input channels (the DenseNet bottlenecks: dW = dY^T X is a GEMM with few
  // pixels and 96..1024 input channels): qb = 2 or 4 consecutive 64-channel chunks of a pixel are laid side by side in the
  // LDS image as if they were qb pixels of a row and walked as the qb "taps" of a 1 x qb kernel with horizontal stride qb.
  // Row of (pixel p, chunk j) inside a halo line: (p >> 1) * 2qb + 2j + (p & 1) — pixel PAIRS stay on adjacent rows, so the
  // rows a transposing read touches (p..p+3, p+8..p+11) keep the bank pattern of the unpacked image (rows 4 apart would
  // all fall into the same 128-byte half of the banks).  One workgroup then multiplies each dY tile with qb chunks instead of one: dY is re-read
  // nq/qb instead of nq times and the per-tile overhead (barrier, DMA wait) is spread over qb times the MFMAs.
  // Host sets kh = 1, kw = qb, tpb = qb, ntg = 1, nq = ceil(nq_tot / qb).  qb = 0: off.
  int qb, qsh, nq_tot;
};

typedef __attribute__((address_space(3))) bf16x4_t* lds_bf4_p;

// LDS image of the weight-gradient kernel: 128-byte rows, chunk slot = c ^ key(row) with key = (row & 2) | ((row >> 1) & 4).
// A 32-lane group of ds_read_b64_tr_b16 touches rows r+{0,1,2,3} and r+{8,9,10,11}, 32 bytes (two chunks of one aligned
// pair) each: row bits 1 and 3 tell those rows apart for every r, so the 16 (row parity, slot) pairs cover all 64 banks
// exactly once — conflict-free for every tap shift.  (conv_fast.h's key, tuned for ds_read_b128, is 2-way conflicted
// here: rows r and r+8 collide.)  The key repeats every 16 rows: row pitches / k-step strides are multiples of 16.
struct WRow {
  static __device__ __forceinline__ int key(int row) { return (row & 2) | ((row >> 1) & 4); }
  static __device__ __forceinline__ int off(int row, int c) { return row * 128 + ((c ^ key(row)) << 4); }
};

__device__ __forceinline__ u32x4 tr_pair(const unsigned char* lo, const unsigned char* hi) {
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_p)lo);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_p)hi);
  const u32x2 l2 = __builtin_bit_cast(u32x2, a), h2 = __builtin_bit_cast(u32x2, b);
  return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// One workgroup (8 waves) = (block of MB output channels, 64-input-channel chunk, a group of <= MAXT taps); it sweeps a
// strided share of the TH x TW pixel tiles and flushes its partial sums once with f32 atomics.
// Waves: ci_tile = wave % 4 (16 input channels each); the other factor 2 splits the output channels (MB = 64: 32 each)
// or the taps (MB = 32: even / odd tap slots).
// CIT: input-channel tiles across waves.  4 = a whole 64-channel chunk; 2 = layers with <= 32 input channels (PSMNet's 3-D
// convolutions, the 32-channel Conv2DownUp blocks): the two tiles that would multiply zero channels are not computed — the
// freed factor 2 goes to the tap split, so a wave carries half the accumulators, MFMAs and fragment reads per tile.
// (bx, gdx, by) = the workgroup's position: tile share bx of gdx, and the (cout block, depth tap, chunk, tap group) index by.
// A stand-alone launch passes blockIdx / gridDim; the grouped launch (wgrad_fast_group_kernel) passes the position inside
// the layer the workgroup was assigned to.
template <int TH, int TW, int MAXT, int MB, bool DMAX, int CIT>
__device__ __forceinline__ void wgrad_fast_body(const WgfArgs& p, const int bx, const int gdx, int by) {
  typedef bf16_t T;
  constexpr int V = 8, CK = 64, RB = 128;
  constexpr int NCO = 2;                     // output-channel MFMA tiles per wave
  constexpr int COG = (MB / 16) / NCO;       // output-channel groups across waves: 2 (MB 64) / 1 (MB 32)
  constexpr int TAPL = (8 / CIT) / COG;      // waves interleaved over the taps: 1 / 2 (CIT 4), 2 / 4 (CIT 2)
  constexpr int MAXTW = (MAXT + TAPL - 1) / TAPL;
  constexpr int NKS = TH * TW / 32;          // MFMA k-steps (32 pixels) per tile
  constexpr int RPR = 64;                    // LDS rows per load round of the 512 lanes
  constexpr int XPF = MAXT == 1 ? (TH * TW) / 64 : 5;   // register-path X loads per lane (rounds) per tile; a stride-1 1x1 halo IS the tile
  static_assert(TH * TW % 64 == 0 && (TW == 32 || TW == 16), "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int s = p.stride, d = p.dil;
  const int qb = p.qb, qsh = p.qsh;          // chunk packing (WgfArgs): rows of the X image are (pixel, chunk) pairs
  const int sw = qb ? qb : s;                // distance of horizontally adjacent output pixels in the X image, in rows
  const int IH = (TH - 1) * s + (p.kh - 1) * d + 1, IW = (TW - 1) * sw + (p.kw - 1) * d + 1;
  const int IWp = (IW + 15) & ~15;
  const int h_rows = IH * IWp;
  const int h_rounds = (h_rows + RPR - 1) / RPR;
  const int xbytes = h_rounds * RPR * RB;
  constexpr int ybytes = TH * TW * RB;
  constexpr int y_rounds = TH * TW / RPR;
  const int sbytes = xbytes + ybytes;        // one stage buffer: [X halo][dY tile]
  const int T_ = p.kh * p.kw;

  // by -> (output-channel block, depth tap, channel chunk, tap group)
  const int tgi = by % p.ntg; by /= p.ntg;
  const int q = by % p.nq; by /= p.nq;
  const int kdi = by % p.kd;
  const int mb = by / p.kd;
  const int t0 = tgi * p.tpb;
  const int nt = qb ? min(qb, p.nq_tot - q * qb) : min(p.tpb, T_ - t0);   // packed: "tap" j is channel chunk q*qb + j
  const int m0 = mb * MB;
  const int ci_tile = wave % CIT;
  const int rest = wave / CIT;
  const int co_tile0 = (rest % COG) * NCO;
  const int tap_lane = rest / COG;
  const int cin_q = qb ? CK : min(CK, p.Cin - q * CK);   // packed: every flushed chunk writes all 64 columns (pad columns are zeros)
  const int cout_m = min(MB, p.Cout - m0);

  f32x4 acc[MAXTW][NCO];
#pragma unroll
  for (int t = 0; t < MAXTW; ++t)
#pragma unroll
    for (int mi = 0; mi < NCO; ++mi) acc[t][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  // ---- fragment addresses (byte offsets inside a stage buffer), once per kernel ----
  // ds_read_b64_tr_b16: lane (p4 = l & 3, r4 = (l15 >> 2), lg) addresses pixel 8*lg + r4 (and + 4) of the k-step, 8-byte
  // column p4 of the 16-channel tile.
  const int p4 = lane & 3, r4 = l15 >> 2;
  const int sub = (p4 & 1) << 3;
  int a_lo[NCO], a_hi[NCO];                  // dY tile: row = pixel index in the tile
#pragma unroll
  for (int mi = 0; mi < NCO; ++mi) {
    const int c = (((co_tile0 + mi) * 16) >> 3) + (p4 >> 1);
    const int pa = 8 * lg + r4;
    a_lo[mi] = xbytes + WRow::off(pa, c) + sub;
    a_hi[mi] = xbytes + WRow::off(pa + 4, c) + sub;
  }
  // X halo: row(k-step ks, tap) = ks * KROWS + lane part + tap offset, KROWS a multiple of 16 (swizzle key unchanged)
  const int pa0 = 8 * lg + r4;               // pixel inside the k-step
  const int ptw = pa0 % TW;
  const int lrow = ((pa0 / TW) * s) * IWp + (qb ? (ptw >> 1) * 2 * qb + (ptw & 1) : ptw * s);
  const int tapr = qb ? 2 : 1;               // rows between consecutive "taps" (packed: chunks of one pixel pair interleave)
  // swizzle key of an X row: that of its PIXEL when chunks are packed (the line pitch is a power of two then)
  auto xkey = [&](int row) {
    if (!qb) return WRow::key(row);
    const int iw = row & (IWp - 1);
    return WRow::key(((iw >> (qsh + 1)) << 1) | (iw & 1));
  };
  auto xoff = [&](int row, int c) { return row * 128 + ((c ^ xkey(row)) << 4); };
  const int krows = (32 / TW) * s * IWp;
  const int cx = ((ci_tile * 16) >> 3) + (p4 >> 1);
  int b_lo[MAXTW], b_hi[MAXTW];
  {
    // taps t0 + tap_lane, + TAPL, ...: one division for the first, then a carry walk (kernel start-up is on the critical
    // path of the many tiny launches)
    int khi = (t0 + tap_lane) / p.kw, kwi = (t0 + tap_lane) - khi * p.kw;
    const int row_first = lrow + ((t0 / p.kw) * d) * IWp + (t0 % p.kw) * d;   // tap t0: what idle slots read
#pragma unroll
    for (int tl = 0; tl < MAXTW; ++tl) {
      const int tt = tl * TAPL + tap_lane;
      const int row = tt < nt ? lrow + (khi * d) * IWp + kwi * d * tapr : row_first;   // idle slot: result never flushed
      b_lo[tl] = xoff(row, cx) + sub;
      b_hi[tl] = xoff(row + 4 * sw, cx) + sub;
      kwi += TAPL;
      while (kwi >= p.kw) { kwi -= p.kw; ++khi; }
    }
  }

  // ---- staging (linear LDS destination, swizzled source: see conv_fast.h) ----
  const int rsub = tid >> 3;
  const int c_l = (tid & 7) ^ WRow::key(rsub);   // logical 16-byte chunk this lane fetches (same in every 64-row round)
  const int c_lx = (tid & 7) ^ xkey(rsub);      // ... of the X image
  const int tid16 = tid * 16;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
  const unsigned magic_iwp = div_magic(IWp);
  // first input channel of the lane's chunk; packed: the chunk index is the row's position inside its pixel, which is the
  // same in every round (64 and the row pitch are multiples of qb)
  const int chx = (qb ? q * qb + ((rsub >> 1) & (qb - 1)) : q) * CK + c_lx * V;
  const bool okx = chx < p.Cin;
  const int chy = c_l * V;                   // channel inside the output-channel block
  const bool oky = chy < cout_m;

  const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
  const int ntiles = p.B * p.Do * tiles_h * tiles_w;

  float psc[V], psf[V];                      // prologue coefficients of the lane's 8 channels (per tile: the group may change)

  // Tile walk without divisions: (image, tile row, tile column) of the NEXT tile to issue advance by gdx tiles.
  int n_img, n_ty, n_tx;
  {
    const int t0i = bx;
    n_img = t0i / (tiles_h * tiles_w);
    const int tr = t0i - n_img * tiles_h * tiles_w;
    n_ty = tr / tiles_w; n_tx = tr - n_ty * tiles_w;
  }
  const int step_tx = gdx % tiles_w, step_r = gdx / tiles_w;
  const int step_ty = step_r % tiles_h, step_img = step_r / tiles_h;

  // Per-lane source offsets of an INTERIOR tile (no padding, no ragged edge): tile base + constant offset per round.
  // Lanes of dead columns / past the tile / past the channels read some valid address; their LDS rows are never used
  // (or only feed dW columns nobody reads).
  constexpr int XR = 8;
  int xo[XR], yo[y_rounds];
  auto xpix = [&](int iw) { return qb ? (((iw >> (qsh + 1)) << 1) | (iw & 1)) : iw; };   // pixel column of a halo-line position
#pragma unroll
  for (int j = 0; j < XR; ++j) {
    const int row = rsub + j * RPR;
    const int ih = fast_div(row, IWp, magic_iwp), iw = row - ih * IWp;
    xo[j] = (okx && row < h_rows && iw < IW) ? (ih * p.W + xpix(iw)) * p.ldx : 0;   // dead pitch columns re-read the tile's first pixel (no traffic)
  }
#pragma unroll
  for (int j = 0; j < y_rounds; ++j) {
    const int pix = rsub + j * RPR;
    const int th = pix / TW, tw = pix - th * TW;   // TW is a power of two
    yo[j] = oky ? (th * p.Wo + tw) * p.lddy : 0;
  }

  struct TileGeo { const T* xb; const T* yb; int ih0, iw0, oh0, ow0, grp; bool slice_ok; };
  auto next_geo = [&]() {      // geometry of the next tile to issue; advances the walk
    TileGeo g;
    g.oh0 = n_ty * TH; g.ow0 = n_tx * TW;
    int b = n_img, dz = 0;
    if (p.Do > 1) { b = n_img / p.Do; dz = n_img - b * p.Do; }
    const int din = dz * p.sd + kdi - p.pad_d;
    g.slice_ok = din >= 0 && din < p.D;
    g.grp = b >= p.bpg ? b / p.bpg : 0;
    g.xb = (const T*)p.x + ((long)b * p.D + (g.slice_ok ? din : 0)) * p.H * p.W * p.ldx + chx;
    g.yb = (const T*)p.dy + (long)n_img * p.Ho * p.Wo * p.lddy + m0 + chy;
    g.ih0 = g.oh0 * s - p.pad_t; g.iw0 = g.ow0 * s - p.pad_l;
    n_tx += step_tx; if (n_tx >= tiles_w) { n_tx -= tiles_w; ++n_ty; }
    n_ty += step_ty; if (n_ty >= tiles_h) { n_ty -= tiles_h; ++n_img; }
    n_img += step_img;
    return g;
  };
  auto x_src = [&](const TileGeo& g, int j) -> const T* {
    const int row = rsub + j * RPR;
    const int ih = fast_div(row, IWp, magic_iwp), iw = row - ih * IWp;
    const int gh = g.ih0 + ih, gw = g.iw0 + xpix(iw);
    const bool in = okx && g.slice_ok && iw < IW && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
    return in ? g.xb + (gh * p.W + gw) * p.ldx : (const T*)sdhip_zero16;
  };
  auto y_src = [&](const TileGeo& g, int j) -> const T* {
    const int pix = rsub + j * RPR;
    const int th = pix / TW, tw = pix - th * TW;
    const int oh = g.oh0 + th, ow = g.ow0 + tw;
    const bool in = oky && oh < p.Ho && ow < p.Wo;
    return in ? g.yb + (oh * p.Wo + ow) * p.lddy : (const T*)sdhip_zero16;
  };
  auto issue = [&](unsigned char* buf) {     // LDS-DMA of the next tile (no prologue: DMAX)
    const TileGeo g = next_geo();
    const unsigned base = (unsigned)(buf - smem) + wave_lds;
    const bool yin = g.oh0 + TH <= p.Ho && g.ow0 + TW <= p.Wo;                      // wave-uniform
    const bool xin = g.slice_ok && h_rounds <= XR && g.ih0 >= 0 && g.iw0 >= 0 && g.ih0 + IH <= p.H && g.iw0 + (qb ? TW : IWp) <= p.W;
    if (yin) {
      const T* yb = oky ? g.yb + (g.oh0 * p.Wo + g.ow0) * p.lddy : (const T*)sdhip_zero16;
#pragma unroll
      for (int j = 0; j < y_rounds; ++j) glds16(yb + yo[j], base + xbytes + j * 8192);
    } else {
#pragma unroll
      for (int j = 0; j < y_rounds; ++j) glds16(y_src(g, j), base + xbytes + j * 8192);
    }
    const T* xbi = okx ? g.xb + (g.ih0 * p.W + g.iw0) * p.ldx : (const T*)sdhip_zero16;
    if (xin) {
#pragma unroll
      for (int j = 0; j < XR; ++j)
        if (j < h_rounds) glds16(xbi + xo[j], base + j * 8192);
    } else {
      for (int j = 0; j < h_rounds; ++j) glds16(x_src(g, j), base + j * 8192);
    }
  };
  (void)issue;
  int cur_grp = -1;

  // ---- one tile's arithmetic on a filled stage buffer ----
  auto compute = [&](unsigned char* buf) {
    if (p.dbias && q == 0 && tgi == 0 && kdi == p.pad_d) {  // uniform: bias gradient = column sums of the dY tile (once)
      const int ch = tid % CK, stripe = tid / CK;
      if (ch < cout_m) {
        const int c = ch >> 3, e = ch & 7;
        for (int pix = stripe; pix < TH * TW; pix += 512 / CK)
          bsum += bf2f(*reinterpret_cast<const T*>(buf + xbytes + WRow::off(pix, c) + e * 2));
      }
    }

    // dY fragments of the whole tile, then every tap walks over the tile
    u32x4 af[NKS][NCO];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int mi = 0; mi < NCO; ++mi) af[ks][mi] = tr_pair(buf + a_lo[mi] + ks * (32 * RB), buf + a_hi[mi] + ks * (32 * RB));
    // X fragments run PD (tap, k-step) steps ahead of the MFMAs that consume them: an LDS read needs ~100+ cycles, a
    // step's two MFMAs cover 32 (with one step of lookahead every tap stalls on its own read).
    constexpr int NST = NKS * MAXTW, PD = NST < 4 ? NST : 4;
    u32x4 bfr[PD];
    auto ldb = [&](int u) -> u32x4 {
      const int ks = u / MAXTW, tl = u % MAXTW;
      const unsigned char* kb = buf + ks * krows * RB;
      return tr_pair(kb + b_lo[tl], kb + b_hi[tl]);
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) bfr[u] = ldb(u);
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int ks = u / MAXTW, tl = u % MAXTW;
      const u32x4 bf = bfr[u % PD];
#pragma unroll
      for (int mi = 0; mi < NCO; ++mi) Mma<T>::run(acc[tl][mi], af[ks][mi], bf);
      if (u + PD < NST) bfr[u % PD] = ldb(u + PD);
      __builtin_amdgcn_sched_barrier(0);   // keep the lookahead: the scheduler otherwise sinks each read next to its use
    }
  };

  // ---- sweep ----
  const int ntl = (ntiles - bx + gdx - 1) / gdx;   // this workgroup's tiles
  if constexpr (DMAX) {
    if (ntl > 0) issue(smem);
    for (int it = 0; it < ntl; ++it) {
      unsigned char* buf = smem + (it & 1) * sbytes;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's DMA has landed
      __syncthreads();                                    // tile visible; everybody has finished tile it-1 (other buffer)
      if (it + 1 < ntl) issue(smem + ((it + 1) & 1) * sbytes);
      compute(buf);
    }
  } else {
    // Register path (BatchNorm prologue on X): BOTH operands of a tile travel through registers, two tiles ahead.  With
    // the loads of tile it+1 issued after the barrier of tile it (the DMA scheme above) a workgroup of the small-map
    // layers spends most of each tile waiting for HBM: its MFMA phase (8..72 MFMAs per wave) is far shorter than a load.
    // Two register sets double the time a load has to land, and since every load is a plain register load the compiler
    // places exact vmcnt waits (LDS-DMA is invisible to it and would force vmcnt(0)).
    struct RegSet { u32x4 x[XPF]; u32x4 y[y_rounds]; bool rin[XPF]; int grp; bool all; };
    RegSet ra, rb;
    auto load = [&](RegSet& r) {
      const TileGeo g = next_geo();
      const bool yin = g.oh0 + TH <= p.Ho && g.ow0 + TW <= p.Wo;                      // wave-uniform
      const bool xin = g.slice_ok && h_rounds <= XR && g.ih0 >= 0 && g.iw0 >= 0 && g.ih0 + IH <= p.H && g.iw0 + (qb ? TW : IWp) <= p.W;
      r.grp = g.grp; r.all = xin;
      const T* xbi = okx ? g.xb + (g.ih0 * p.W + g.iw0) * p.ldx : (const T*)sdhip_zero16;
      if (xin) {
#pragma unroll
        for (int j = 0; j < XPF; ++j)
          if (j < h_rounds) r.x[j] = *reinterpret_cast<const u32x4*>(xbi + xo[j]);
      } else {
#pragma unroll
        for (int j = 0; j < XPF; ++j) {
          if (j < h_rounds) {
            const T* src = x_src(g, j);
            r.rin[j] = src != (const T*)sdhip_zero16;
            r.x[j] = *reinterpret_cast<const u32x4*>(src);
          }
        }
      }
      if (yin) {
        const T* yb = oky ? g.yb + (g.oh0 * p.Wo + g.ow0) * p.lddy : (const T*)sdhip_zero16;
#pragma unroll
        for (int j = 0; j < y_rounds; ++j) r.y[j] = *reinterpret_cast<const u32x4*>(yb + yo[j]);
      } else {
#pragma unroll
        for (int j = 0; j < y_rounds; ++j) r.y[j] = *reinterpret_cast<const u32x4*>(y_src(g, j));
      }
    };
    auto store = [&](RegSet& r, unsigned char* buf) {   // BatchNorm prologue on X, then both operands into LDS
      if (p.in_scale && r.grp != cur_grp) {   // wave-uniform; the group changes at most once per sweep
        cur_grp = r.grp;
        const float* sc = p.in_scale + cur_grp * p.Cin + chx;
        const float* sf = p.in_shift + cur_grp * p.Cin + chx;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const bool okc = chx + e < p.Cin;
          psc[e] = okc ? sc[okc ? e : 0] : 0.f;
          psf[e] = okc ? sf[okc ? e : 0] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        if (j < h_rounds) {
          const bool in = r.all || r.rin[j];
          u32x4 raw = in ? r.x[j] : u32x4{0u, 0u, 0u, 0u};
          if (in && p.in_scale) {
            float f[V];
            Chunk<T>::unpack(raw, f);
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float v = fmaf(f[e], psc[e], psf[e]);   // pad channels (>= Cin) only feed dW columns nobody reads
              f[e] = p.in_relu ? fmaxf(v, 0.f) : v;
            }
            raw = Chunk<T>::pack(f);
          }
          *reinterpret_cast<u32x4*>(buf + j * 8192 + tid16) = raw;
        }
      }
#pragma unroll
      for (int j = 0; j < y_rounds; ++j) *reinterpret_cast<u32x4*>(buf + xbytes + j * 8192 + tid16) = r.y[j];
    };
    if constexpr (MAXT <= 9) {
      if (ntl > 0) load(ra);
      if (ntl > 1) load(rb);
      for (int it = 0; it < ntl; it += 2) {
        store(ra, smem);
        __syncthreads();                    // tile visible; everybody has finished tile it-2 in this buffer (see tile it-1's barrier)
        if (it + 2 < ntl) load(ra);
        compute(smem);
        if (it + 1 < ntl) {                 // uniform
          store(rb, smem + sbytes);
          __syncthreads();
          if (it + 3 < ntl) load(rb);
          compute(smem + sbytes);
        }
      }
    } else {                                // 25 taps: the accumulators leave no room for a second register set
      if (ntl > 0) load(ra);
      for (int it = 0; it < ntl; ++it) {
        unsigned char* buf = smem + (it & 1) * sbytes;
        store(ra, buf);
        __syncthreads();
        if (it + 1 < ntl) load(ra);
        compute(buf);
      }
    }
  }

  // ---- flush: f32 atomics into the packed gradient buffer [kd][nq][T][Mpad][CK] ----
  if (ci_tile * 16 < cin_q && co_tile0 * 16 < cout_m) {   // wave-uniform
#pragma unroll
    for (int tl = 0; tl < MAXTW; ++tl) {
      const int tt = tl * TAPL + tap_lane;
      if (tt < nt) {
        float* dst = p.dwp + ((long)((kdi * p.nq + q) * T_ + t0 + tt) * p.Mpad) * CK;
#pragma unroll
        for (int mi = 0; mi < NCO; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + (co_tile0 + mi) * 16 + 4 * lg + r;
            if (m < p.Mpad) atomicAdd(dst + (long)m * CK + ci_tile * 16 + l15, acc[tl][mi][r]);
          }
        }
      }
    }
  }
  if (p.dbias && q == 0 && tgi == 0 && kdi == p.pad_d) {
    const int ch = tid % CK;
    if (ch < cout_m) atomicAdd(p.dbias + m0 + ch, bsum);
  }
}

template <int TH, int TW, int MAXT, int MB, bool DMAX, int CIT = 4>
__global__ __launch_bounds__(512) void wgrad_fast_kernel(const WgfArgs p) {
  wgrad_fast_body<TH, TW, MAXT, MB, DMAX, CIT>(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// ---- grouped launch: the weight gradients of SEVERAL layers in one grid -----------------------------------------------
// Nothing downstream of a weight gradient runs before the optimizer, so the step queues them and launches every bucket of
// layers that share an instantiation as ONE grid (sdhip_conv2d_wgrad_group): the small layers of a DenseNet block fill 64
// of 256 CUs each when launched alone (their workgroup count is capped by the flush cost per workgroup), a bucket of them
// fills the chip, and the dependent-node floor (~4.4 us) is paid once per bucket instead of once per layer.
// The table travels BY VALUE in the kernel arguments (<= 4 KB): a captured step needs no device-side table and no copy node.
constexpr int kWgfGroupMax = 16;
struct WgfGroup {
  int n;
  int wg0[kWgfGroupMax + 1];                 // first workgroup of layer i (prefix sums of gx[i] * gy[i])
  int gx[kWgfGroupMax];                      // tile shares of layer i
  WgfArgs L[kWgfGroupMax];
};
static_assert(sizeof(WgfGroup) <= 4096, "kernel argument block");

template <int TH, int TW, int MAXT, int MB, bool DMAX, int CIT = 4>
__global__ __launch_bounds__(512) void wgrad_fast_group_kernel(const WgfGroup g) {
  const int b = (int)blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.wg0[i + 1]) ++i;          // scalar walk over <= 16 entries
  const int local = b - g.wg0[i];
  const int gx = g.gx[i];
  const int by = local / gx;
  wgrad_fast_body<TH, TW, MAXT, MB, DMAX, CIT>(g.L[i], local - by * gx, gx, by);
}

// What launching one layer needs, decided on the host; `group` != nullptr marks a plan that may join a grouped launch.
struct WgfPlan {
  const void* single; const void* group;     // the two kernels of the instantiation
  size_t lds;
  int ntiles, gy, gx, occ;
  double t_tile_us, flush_us;                // cost model: one tile of one workgroup; one workgroup's closing atomics
  WgfArgs a;
};

// Returns 0 and fills `pl` (gx = the stand-alone grid), 1 when the tile does not fit (caller tries a smaller one).
template <int TH, int TW, int MAXT, int MB, bool DMAX, int CIT = 4>
int plan_wgf(const WgfArgs& a, WgfPlan& pl) {
  auto kern = wgrad_fast_kernel<TH, TW, MAXT, MB, DMAX, CIT>;
  auto gkern = wgrad_fast_group_kernel<TH, TW, MAXT, MB, DMAX, CIT>;
  const int IH = (TH - 1) * a.stride + (a.kh - 1) * a.dil + 1, IW = (TW - 1) * (a.qb ? a.qb : a.stride) + (a.kw - 1) * a.dil + 1;
  const int IWp = (IW + 15) & ~15;
  const int hrounds = (IH * IWp + 63) / 64;
  const size_t lds = 2 * ((size_t)hrounds * 8192 + (size_t)TH * TW * 128);
  if (lds > 160 * 1024) return 1;                 // caller tries a smaller tile
  if (!DMAX && hrounds > (MAXT == 1 ? (TH * TW) / 64 : 5)) return 1;             // register-path prefetch plan (XPF in the kernel)
  if (a.qb && (IWp % 64) && (64 % IWp)) return 1;   // packed rows: a lane must meet the same chunk in every load round
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gkern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  const int nmb = sdhip_cdiv(a.Cout, MB);
  const int gy = nmb * a.kd * a.nq * a.ntg;
  const int ntiles = a.B * a.Do * sdhip_cdiv(a.Ho, TH) * sdhip_cdiv(a.Wo, TW);
  // Workgroups per (cout block, chunk, tap group): each one sweeps ntiles/gx tiles and then flushes its whole partial
  // sum with f32 atomics, which execute at the memory side at ~1.3 TB/s chip-wide.  On small feature maps the flush of
  // 256 workgroups costs more than their MFMAs: balance  (ntiles/gx) * t_tile  against  gx * flush bytes / 1.3 TB/s.
  const double tune_rate = sdhip_diag().tune_atomic_tbs;
  const double t_tile_us = 0.35 + 0.17 * a.tpb;
  const double flush_us = (double)a.tpb * MB * 64 * 4 / (tune_rate * 1e6);
  int gx = (int)(sqrt((double)ntiles * t_tile_us / flush_us) + 0.5);
  // co-resident 8-wave workgroups per CU for this instantiation and LDS size (registers allow two for the 1x1 / narrow
  // variants): small problems want all their workgroups in flight at once
  static size_t occ_lds = 0;
  static int occ = 1;
  if (occ_lds != lds) {
    int nb = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 512, lds) != hipSuccess || nb < 1) nb = 1;
    occ = nb > 2 ? 2 : nb;
    occ_lds = lds;
  }
  if (gx > sdhip_cdiv(256 * occ, gy)) gx = sdhip_cdiv(256 * occ, gy);
  if (gx > sdhip_cdiv(ntiles, 2)) gx = sdhip_cdiv(ntiles, 2);     // at least two tiles per workgroup: the pipeline overlaps them
  if (gx < 1) gx = 1;
  pl.single = (const void*)kern; pl.group = (const void*)gkern;
  pl.lds = lds; pl.ntiles = ntiles; pl.gy = gy; pl.gx = gx; pl.occ = occ;
  pl.t_tile_us = t_tile_us; pl.flush_us = flush_us; pl.a = a;
  return 0;
}

inline int launch_wgf_plan(const WgfPlan& pl, hipStream_t s) {
  WgfArgs a = pl.a;
  void* args[] = {&a};
  if (hipLaunchKernel(pl.single, dim3(pl.gx, pl.gy), dim3(512), args, pl.lds, s) != hipSuccess)
    SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: launch failed: %s", hipGetErrorString(hipGetLastError()));
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// out == nullptr: launch now.  out != nullptr: only plan (the grouped entry point collects plans first).
template <int TH, int TW, int MAXT, int MB, bool DMAX, int CIT = 4>
int launch_wgf(const WgfArgs& a, hipStream_t s, WgfPlan* out) {
  WgfPlan pl;
  const int rc = plan_wgf<TH, TW, MAXT, MB, DMAX, CIT>(a, pl);
  if (rc != 0) return rc;
  if (out) { *out = pl; return SDHIP_OK; }
  return launch_wgf_plan(pl, s);
}

// tile choice: wide maps 4x32, else 4x16; 1 = nothing fits (caller falls back to the general kernel)
template <int MAXT, int MB, bool DMAX, int CIT = 4>
int launch_wgf_tile(const WgfArgs& a, hipStream_t s, WgfPlan* out) {
  if (a.Wo >= 24) {
    const int rc = launch_wgf<4, 32, MAXT, MB, DMAX, CIT>(a, s, out);
    if (rc != 1) return rc;
  }
  return launch_wgf<4, 16, MAXT, MB, DMAX, CIT>(a, s, out);
}

template <bool DMAX>
int launch_wgf_taps(const WgfArgs& a, int T, hipStream_t s, WgfPlan* out) {
  const bool narrow = a.Cout <= 32;   // 32 output channels per workgroup: the second wave group takes the odd taps instead
  if (T == 1) return narrow ? launch_wgf_tile<1, 32, DMAX>(a, s, out) : launch_wgf_tile<1, 64, DMAX>(a, s, out);
  const bool half = a.Cin <= 32 && !sdhip_diag().wgrad_no_half;   // two input-channel tiles instead of four (CIT)
  if (a.tpb <= 9) {
    if (half) return narrow ? launch_wgf_tile<9, 32, DMAX, 2>(a, s, out) : launch_wgf_tile<9, 64, DMAX, 2>(a, s, out);
    return narrow ? launch_wgf_tile<9, 32, DMAX>(a, s, out) : launch_wgf_tile<9, 64, DMAX>(a, s, out);
  }
  return half ? launch_wgf_tile<25, 32, DMAX, 2>(a, s, out) : launch_wgf_tile<25, 32, DMAX>(a, s, out);
}

// chunk-packed 1x1 weight gradient (WgfArgs::qb): 4x16 pixel tiles, up to 4 chunks as taps
template <bool DMAX>
int launch_wgf_packed(const WgfArgs& a, hipStream_t s, WgfPlan* out) { return launch_wgf<4, 16, 4, 64, DMAX>(a, s, out); }

}  // namespace
