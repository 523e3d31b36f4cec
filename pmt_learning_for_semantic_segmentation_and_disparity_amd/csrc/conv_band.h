// Persistent K x K convolution for the full-resolution 5x5 layers (Conv2DownUp, models/dsnet_t2.py:80-117 of the reference;
// the ten launches that lead the training step's profile) and the 3x3 layers with <= 32 input channels: bf16, stride 1, no
// dilation, <= 64 input and output channels.
//
// What the tap-group pipeline of conv_fast.h cannot do inside its 80 KiB (two workgroups per CU): keep more than one tap
// of 64x64 weights per stage next to a 64-channel halo tile — 25 stages of 32 MFMAs per wave, a barrier and a DMA drain
// between each.  This kernel owns the whole CU instead (one 512-thread workgroup, ~144 KiB of LDS) and is persistent:
//
//   tile    16 x 32 output pixels x all output channels; wave w computes pixel rows 2w, 2w+1 (4 MFMA pixel tiles) times
//           BN output channels, exactly the register tile of conv_fast.h
//   chunk   (tile, 32-channel half of the input): its halo image, (16+K-1) x 40 rows of 64 bytes, is 50 KiB, so TWO fit:
//           the halo of chunk c+1 streams in by LDS-DMA, a couple of rounds per stage, while chunk c is computed
//   stage   one kernel ROW of a chunk: K taps x BN rows of 64 bytes of weights (20 KiB), double-buffered; K * 4 * BN/16
//           MFMAs per wave between two barriers (80 for 5x5 / 64 channels, against 32).  3x3: the whole kernel is one stage and
//           its nine taps of weights stay resident (BandCfg::WRES), so the stage loop issues halo DMA only
//
// Every fragment address is  buffer + wave row + compile-time constant + one of K per-lane registers:  with the halo
// row pitch (40) and the 16-pixel column step multiples of 8 rows, the swizzle key of a pixel depends on
// (kw + lane) only, so the inner loop carries no address arithmetic at all.
//
// LDS-DMA bookkeeping (the DMA is inline asm, invisible to the compiler's wait-count pass): the weights of the next
// stage are issued first after a barrier, the halo rounds of the next chunk after them; `s_waitcnt vmcnt(n)` with
// n = the halo instructions issued since then waits for exactly the weights; at a tile boundary n = the epilogue's
// stores (vmcnt retires loads, stores and LDS-DMA together, in issue order).
#pragma once
#include <type_traits>
#include "conv_fast.h"

namespace {

struct BandArgs {
  const void* x; const void* wp; void* y;
  const float* bias; double* stats;
  int B, H, W, Ho, Wo, pad_t, pad_l;
  int Cin, ldx, Cout, Mpad, ldy;
  int bpg, act, stats_ld, nrep;
  const void* res; int ldres;              // y = result + res (same geometry as y; res == y: accumulate in place); nullptr: off
  long rep_stride;
  int tiles_w, tiles_hw, ntiles, tpw;      // tiles per output row / per image / in total / per workgroup
  unsigned magic_hw, magic_tw;             // div_magic(tiles_hw), div_magic(tiles_w)
  int D, Do, pad_d;                        // volumes (KD = 3): input / output depth, front padding; B counts batch entries then
  unsigned magic_do;                       // div_magic(Do)
  int dbg;                                 // diagnostic bits (SDHIP_TUNE_BAND_DBG): 1 no halo prefetch, 2 no weight DMA, 4 no MFMA, 8 no stores
};

template <int I, int N, typename F> __device__ __forceinline__ void band_static_for(F&& f) {   // f(integral_constant<I>) ... <N-1>
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); band_static_for<I + 1, N>(f); }
}
template <int N> __device__ __forceinline__ void band_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int K, int BN, int NW = 8, int KD = 1>
struct BandCfg {
  static constexpr int RPR = NW * 16;                    // LDS rows (64 bytes) one DMA round of the workgroup fills
  static constexpr int RPW = 16 / NW;                    // output pixel rows per wave
  static constexpr int TH = 16, TW = 32;
  static constexpr int IH = TH + K - 1, IW = TW + K - 1, IWp = (IW + 7) & ~7;
  static constexpr int HROWS = IH * IWp;                 // halo rows (64 bytes each) of one chunk
  static constexpr int HR = (HROWS + RPR - 1) / RPR;     // LDS-DMA rounds (NW x 64 lanes x 16 bytes = RPR rows) per halo image
  static constexpr int HB = HROWS * 64;
  // A stage = RS kernel rows.  5x5: one row per stage, the weights of the next stage double-buffered.  3x3 (<= 32 input
  // channels only): the whole kernel is ONE stage per chunk and its 9 taps of weights (<= 37 KB) stay RESIDENT — every
  // chunk of every tile uses the same ones — so the stage loop issues halo DMA only.
  static constexpr int RS = K == 3 ? 3 : 1;
  static constexpr int NSTG = K / RS;                    // stages per chunk
  static constexpr bool WRES = NSTG == 1 && KD == 1;     // weights resident (host: single channel half); KD = 3: every depth tap is a chunk with weights of its own, streamed like the 5x5 stages
  static constexpr int TPS = RS * K;                     // taps per stage
  static constexpr int WROWS = TPS * BN;
  static constexpr int WR = (WROWS + RPR - 1) / RPR;
  static constexpr int WB = WROWS * 64;
  static constexpr int NWB = WRES ? 1 : 2;               // weight buffers
  static constexpr int WRS = WRES ? 0 : WR;              // weight DMA slots per stage
  // halo rounds of the next chunk issued per stage: spread over stages 0 .. NSTG-2, or all in the only stage
  static constexpr int HPS = NSTG == 1 ? HR : (HR + NSTG - 2) / (NSTG - 1);
  static constexpr int RED = NW * 2 * BN * 4;            // statistics scratch: [NW waves][2][BN] floats
  static constexpr int SPT = (WRS + HPS + TPS - 1) / TPS;   // DMA slots per tap
  static constexpr size_t LDS = 2 * HB + NWB * WB + RED;
  static_assert(HROWS % 16 == 0 && WROWS % 16 == 0, "a wave's 16 rows of a DMA round are either all inside or all outside the image");
  static_assert(NSTG == 1 || HPS * (NSTG - 1) >= HR, "halo rounds must fit the stages of one chunk");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// One LDS-DMA instruction through a buffer resource: lane address = base + soff + voff; a lane whose voff is not below
// the resource's num_records fetches zeros (padding, dead channels) — no zero page, no 64-bit address arithmetic.
typedef __attribute__((ext_vector_type(4))) int band_rsrc_t;
constexpr int kBandOob = 0x7fffff00;           // num_records of every resource and the voff of a padding lane
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void band_dma(int voff, band_rsrc_t rsrc, unsigned lds_wave_base, int soff) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               : : "v"(voff), "s"(rsrc), "s"(lds_wave_base), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ band_rsrc_t band_rsrc(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  band_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));   // stride 0: raw buffer
  r[2] = kBandOob;
  r[3] = 0x00020000;
  return r;
}

template <int K, int BN, int VAR, int NW, int KD = 1>
__global__ __launch_bounds__(NW * 64) void conv_band_kernel(const BandArgs p) {
  constexpr bool DBG = (VAR & 128) != 0;                  // diagnostic build: honours p.dbg (timing breakdowns, wrong results)
  const int dbg = DBG ? p.dbg : 0;
  using C = BandCfg<K, BN, NW, KD>;
  using T = bf16_t;
  constexpr int IWp = C::IWp, IW = C::IW, HB = C::HB, WB = C::WB, RPR = C::RPR, RPW = C::RPW;
  constexpr int NT_CO = BN / 16, NT_PIX = 2 * RPW;
  constexpr int NSTORE = NT_CO * NT_PIX / 2;             // global stores per lane of an interior tile's epilogue
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;

  const int t_begin = blockIdx.x * p.tpw;
  const int t_end = min(p.ntiles, t_begin + p.tpw);
  if (t_begin >= t_end) return;                          // (host launches no such workgroup)
  const int nhalf = KD > 1 ? KD : (p.Cin > 32 ? 2 : 1);   // chunks per tile: channel halves, or (volumes, <= 32 channels) depth taps
  const int nchunks = (t_end - t_begin) * nhalf;

  // ---- LDS-DMA source side ----
  // lane tid fills physical 16-byte slot (tid & 3) of row round*128 + (tid >> 2); it fetches the logical chunk
  // slot ^ key(row), key = bit 2 of the row -> bit 1 of the slot (LdsRow<1>), the same in every round (128 % 8 == 0)
  const int rsub = tid >> 2;
  const int c_l = (tid & 3) ^ ((rsub >> 1) & 2);
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
  const band_rsrc_t xr = band_rsrc(p.x), wr = band_rsrc(p.wp);
  const int img_bytes = p.H * p.W * p.ldx * 2;

  // volumes: z runs fastest, so that a workgroup walks its spatial tile through consecutive output slices and finds two of the
  // three input slices of a tile in its XCD's L2 (it loaded them for the tile before); b counts batch entries, zt is the slice
  int czt = 0;                                            // output slice of the tile decoded last
  auto tile_of = [&](int t, int& b, int& oh0, int& ow0) {
    if constexpr (KD > 1) { const int q = fast_div(t, p.Do, p.magic_do); czt = t - q * p.Do; t = q; }
    b = fast_div(t, p.tiles_hw, p.magic_hw);
    const int r = t - b * p.tiles_hw;
    const int ty = fast_div(r, p.tiles_w, p.magic_tw);
    oh0 = ty * C::TH;
    ow0 = (r - ty * p.tiles_w) * C::TW;
  };
  // byte offsets (inside one image, channel half 0) of this lane's piece of every halo round of a tile
  int hv[C::HR];
  auto halo_offsets = [&](int oh0, int ow0) {
    const int ih0 = oh0 - p.pad_t, iw0 = ow0 - p.pad_l;
#pragma unroll
    for (int r = 0; r < C::HR; ++r) {
      const int row = r * RPR + rsub;
      const int ih = row / IWp, iw = row - ih * IWp;
      const int gh = ih0 + ih, gw = iw0 + iw;
      const bool in = iw < IW && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W && c_l * 8 < p.Cin;
      hv[r] = in ? ((gh * p.W + gw) * p.ldx + c_l * 8) * 2 : kBandOob;
    }
  };
  auto halo_dma = [&](int r, int soff, int hsel) {         // round r of the image at byte offset soff (< 0: a slice outside the volume, zeros) into halo buffer hsel
    if (r < C::HR && r * RPR + wave * 16 < C::HROWS) band_dma(soff >= 0 ? hv[r] : kBandOob, xr, hsel * HB + r * (RPR * 64) + wave_lds, soff >= 0 ? soff : 0);   // wave-uniform
  };
  // byte offset of the image a chunk reads: (batch entry / image b, chunk h of output slice zt)
  auto chunk_src = [&](int b, int zt, int h) -> int {
    if constexpr (KD > 1) {
      const int zin = zt + h - p.pad_d;
      return (zin >= 0 && zin < p.D) ? (b * p.D + zin) * img_bytes : -1;
    } else {
      return b * img_bytes + h * 64;
    }
  };
  // weights: row tap*BN + m of a stage <- row (j*K + tap)*Mpad + m of the packed [T][Mpad][64] image
  int wv[C::WR];
#pragma unroll
  for (int r = 0; r < C::WR; ++r) {
    const int row = r * RPR + rsub;
    const int tap = row / BN, m = min(row % BN, p.Mpad - 1);
    wv[r] = (tap * p.Mpad + m) * 128 + c_l * 16;
  }
  const int wrow_bytes = C::TPS * p.Mpad * 128;            // one stage (RS kernel rows) of the packed image
  auto w_dma = [&](int r, int j, int h, int wsel) {         // stage j of chunk h: channel half h (+64 bytes in the packed row) or depth tap h (its own 3x3 image)
    if (r < C::WR && r * RPR + wave * 16 < C::WROWS)
      band_dma(wv[r], wr, 2 * HB + wsel * WB + r * (RPR * 64) + wave_lds, j * wrow_bytes + (KD > 1 ? h * wrow_bytes : h * 64));
  };

  // ---- fragment addressing ----
  int b_k[K];                                             // halo byte offset of (row kw + l15, logical chunk lg)
#pragma unroll
  for (int kw = 0; kw < K; ++kw) b_k[kw] = LdsRow<1>::off(kw + l15, lg);
  const int a_base = LdsRow<1>::off(l15, lg);             // weight row l15 (+ tap*BN + mi*16 rows: key unchanged)
  const int wave_row = wave * (RPW * IWp * 64);

  float bv[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = mi * 16 + 4 * lg + r;
      bv[mi][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }

  f32x4 acc[NT_CO][NT_PIX];
  float s1[NT_CO][4], s2[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }

  float* const red = reinterpret_cast<float*>(smem + 2 * HB + C::NWB * WB);
  auto flush_stats = [&](int grp) {                        // every wave; ends with the sums of `grp` added to p.stats
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[mi][r]), c2 = row16_sum(s2[mi][r]);
        if (l15 == 0) {
          const int m = mi * 16 + 4 * lg + r;
          red[(wave * 2 + 0) * BN + m] = a;
          red[(wave * 2 + 1) * BN + m] = c2;
        }
        s1[mi][r] = 0.f; s2[mi][r] = 0.f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < 2 * BN) {
      const int which = tid / BN, m = tid - which * BN;
      if (m < p.Cout) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[(w * 2 + which) * BN + m];
        atomicAdd(p.stats + (long)(blockIdx.x % p.nrep) * p.rep_stride + ((long)grp * 2 + which) * p.stats_ld + m, (double)tot);
      }
    }
  };

  if constexpr ((VAR & 2) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  // ---- prime: the weights of the first kernel row, then the halo of chunk 0 ----
  int cb, coh0, cow0, ch_half = 0, ct = t_begin;           // current chunk: image, tile origin, channel half, tile
  tile_of(ct, cb, coh0, cow0);
  halo_offsets(coh0, cow0);
#pragma unroll
  for (int r = 0; r < C::WR; ++r) w_dma(r, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < C::HR; ++r) halo_dma(r, chunk_src(cb, czt, 0), 0);
  int cz = czt;                                            // output slice of the current tile
  int wsel = 0;
  bool epi_counted = false;                                // the previous chunk ended with exactly NSTORE stores per lane

  // 3x3, <= 32 output channels: the resident weights (9 taps x 2 fragments) stay in REGISTERS — every tile uses the same ones, and with
  // two output-channel tiles per wave the weight fragments were a third of the kernel's LDS reads, which bound it (6 reads per 8 MFMAs)
  constexpr bool WREG = C::WRES && BN <= 32;
  u32x4 wreg[WREG ? C::TPS : 1][NT_CO];

  for (int c = 0; c < nchunks; ++c) {
    const int hsel = c & 1;
    const bool has_next = c + 1 < nchunks;
    // next chunk
    int nb = cb, noh0 = coh0, now0 = cow0, nh = ch_half + 1, nt = ct, nz = cz;
    if (nh == nhalf) {
      nh = 0; nt = ct + 1;
      if (has_next) { tile_of(nt, nb, noh0, now0); halo_offsets(noh0, now0); nz = czt; }
    }
    const int nsoff = chunk_src(nb, nz, nh);
    if (ch_half == 0) {
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* const hb = smem + hsel * HB + wave_row;

    band_static_for<0, C::NSTG>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // The weights of this stage (and, entering stage 0, the whole halo image) have landed once at most the operations
      // issued behind them are outstanding: the halo rounds of the previous stage (counted only where every wave issued
      // the same number), or the previous tile's epilogue stores (vmcnt retires in issue order).
      constexpr int r_lo = (j - 1) * C::HPS, r_hi = j * C::HPS < C::HR ? j * C::HPS : C::HR;
      constexpr bool counted = j >= 1 && r_hi > r_lo && r_hi * RPR <= C::HROWS;
      if (j == 0 && epi_counted) band_wait<NSTORE>();
      else if (counted && has_next) band_wait<counted ? r_hi - r_lo : 0>();
      else band_wait<0>();
      if (!(dbg & 16)) __builtin_amdgcn_s_barrier();

      const unsigned char* const wl = smem + 2 * HB + wsel * WB + a_base;
      const unsigned char* const hj = hb + j * C::RS * (IWp * 64);
      if constexpr (WREG) {
        if (c == 0) {                                       // (the weights landed with the first image: the wait above)
#pragma unroll
          for (int t = 0; t < C::TPS; ++t)
#pragma unroll
            for (int mi = 0; mi < NT_CO; ++mi) wreg[t][mi] = *reinterpret_cast<const u32x4*>(wl + t * (BN * 64) + mi * 1024);
        }
      }
      u32x4 af[2][NT_CO], bf[2][NT_PIX];
      if constexpr (DBG) if (dbg & 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) { if (q < NT_CO) af[i][q] = u32x4{0x3f803f80u, (unsigned)tid, 0x3f803f80u, 0x3f803f80u}; bf[i][q] = u32x4{0x3f803f80u, 0x3f803f80u, (unsigned)lane, 0x3f803f80u}; }
      }
      auto load = [&](int set, int t) {                     // tap t of the stage: kernel row t / K of it, column t % K
        if (dbg & 32) return;
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi) {
          if constexpr (WREG) af[set][mi] = wreg[t][mi];
          else af[set][mi] = *reinterpret_cast<const u32x4*>(wl + t * (BN * 64) + mi * 1024);
        }
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni)
          bf[set][ni] = *reinterpret_cast<const u32x4*>(hj + b_k[t % K] + ((t / K + (ni >> 1)) * IWp + (ni & 1) * 16) * 64);
      };
      // DMA slot s of the stage, issued between the fragment reads of tap s+1 and the MFMAs of tap s: first the weights
      // of the next stage, behind them this stage's share of the next chunk's halo
      auto dma_slot = [&](int sl) {
        if (sl < C::WRS) {
          if (dbg & 2) return;
          if (j + 1 < C::NSTG) w_dma(sl, j + 1, ch_half, wsel ^ 1);
          else if (has_next) w_dma(sl, 0, nh, wsel ^ 1);
        } else if (sl < C::WRS + C::HPS) {
          if (dbg & 1) return;
          if (has_next && (C::NSTG == 1 || j < C::NSTG - 1)) halo_dma(j * C::HPS + sl - C::WRS, nsoff, hsel ^ 1);
        }
      };
      load(0, 0);
#pragma unroll
      for (int kw = 0; kw < C::TPS; ++kw) {                 // (kw: tap index inside the stage)
        if constexpr ((VAR & 1) != 0) {
#pragma unroll
          for (int sl = kw * C::SPT; sl < (kw + 1) * C::SPT; ++sl) dma_slot(sl);
        }
        if (kw + 1 < C::TPS) load((kw + 1) & 1, kw + 1);
        if constexpr ((VAR & 1) == 0) {
#pragma unroll
          for (int sl = kw * C::SPT; sl < (kw + 1) * C::SPT; ++sl) dma_slot(sl);
        }
        if (!(dbg & 4)) {
#pragma unroll
          for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT_PIX; ++ni) Mma<T>::run(acc[mi][ni], af[kw & 1][mi], bf[kw & 1][ni]);
        }
        if constexpr ((VAR & 4) != 0) {                    // one fragment read per two MFMAs
#pragma unroll
          for (int g = 0; g < NT_CO + NT_PIX; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (NT_CO * NT_PIX) / (NT_CO + NT_PIX), 0);
          }
        }
      }
      if constexpr (!C::WRES) wsel ^= 1;
    });

    epi_counted = false;
    if (ch_half == nhalf - 1) {
      // ---- epilogue of tile ct: bias / activation / store / BatchNorm statistics of the stored values ----
      const int oimg = KD > 1 ? cb * p.Do + cz : cb;        // output image (slice)
      const int grp = oimg < p.bpg ? 0 : oimg / p.bpg;
      if (p.bias) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] += bv[mi][r];
      }
      if (p.act == 1) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 1.f / (1.f + __expf(-acc[mi][ni][r]));
      }
      T* const yb = (T*)p.y + (long)oimg * p.Ho * p.Wo * p.ldy;
      const T* const rb = (const T*)p.res + (long)oimg * p.Ho * p.Wo * p.ldres;
      const bool interior = coh0 + C::TH <= p.Ho && cow0 + C::TW <= p.Wo && BN <= p.Cout;   // workgroup-uniform
      if (interior) {
        // 16-byte stores: v_permlane16_swap trades the 4 channels a lane holds for pixel tile 2q+1 against the NEXT 4
        // channels (lane + 16) of pixel tile 2q, so that lane rows 0/2 own 8 consecutive channels of tile 2q and rows
        // 1/3 of tile 2q+1: half the store instructions of the 8-byte form, the cost that bounds the epilogue
        const long pix0 = (long)(coh0 + RPW * wave) * p.Wo + cow0 + (lg & 1) * 16 + l15;   // this lane's pixel of row q = 0 after the swap
        T* const d0 = yb + pix0 * p.ldy + 8 * (lg >> 1);
        const T* const r0 = rb + pix0 * p.ldres + 8 * (lg >> 1);
        // q outside, mi inside: the four 32-byte pieces of a pixel's 128-byte line leave in four CONSECUTIVE store
        // instructions and merge in L2 (with mi outside, WRITE_SIZE read 1.8x the output: partial lines written back)
#pragma unroll
        for (int q = 0; q < NT_PIX / 2; ++q) {               // pixel tiles 2q, 2q+1: the two 16-pixel halves of one row
          T* dst = d0 + (long)q * p.Wo * p.ldy;
          const T* rsrc = r0 + (long)q * p.Wo * p.ldres;
          u32x4 ov[NT_CO];
#pragma unroll
          for (int mi = 0; mi < NT_CO; ++mi) {
            const f32x4 v0 = acc[mi][2 * q], v1 = acc[mi][2 * q + 1];
            const u32x2 o0 = u32x2{pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3])};
            const u32x2 o1 = u32x2{pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
            if (p.stats) {
              const f32x4 w0 = f32x4{bflo(o0[0]), bfhi(o0[0]), bflo(o0[1]), bfhi(o0[1])};
              const f32x4 w1 = f32x4{bflo(o1[0]), bfhi(o1[0]), bflo(o1[1]), bfhi(o1[1])};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                s1[mi][r] += w0[r] + w1[r];
                s2[mi][r] = fmaf(w1[r], w1[r], fmaf(w0[r], w0[r], s2[mi][r]));
              }
            }
            u32x4 o;
            if (p.res) {                                     // uniform: y = result + res, summed in f32 and rounded once
              float f[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // (scalars first: __builtin_bit_cast on a vector ELEMENT reads element 0 under hipcc 7.2, see sdhip_common.h)
                const float a0 = v0[r], a1 = v1[r];
                const u32x2 e = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                const unsigned e0 = e[0], e1 = e[1];
                f[r] = __uint_as_float(e0); f[4 + r] = __uint_as_float(e1);
              }
              const u32x4 old = *reinterpret_cast<const u32x4*>(rsrc + mi * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = pack2bf(f[2 * e] + bflo(old[e]), f[2 * e + 1] + bfhi(old[e]));
            } else {
              const u32x2 e0 = __builtin_amdgcn_permlane16_swap(o0[0], o1[0], false, false);
              const u32x2 e1 = __builtin_amdgcn_permlane16_swap(o0[1], o1[1], false, false);
              o = u32x4{e0[0], e1[0], e0[1], e1[1]};
            }
            ov[mi] = o;
          }
          if (!(dbg & 8)) {                                  // the line's pieces back to back
#pragma unroll
            for (int mi = 0; mi < NT_CO; ++mi) *reinterpret_cast<u32x4*>(dst + mi * 16) = ov[mi];
          }
        }
        epi_counted = !(dbg & 8) && !p.res;              // (the addend's loads sit among the stores)
      } else {
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) {
          const int oh = coh0 + RPW * wave + (ni >> 1), ow = cow0 + (ni & 1) * 16 + l15;
          const bool valid = oh < p.Ho && ow < p.Wo;
          T* dst = yb + ((long)oh * p.Wo + ow) * p.ldy;
          const T* rsrc = rb + ((long)oh * p.Wo + ow) * p.ldres;
#pragma unroll
          for (int mi = 0; mi < NT_CO; ++mi) {
            const int co = mi * 16 + 4 * lg;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r];
            if (valid && co + 3 < p.Cout) {
              if (p.res) {
                const u32x2 old = *reinterpret_cast<const u32x2*>(rsrc + co);
                v[0] += bflo(old[0]); v[1] += bfhi(old[0]); v[2] += bflo(old[1]); v[3] += bfhi(old[1]);
              }
              const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
              *reinterpret_cast<u32x2*>(dst + co) = o;
              v[0] = bflo(o[0]); v[1] = bfhi(o[0]); v[2] = bflo(o[1]); v[3] = bfhi(o[1]);
            } else if (valid && co < p.Cout) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (co + r < p.Cout) {
                  if (p.res) v[r] += Elem<T>::ld(rsrc + co + r);
                  Elem<T>::st(dst + co + r, v[r]); v[r] = Elem<T>::rnd(v[r]);
                }
                else v[r] = 0.f;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = 0.f;
            }
            if (p.stats) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] = fmaf(v[r], v[r], s2[mi][r]); }
            }
          }
        }
      }
      if (p.stats) {
        const int nimg = KD > 1 ? nb * p.Do + nz : nb;
        const int ngrp = nimg < p.bpg ? 0 : nimg / p.bpg;
        if (!has_next || ngrp != grp) { flush_stats(grp); epi_counted = false; }   // workgroup-uniform
      }
    }
    cb = nb; coh0 = noh0; cow0 = now0; ch_half = nh; ct = nt; cz = nz;
  }
}

inline bool band_ok(int ntiles) { return ntiles >= 192 && ntiles < 65536; }

template <int K, int BN, int VAR, int NW = 8, int KD = 1>
int launch_band_bn(const BandArgs& a, int grid, hipStream_t s) {
  auto kern = conv_band_kernel<K, BN, VAR, NW, KD>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv_band: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  const size_t lds = BandCfg<K, BN, NW, KD>::LDS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

template <int K, int KD = 1>
int launch_band(BandArgs& a, hipStream_t s) {
  const int tiles_h = sdhip_cdiv(a.Ho, 16);
  a.tiles_w = sdhip_cdiv(a.Wo, 32);
  a.tiles_hw = tiles_h * a.tiles_w;
  a.ntiles = a.tiles_hw * a.B * (KD > 1 ? a.Do : 1);
  a.magic_do = (KD > 1 && a.Do > 1) ? (unsigned)(0x100000000ULL / (unsigned)a.Do) + 1u : 0u;
  a.magic_hw = a.tiles_hw > 1 ? (unsigned)(0x100000000ULL / (unsigned)a.tiles_hw) + 1u : 0u;
  a.magic_tw = a.tiles_w > 1 ? (unsigned)(0x100000000ULL / (unsigned)a.tiles_w) + 1u : 0u;
  a.tpw = sdhip_cdiv(a.ntiles, 256);
  a.dbg = sdhip_diag().tune_band_dbg;
  const int grid = sdhip_cdiv(a.ntiles, a.tpw);
  // VAR 5 = DMA slot ahead of the fragment reads + one read per two MFMAs (sched_group_barrier): 207 -> 201 us against the
  // other orders; 4 waves (one per SIMD, 64 x 128 register tiles) measured 265 us — the partner wave's cover is worth more
  // than the quarter of LDS reads saved.  VAR 128: diagnostic build that honours SDHIP_TUNE_BAND_DBG.
  a.dbg &= 0xff;
  if constexpr (K == 5) {
    if (a.Mpad > 32) return a.dbg ? launch_band_bn<K, 64, 128 + 5>(a, grid, s) : launch_band_bn<K, 64, 5>(a, grid, s);
  } else if constexpr (KD > 1) {
    return launch_band_bn<K, 32, 5, 8, KD>(a, grid, s);      // (host: <= 32 channels on both sides)
  } else {
    if (a.Mpad > 32) return launch_band_bn<K, 64, 5>(a, grid, s);
  }
  return launch_band_bn<K, 32, 5>(a, grid, s);
}

}  // namespace
