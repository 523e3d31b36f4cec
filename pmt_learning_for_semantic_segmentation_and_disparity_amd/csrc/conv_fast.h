// Fast path of the direct convolution (forward / data-gradient form) for gfx950.
//
// Same algorithm as conv_fwd.hip's general kernel (halo tile of one input-channel chunk staged once in LDS, all
// kh*kw taps walk over it, tap-group weights double-buffered next to it), restricted to what the hot layers need —
// 16-byte-aligned pixels (ld % 8 == 0 for bf16), no per-tap mode — so that the inner loop carries no per-element
// conditions: every LDS fragment read is unconditional, the tap loop is a two-register-set software pipeline
// (the reads of step u+1 are in flight behind the MFMAs of step u) and all tile geometry lives in SGPRs.
//
// LDS images (rows of RB = 64*KS bytes; KS = k-steps per row: 2 = 128-byte rows, 1 = 64-byte rows for Cin <= 32 bf16):
//   halo  : row = ih * IWp + iw with the row pitch IWp rounded up to a multiple of 8, so that the swizzle key of a
//           pixel does not depend on its tile row, on the tap's kernel row or on the 16-pixel column block
//   weight: row = tap_in_group * BN + m
// 16-byte chunk c of row r sits at chunk slot  c ^ (r & 6)        (KS = 2)
//                                              c ^ ((r >> 1) & 2) (KS = 1)
// which is conflict-free for ds_read_b128 under the gfx950 lane grouping {0-3,12-15,20-27},{4-11,16-19,28-31},...
// for EVERY alignment of the 16 consecutive rows a fragment read touches (found by exhaustive search; the plain
// (r >> 1) & 7 key is 2-way conflicted for half of the tap shifts).
#pragma once
#include "conv_common.h"

namespace {

struct FastArgs {
  const void* x; const void* wp; void* y;
  const float* bias; const float* in_scale; const float* in_shift; double* stats;
  int B, H, W, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l;
  int D, Do, kd, sd, pad_d;
  int Cin, ldx, Cout, Mpad, ldy;
  int in_relu, bpg, act, accumulate;   // bpg: images per statistics group
  unsigned magic_iwp, magic_tw;        // div_magic(halo row pitch), div_magic(tiles per output row) of the chosen tile
  int tiles_w, ntg;                    // tiles per output row, tap groups per chunk
  int tg, stats_ld, nrep, tail, dma;   // tail: Cin % (elements per 16 bytes) != 0 -> mask the last chunk; dma: halo by LDS-DMA
  long rep_stride;
  // interleaved output (one sub-pixel phase of a stride-2 transposed convolution): output voxel (dz, oh, ow) of image b is
  // stored at (dz*omul + ooz, oh*omul + ooy, ow*omul + oox) of a volume omul times as large per axis.  omul = 1: plain.
  int omul, ooz, ooy, oox;
  // BatchNorm-backward sums in the epilogue (sdhip_conv2d_fwd_bnbwd): this launch is the data gradient that produces
  // g = dL/d relu(bn(u)); with u = bx (same geometry as y, pixel stride ldbx) and z = u*bsc + bsh the statistics slots
  // receive sum(gm * u) and sum(gm), gm = g where z > 0 else 0 — the reductions of sdhip_affine_act_bwd — instead of
  // sum(y), sum(y^2).  nullptr: plain statistics.
  const void* bx; const float* bsc; const float* bsh; int ldbx;
  const void* res; int ldres;   // BX launches only: y = result + res (a second gradient contribution, same geometry as y)
  // bx_mode 1 ("apply"): y = res + gm * bsc — the whole first phase of sdhip_affine_act_bwd with accumulate: the masked,
  // scaled gradient is ADDED to res (which may be y itself) and the two sums go, as f32, to ((float*)stats)[which][rep][group][Cout]
  // (the dscale / dshift replicas that sdhip_affine_act_bwd fills); bx_groups = number of statistics groups
  int bx_mode, bx_groups;
  // Consumer-side BatchNorm finalize (sdhip_conv2d_fwd_bnpro; register-staging launches only): the input prologue's
  // scale / shift are derived HERE from the batch statistics the producing convolution's epilogue wrote — no per-channel
  // kernel between the two convolutions.  pin_stats: f64 [pin_nrep][groups][2][pin_ld]; workgroup (0,0,0) also writes
  // scale / shift / mean / invstd ([groups][Cin], for the backward pass) and updates the running statistics.
  const double* pin_stats; int pin_ld, pin_nrep, pin_groups, pin_off, pin_cs;
  const float* pin_gamma; const float* pin_beta;
  float* pin_scale; float* pin_shift; float* pin_mean; float* pin_invstd; float* pin_rmean; float* pin_rvar;
  float pin_eps, pin_momentum; double pin_count;
  // ... of a DenseNet slab: channels [pend_c0, pend_c0 + pend_n) are the previous layer's output, whose statistics still sit in
  // the replicas pend_stats (f64 [pend_nrep][groups][2][pend_ld], channel index c - pend_c0); the writer workgroup folds them
  // into the slab statistics pin_fold (= pin_stats, writable) on the way.  pend_n = 0: none.
  const double* pend_stats; double* pin_fold; int pend_ld, pend_nrep, pend_c0, pend_n;
};

constexpr int kPinMaxC = 1024;   // input channels the consumer-side finalize table holds

// source of every padding / dead lane of an LDS-DMA load
__device__ __attribute__((aligned(16))) unsigned int sdhip_zero16[4] = {0u, 0u, 0u, 0u};

// one global_load_lds_dwordx4: the wave's 64 lanes fetch 16 bytes each from their own address and the hardware writes
// them to LDS byte address lds_wave_base + lane*16 (no VGPR destination; completion is tracked by vmcnt).
// Issued as inline asm on purpose: when hipcc sees an LDS-DMA in flight it puts `s_waitcnt vmcnt(0)` in front of every
// following ds_read it cannot prove disjoint (all of them, with dynamic LDS offsets), which would serialise the
// prefetch of stage s+1 with the fragment reads of stage s.  The kernel waits for its DMA explicitly before the
// barrier that publishes a stage instead.
// M0 (the LDS base of the DMA) is declared clobbered instead of being saved / restored around every instruction: it is
// not an allocatable register — the backend (re)initialises it immediately in front of each of its own uses and, with
// the clobber declared, knows this asm redefines it.  Three scalar moves less per DMA instruction = 3 % on the 5x5 kernel.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16(const void* g, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_wave_base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

// sum over the 16 lanes of a DPP row (the 16 pixels of one MFMA output tile); every lane gets the total.
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: four v_add_f32_dpp, no LDS traffic.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return dpp_add<0x140>(v);
}

template <int KS> struct LdsRow;
template <> struct LdsRow<2> {
  static __device__ __forceinline__ int off(int row, int c) { return row * 128 + ((c ^ (row & 6)) << 4); }
};
template <> struct LdsRow<1> {
  static __device__ __forceinline__ int off(int row, int c) { return row * 64 + ((c ^ ((row >> 1) & 2)) << 4); }
};

// DMA: the halo tile goes global -> LDS by LDS-DMA (no BatchNorm prologue, no odd channel tail); otherwise through
// registers with the prologue applied on the way.  Two instantiations, so that the DMA variant contains no ordinary
// global load at all inside its stage loop (hipcc would wait for it — and with it for the DMA — in the MFMA loop).
// BX: the BatchNorm-backward-sums epilogue (FastArgs::bx) — an instantiation of its own, because its extra live values
// push the 8x32x64 form from 244 to 264 VGPRs, i.e. from two workgroups per CU to one, for EVERY launch of that form.
template <typename T, int TH, int TW, int BN, int KS, bool DMA, bool BX = false>
__global__ __launch_bounds__(256) void conv_fast_kernel(const FastArgs p) {
  constexpr int V = Chunk<T>::N;          // elements per 16 bytes
  constexpr int CK = 8 * V;               // channels per packed weight row (always 128 bytes)
  constexpr int CKS = 4 * KS * V;         // channels per staged LDS row
  constexpr int RB = 64 * KS;             // LDS row bytes
  constexpr int SH = KS + 1;              // log2(16-byte chunks per LDS row)
  constexpr int CH = 4 * KS;
  constexpr int TWT = TW / 16, NPT = TH * TWT, NT_PIX = NPT / 4, NT_CO = BN / 16;
  constexpr bool PF = (TH * TW <= 64);    // small tiles: halo register-prefetched and double-buffered
  constexpr int HPF = (DMA && KS == 1) ? 6 : 5;   // load rounds of the prefetched (small-tile) halo (6: the 9 x 40 64-byte rows of a stride-2 3x3 tile; DMA kernels hold no data registers for it)
  static_assert(NPT % 4 == 0 && NT_PIX >= 1, "every wave needs at least one 16-pixel MFMA tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int s = p.stride, d = p.dil;
  const int tiles_w = p.tiles_w;
  const int ty = fast_div(blockIdx.x, tiles_w, p.magic_tw);   // < 2^16 tiles per image (host-checked)
  const int oh0 = ty * TH, ow0 = (blockIdx.x - ty * tiles_w) * TW;
  const int n0 = blockIdx.y * BN;
  const int b = p.Do == 1 ? (int)blockIdx.z : (int)blockIdx.z / p.Do, dz = blockIdx.z - b * p.Do;
  const int grp = b < p.bpg ? 0 : b / p.bpg;
  const int IH = (TH - 1) * s + (p.kh - 1) * d + 1, IW = (TW - 1) * s + (p.kw - 1) * d + 1;
  const int IWp = (IW + 7) & ~7;
  const int ih0 = oh0 * s - p.pad_t, iw0 = ow0 * s - p.pad_l;
  const int Tn = p.kh * p.kw;
  const int nq = KS == 2 ? (p.Cin + CK - 1) / CK : 1;
  const int nqq = p.kd * nq;
  const int halo_bytes = ((IH * IWp + (256 >> SH) - 1) & ~((256 >> SH) - 1)) * RB;   // whole load rounds (4 KiB)
  const int wbuf_bytes = ((p.tg * BN + (256 >> SH) - 1) & ~((256 >> SH) - 1)) * RB;
  unsigned char* const halo0 = smem;
  unsigned char* const wl0 = smem + halo_bytes * ((PF && nqq > 1) ? 2 : 1);   // a single chunk needs no second halo buffer

  f32x4 acc[NT_CO][NT_PIX];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* const xb0 = (const T*)p.x + (long)b * p.D * p.H * p.W * p.ldx;
  const T* const wpk = (const T*)p.wp;
  const int ntg = p.ntg;

  // ---- fragment addressing: everything per lane is computed once ----
  // wave's pixel tiles: pt = wave*NT_PIX + ni -> tile row pt / TWT, column block pt % TWT.  Row pitch IWp and the
  // 16-pixel column step are multiples of 8 rows, so all of a lane's pixel tiles share one swizzle key per tap.
  const int pt0 = wave * NT_PIX;
  const int prow0 = ((pt0 / TWT) * s) * IWp + ((pt0 % TWT) * 16 + l15) * s;   // halo row of pixel tile 0, tap (0,0)
  int pdelta[NT_PIX];   // wave-uniform row distance of pixel tile ni from pixel tile 0 (multiple of 8)
#pragma unroll
  for (int ni = 0; ni < NT_PIX; ++ni) {
    const int pt = pt0 + ni;
    pdelta[ni] = (((pt / TWT) - (pt0 / TWT)) * s * IWp + ((pt % TWT) - (pt0 % TWT)) * 16 * s) * RB;
  }
  const int lg4 = lg << 4;
  const int a_base = LdsRow<KS>::off(l15, lg);   // weight row l15 (+ mi*16 rows, + tap*BN rows: key unchanged)

  // ---- staging ----
  // Lane <-> LDS mapping is LINEAR: in load round j lane tid fills bytes [(j*256 + tid)*16, +16) of the image, i.e.
  // physical chunk slot (tid & (CH-1)) of row j*RPR + (tid >> SH).  The swizzle lives on the SOURCE side: the lane
  // fetches logical chunk slot ^ key(row); RPR is a multiple of 8 rows, so the key (and the lane's channel offset)
  // is the same in every round.  That is the layout `global_load_lds` (LDS-DMA, 1 KiB per wave-instruction, no
  // VGPR round trip) writes, and the register path (BatchNorm prologue / odd channel tails) uses it too.
  constexpr int RPR = 256 >> SH;            // LDS rows per round
  const int rsub = tid >> SH;
  const int c_l = (tid & (CH - 1)) ^ (KS == 2 ? (rsub & 6) : ((rsub >> 1) & 2));   // logical chunk this lane fetches
  const int tid16 = tid * 16;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);   // LDS byte address of this wave's KiB in round 0 of smem
  const unsigned magic_iwp = p.magic_iwp;   // host-computed: a 64-bit division per wave is ~100 instructions of kernel start-up
  constexpr bool dma = DMA;
  // (scale, shift) table of the consumer-side finalize: 2 x pin_cs floats behind the kernel's other dynamic LDS (pin_off)
  float* const ptab_sc = reinterpret_cast<float*>(smem + (DMA ? 0 : p.pin_off));
  float* const ptab_sh = ptab_sc + (DMA ? 0 : p.pin_cs);
  const bool has_pro = p.in_scale != nullptr || (!DMA && p.pin_stats != nullptr);
  const int h_rows = IH * IWp;
  const int h_rounds = (h_rows + RPR - 1) / RPR;
  int h_src[HPF];     // small tiles: element offset of the lane's chunk inside one depth slice per round, or -1 (padding)
  if constexpr (PF) {
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      const int row = rsub + j * RPR;
      const int ih = fast_div(row, IWp, magic_iwp), iw = row - ih * IWp;
      const int gh = ih0 + ih, gw = iw0 + iw;
      h_src[j] = (row < h_rows && iw < IW && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) ? (gh * p.W + gw) * p.ldx + c_l * V : -1;
    }
  }
  const int mvalid = min(BN, p.Mpad - n0);

  u32x4 rh[HPF];
  // chunk qq = (depth tap, channel chunk); plain 2-D convolutions (kd == 1) skip the scalar divisions
  const bool flat = p.kd == 1;
  auto chunk_of = [&](int qq) { return flat ? qq : qq % nq; };
  auto slice_of = [&](int qq) { return flat ? dz * p.sd - p.pad_d : dz * p.sd + qq / nq - p.pad_d; };

  auto mask_tail = [&](u32x4 raw, int ch0) -> u32x4 {   // zero the elements at channel >= Cin (rare: odd channel counts)
    const int nv = p.Cin - ch0;
    if (nv < V) {
      float f[V];
      Chunk<T>::unpack(raw, f);
#pragma unroll
      for (int e = 0; e < V; ++e) f[e] = e < nv ? f[e] : 0.f;
      raw = Chunk<T>::pack(f);
    }
    return raw;
  };
  // BatchNorm affine (+ReLU) of the producer, fused on load.  A lane's channel offset inside a chunk is fixed, so its
  // 2*V scale/shift values are fetched once per chunk (four 16-byte loads), not once per pixel.
  float psc[V], psf[V];
  auto prologue_load = [&](int ch0) {
    if (ch0 < p.Cin) {   // Cin % V == 0 whenever a prologue is fused (host: the tail path never carries one)
      if constexpr (!DMA) {
        if (p.pin_stats) {   // consumer-side finalize: the table built at kernel start
#pragma unroll
          for (int e = 0; e < V; ++e) { psc[e] = ptab_sc[ch0 + e]; psf[e] = ptab_sh[ch0 + e]; }
          return;
        }
      }
      const float* sc = p.in_scale + grp * p.Cin + ch0;
      const float* sf = p.in_shift + grp * p.Cin + ch0;
#pragma unroll
      for (int e = 0; e < V; e += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sc + e), b4 = *reinterpret_cast<const f32x4*>(sf + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) { psc[e + i] = a4[i]; psf[e + i] = b4[i]; }
      }
    }
  };
  auto prologue = [&](u32x4 raw) -> u32x4 {
    float f[V];
    Chunk<T>::unpack(raw, f);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float v = fmaf(f[e], psc[e], psf[e]);
      f[e] = p.in_relu ? fmaxf(v, 0.f) : v;
    }
    return Chunk<T>::pack(f);
  };

  // small tiles: issue the whole halo tile of chunk qq (DMA: straight into `dst`; register path: into rh[])
  auto halo_issue = [&](int qq, unsigned char* dst) {
    const int q = chunk_of(qq);
    const int din = slice_of(qq);
    const bool ok = din >= 0 && din < p.D && q * CKS + c_l * V < p.Cin;
    const T* xb = xb0 + (long)din * p.H * p.W * p.ldx + q * CKS;
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      if (j < h_rounds) {
        const T* src = (ok && h_src[j] >= 0) ? xb + h_src[j] : (const T*)sdhip_zero16;
        if constexpr (dma) glds16(src, (unsigned)(dst - smem) + j * 4096 + wave_lds);
        else rh[j] = *reinterpret_cast<const u32x4*>(src);
      }
    }
  };
  auto halo_commit = [&](int qq, unsigned char* dst) {   // register path only
    const int ch0 = chunk_of(qq) * CKS + c_l * V;
    const int din = slice_of(qq);
    const bool ok = din >= 0 && din < p.D && ch0 < p.Cin;
    if (has_pro) prologue_load(ch0);
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      if (j < h_rounds) {
        u32x4 raw = rh[j];
        if (ok && h_src[j] >= 0) {
          if (p.tail) raw = mask_tail(raw, ch0);
          if (has_pro) raw = prologue(raw);
        }
        *reinterpret_cast<u32x4*>(dst + j * 4096 + tid16) = raw;
      }
    }
  };
  // large tiles: stage the whole halo tile now (DMA: every round in flight at once; register path: 4 rounds at a time)
  auto halo_sync_stage = [&](int qq, unsigned char* dst) {
    const int ch0 = chunk_of(qq) * CKS + c_l * V;
    const int din = slice_of(qq);
    const bool ok = din >= 0 && din < p.D && ch0 < p.Cin;
    const T* xb = xb0 + (long)din * p.H * p.W * p.ldx + ch0;
    auto src_of = [&](int j) -> const T* {
      const int row = rsub + j * RPR;
      const int ih = fast_div(row, IWp, magic_iwp), iw = row - ih * IWp;
      const int gh = ih0 + ih, gw = iw0 + iw;
      const bool in = ok && iw < IW && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
      return in ? xb + (gh * p.W + gw) * p.ldx : (const T*)sdhip_zero16;
    };
    if constexpr (dma) {
      for (int j = 0; j < h_rounds; ++j) glds16(src_of(j), (unsigned)(dst - smem) + j * 4096 + wave_lds);
    } else {
      if (has_pro) prologue_load(ch0);
      for (int j0 = 0; j0 < h_rounds; j0 += 4) {
        u32x4 raw[4];
        bool in[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j0 + j < h_rounds) {   // uniform; never leave an unconsumed load pending (it would be waited for in the MFMA loop)
            const T* src = src_of(j0 + j);
            in[j] = src != (const T*)sdhip_zero16;
            raw[j] = *reinterpret_cast<const u32x4*>(src);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j0 + j < h_rounds) {
            if (in[j] && p.tail) raw[j] = mask_tail(raw[j], ch0);
            if (in[j] && has_pro) raw[j] = prologue(raw[j]);
            *reinterpret_cast<u32x4*>(dst + (j0 + j) * 4096 + tid16) = raw[j];
          }
        }
      }
    }
  };
  // weights of stage (chunk qq, taps t0..): always LDS-DMA, issued right after the barrier that retired the buffer.
  // Rows of output channels past Mpad (last block) and taps past the kernel are never used by a stored result: they
  // re-read row 0 of the stage instead of being zero-filled.
  constexpr int RPT = BN >= RPR ? BN / RPR : 1;   // load rounds per tap
  constexpr int TPR = BN >= RPR ? 1 : RPR / BN;   // taps per load round (narrow blocks)
  auto w_issue = [&](int qq, int t0, unsigned char* dst) {   // packed weights are [kd][nq][T][Mpad][CK]
    const int nt = min(p.tg, Tn - t0);
    const T* base = wpk + ((long)(qq * Tn + t0) * p.Mpad + n0) * CK + c_l * V;
    unsigned lds = (unsigned)(dst - smem) + wave_lds;
    if constexpr (BN >= RPR) {
      for (int tl = 0; tl < nt; ++tl) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const int m = min(r * RPR + rsub, mvalid - 1);
          glds16(base + (tl * p.Mpad + m) * CK, lds);
          lds += 4096;
        }
      }
    } else {
      const int m = min(rsub % BN, mvalid - 1);
      for (int t = 0; t < nt; t += TPR) {
        const int tl = min(t + rsub / BN, nt - 1);
        glds16(base + (tl * p.Mpad + m) * CK, lds);
        lds += 4096;
      }
    }
  };

  // ---- the MFMA steps of one stage: (tap, k-step) pairs, two fragment register sets ----
  // tap walk state (wave-uniform), carried across the stages of a chunk: halo-row offset of the current tap
  int kwi = 0, tap_off = 0;
  const int tap_row_step = d * IWp - p.kw * d;
  auto compute = [&](int t0, const unsigned char* halo, const unsigned char* wl) {
    const int nt = min(p.tg, Tn - t0);
    const int n = nt * KS;
    u32x4 af[2][NT_CO], bf[2][NT_PIX];
    int b_addr = 0, a_addr = 0;
    auto next_tap = [&](int tl) {   // fragment byte addresses of tap tl of the group, k-step 0, pixel tile 0 / cout tile 0
      const int row = prow0 + tap_off;
      if constexpr (KS == 2) b_addr = row * RB + (lg4 ^ ((row & 6) << 4));
      else b_addr = row * RB + (lg4 ^ (((row >> 1) & 2) << 4));
      a_addr = a_base + tl * (BN * RB);
      tap_off += d;                                            // next tap of the kernel row ...
      if (++kwi == p.kw) { kwi = 0; tap_off += tap_row_step; }  // ... or first tap of the next kernel row
    };
    auto load = [&](int set, int ks) {
      const unsigned char* ab = wl + (a_addr ^ (ks << 6));
      const unsigned char* bb = halo + (b_addr ^ (ks << 6));
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi) af[set][mi] = *reinterpret_cast<const u32x4*>(ab + mi * (16 * RB));
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) bf[set][ni] = *reinterpret_cast<const u32x4*>(bb + pdelta[ni]);
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) Mma<T>::run(acc[mi][ni], af[set][mi], bf[set][ni]);
    };
    if constexpr (KS == 2) {
      next_tap(0);
      load(0, 0);
      for (int tl = 0; tl < nt; ++tl) {
        load(1, 1);
        mma(0);
        if (tl + 1 < nt) { next_tap(tl + 1); load(0, 0); }
        mma(1);
      }
    } else {
      next_tap(0);
      load(0, 0);
      for (int u = 0; u < n; u += 2) {
        if (u + 1 < n) { next_tap(u + 1); load(1, 0); }
        mma(0);
        if (u + 2 < n) { next_tap(u + 2); load(0, 0); }
        if (u + 1 < n) mma(1);
      }
    }
  };

  // the first weights (LDS-DMA) and, for small tiles, the raw halo loads go out BEFORE the consumer-side finalize below: they
  // do not depend on it and cover its round trips to the statistics
  w_issue(0, 0, wl0);
  if constexpr (PF) halo_issue(0, halo0);
  if constexpr (!DMA) {
    if (p.pin_stats) {   // uniform: scale / shift of this workgroup's statistics group, arithmetic of bn_finalize_kernel
      const int G = p.pin_groups;
      auto sums_of = [&](int g, int c, double& a1, double& a2) {   // sum and sum of squares of channel c in group g
        a1 = 0.; a2 = 0.;
        const int cp = c - p.pend_c0;
        if (cp >= 0 && cp < p.pend_n) {
          for (int r = 0; r < p.pend_nrep; ++r) {
            const double* Sr = p.pend_stats + (long)r * G * 2 * p.pend_ld;
            a1 += Sr[((long)g * 2 + 0) * p.pend_ld + cp];
            a2 += Sr[((long)g * 2 + 1) * p.pend_ld + cp];
          }
        } else {
          for (int r = 0; r < p.pin_nrep; ++r) {
            const double* Sr = p.pin_stats + (long)r * G * 2 * p.pin_ld;
            a1 += Sr[((long)g * 2 + 0) * p.pin_ld + c];
            a2 += Sr[((long)g * 2 + 1) * p.pin_ld + c];
          }
        }
      };
      for (int c = tid; c < p.Cin; c += 256) {
        double a1, a2;
        sums_of(grp, c, a1, a2);
        const double mu = a1 / p.pin_count;
        double var = a2 / p.pin_count - mu * mu;
        if (var < 0.) var = 0.;
        const float inv = (float)(1.0 / sqrt(var + (double)p.pin_eps));
        const float scv = (p.pin_gamma ? p.pin_gamma[c] : 1.f) * inv;
        ptab_sc[c] = scv;
        ptab_sh[c] = (float)((double)(p.pin_beta ? p.pin_beta[c] : 0.f) - mu * (double)scv);
      }
      if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {   // the one writer: backward-pass vectors, running statistics, fold
        for (int c = tid; c < p.Cin; c += 256) {
          const float gm = p.pin_gamma ? p.pin_gamma[c] : 1.f, bt = p.pin_beta ? p.pin_beta[c] : 0.f;
          float rm = p.pin_rmean ? p.pin_rmean[c] : 0.f, rv = p.pin_rvar ? p.pin_rvar[c] : 0.f;
          const bool pend = c >= p.pend_c0 && c < p.pend_c0 + p.pend_n;
          for (int g = 0; g < G; ++g) {
            double a1, a2;
            sums_of(g, c, a1, a2);
            if (pend) {   // nobody reads these slab entries during this launch (every workgroup takes them from the replicas)
              p.pin_fold[((long)g * 2 + 0) * p.pin_ld + c] = a1;
              p.pin_fold[((long)g * 2 + 1) * p.pin_ld + c] = a2;
            }
            const double mu = a1 / p.pin_count;
            double var = a2 / p.pin_count - mu * mu;
            if (var < 0.) var = 0.;
            const float inv = (float)(1.0 / sqrt(var + (double)p.pin_eps));
            const float scv = gm * inv;
            p.pin_scale[g * p.Cin + c] = scv;
            p.pin_shift[g * p.Cin + c] = (float)((double)bt - mu * (double)scv);
            p.pin_mean[g * p.Cin + c] = (float)mu;
            p.pin_invstd[g * p.Cin + c] = inv;
            const double unb = p.pin_count > 1. ? var * p.pin_count / (p.pin_count - 1.) : var;
            rm = (1.f - p.pin_momentum) * rm + p.pin_momentum * (float)mu;
            rv = (1.f - p.pin_momentum) * rv + p.pin_momentum * (float)unb;
          }
          if (p.pin_rmean) { p.pin_rmean[c] = rm; p.pin_rvar[c] = rv; }
        }
      }
      __syncthreads();
    }
  }
  // ---- pipeline over stages (depth tap, channel chunk, tap group) ----
  // The loads of stage st+1 (its tap-group weights and, at a chunk boundary of a small tile, its halo tile) are issued
  // right after the barrier of stage st and land (DMA) or wait in registers behind the MFMAs of stage st.
  if constexpr (!PF) halo_sync_stage(0, halo0);
  int q = 0, tgi = 0;
  const int S = nqq * ntg;
  for (int st = 0; st < S; ++st) {
    const int t0 = tgi * p.tg;
    unsigned char* halo = halo0 + ((PF && (q & 1)) ? halo_bytes : 0);
    unsigned char* wl = wl0 + (st & 1) * wbuf_bytes;
    if constexpr (PF && !dma) { if (tgi == 0) halo_commit(q, halo); }
    // the DMA (inline asm, invisible to the compiler's wait-count pass) of this stage must have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // stage st is visible (DMA drained); every wave has finished stage st-1
    int qn = q, tgn = tgi + 1;
    if (tgn == ntg) { tgn = 0; ++qn; }
    const bool more = st + 1 < S;
    if (more) {
      if (PF && tgn == 0) halo_issue(qn, halo0 + ((qn & 1) ? halo_bytes : 0));
      w_issue(qn, tgn * p.tg, wl0 + ((st + 1) & 1) * wbuf_bytes);
    }
    compute(t0, halo, wl);
    if (!PF && more && tgn == 0) {        // chunk boundary with a single (large) halo buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      halo_sync_stage(qn, halo0);
    }
    if (tgn == 0) { kwi = 0; tap_off = 0; }   // next chunk starts at tap (0,0) again
    q = qn; tgi = tgn;
  }

  // ---- epilogue: bias / activation / accumulate / store / BN statistics ----
  // Everything wave-uniform (bias? activation? interior tile?) is decided once, outside the per-tile code, so the
  // common case — interior tile, plain store — is straight-line: 4 VALU + one 8/16-byte store per 16x16 output tile.
  if (p.bias) {
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = n0 + mi * 16 + 4 * lg + r;
        bv[r] = co < p.Cout ? p.bias[co] : 0.f;
      }
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] += bv[r];
    }
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

  float s1[NT_CO][4], s2[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }

  const int om = p.omul;
  const int WoD = p.Wo * om;                                                   // row pitch (pixels) of the stored image
  const long zimg = om == 1 ? (long)blockIdx.z : ((long)b * p.Do + dz) * om + p.ooz;
  T* const yb = (T*)p.y + zimg * (p.Ho * om) * WoD * p.ldy + ((long)p.ooy * WoD + p.oox) * p.ldy;
  // (accumulating launches take the general path: its per-tile branches keep the old values' loads from being hoisted
  //  together, which would cost ~100 VGPRs — a wave of occupancy — in every launch of this kernel)
  const bool interior = oh0 + TH <= p.Ho && ow0 + TW <= p.Wo && n0 + BN <= p.Cout && !p.accumulate;   // wave-uniform
  if constexpr (BX) {   // host: bx / stats set, bf16, Cout % 4 == 0, omul == 1, no accumulate / bias / activation
    {
    const T* const ub = (const T*)p.bx + zimg * p.Ho * p.Wo * p.ldbx;
    const T* const rb = (const T*)p.res + zimg * p.Ho * p.Wo * p.ldres;
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
      const int co = n0 + mi * 16 + 4 * lg;
      const bool cok = co < p.Cout;
      f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, sh = sc;
      if (cok) {
        sc = *reinterpret_cast<const f32x4*>(p.bsc + (long)grp * p.Cout + co);
        sh = *reinterpret_cast<const f32x4*>(p.bsh + (long)grp * p.Cout + co);
      }
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) {
        const int pt = pt0 + ni;
        const int oh = oh0 + pt / TWT, ow = ow0 + (pt % TWT) * 16 + l15;
        if (cok && oh < p.Ho && ow < p.Wo) {
          const long pix = (long)oh * p.Wo + ow;
          const u32x2 uu = *reinterpret_cast<const u32x2*>(ub + pix * p.ldbx + co);
          f32x4 v = acc[mi][ni];
          const float u4[4] = {bflo(uu[0]), bfhi(uu[0]), bflo(uu[1]), bfhi(uu[1])};
          if (p.bx_mode == 1) {                 // uniform: mask and scale the convolution's value, then add it to res
            const u32x2 rr = *reinterpret_cast<const u32x2*>(rb + pix * p.ldres + co);
            const float r4[4] = {bflo(rr[0]), bfhi(rr[0]), bflo(rr[1]), bfhi(rr[1])};
            float o4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gm = fmaf(u4[r], sc[r], sh[r]) > 0.f ? v[r] : 0.f;
              s1[mi][r] = fmaf(gm, u4[r], s1[mi][r]);
              s2[mi][r] += gm;
              o4[r] = fmaf(gm, sc[r], r4[r]);
            }
            *reinterpret_cast<u32x2*>(yb + pix * p.ldy + co) = u32x2{pack2bf(o4[0], o4[1]), pack2bf(o4[2], o4[3])};
          } else {
          if (p.res) {                          // uniform
            const u32x2 rr = *reinterpret_cast<const u32x2*>(rb + pix * p.ldres + co);
            v += f32x4{bflo(rr[0]), bfhi(rr[0]), bflo(rr[1]), bfhi(rr[1])};
          }
          const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(yb + pix * p.ldy + co) = o;
          const float g4[4] = {bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1])};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gm = fmaf(u4[r], sc[r], sh[r]) > 0.f ? g4[r] : 0.f;
            s1[mi][r] = fmaf(gm, u4[r], s1[mi][r]);
            s2[mi][r] += gm;
          }
          }
        }
      }
    }
    }
  }
  if constexpr (BX) {
  } else if (interior) {
    T* const d0 = yb + ((long)(oh0 + pt0 / TWT) * om * WoD + (ow0 + (pt0 % TWT) * 16 + l15) * om) * p.ldy + n0 + 4 * lg;
#pragma unroll
    for (int ni = 0; ni < NT_PIX; ++ni) {
      const int pt = pt0 + ni;
      T* dst = d0 + (long)(((pt / TWT) - (pt0 / TWT)) * om * WoD + ((pt % TWT) - (pt0 % TWT)) * 16 * om) * p.ldy;   // uniform offset
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi) {
        f32x4 v = acc[mi][ni];
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(dst + mi * 16) = v;
        } else {
          const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(dst + mi * 16) = o;
          if (p.stats) v = f32x4{bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1])};   // statistics of the STORED values
        }
        if (p.stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] = fmaf(v[r], v[r], s2[mi][r]); }
        }
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < NT_PIX; ++ni) {
      const int pt = pt0 + ni;
      const int oh = oh0 + pt / TWT, ow = ow0 + (pt % TWT) * 16 + l15;
      const bool valid = oh < p.Ho && ow < p.Wo;
      T* dst = yb + ((long)oh * om * WoD + (long)ow * om) * p.ldy;
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi) {
        const int co = n0 + mi * 16 + 4 * lg;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r];
        if (valid && co + 3 < p.Cout) {
          if constexpr (sizeof(T) == 4) {
            f32x4* d4 = reinterpret_cast<f32x4*>(dst + co);
            f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
            if (p.accumulate) o += *d4;
            *d4 = o;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = o[r];
          } else {
            u32x2* d2 = reinterpret_cast<u32x2*>(dst + co);
            if (p.accumulate) {
              const u32x2 old = *d2;
              v[0] += bflo(old[0]); v[1] += bfhi(old[0]); v[2] += bflo(old[1]); v[3] += bfhi(old[1]);
            }
            const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *d2 = o;
            v[0] = bflo(o[0]); v[1] = bfhi(o[0]); v[2] = bflo(o[1]); v[3] = bfhi(o[1]);
          }
        } else if (valid && co < p.Cout) {   // last, partial group of 4 channels
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (co + r < p.Cout) {
              if (p.accumulate) v[r] += Elem<T>::ld(dst + co + r);
              Elem<T>::st(dst + co + r, v[r]);
              v[r] = Elem<T>::rnd(v[r]);
            } else {
              v[r] = 0.f;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] = fmaf(v[r], v[r], s2[mi][r]); }
      }
    }
  }

  if (p.stats) {  // uniform
    __syncthreads();  // all fragment reads done: LDS is reused for the cross-wave reduction
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN]
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[mi][r]), c2 = row16_sum(s2[mi][r]);   // over the 16 pixels of the MFMA tile (DPP)
        if (l15 == 0) {
          const int m = mi * 16 + 4 * lg + r;
          red[(wave * 2 + 0) * BN + m] = a;
          red[(wave * 2 + 1) * BN + m] = c2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, m = tid - which * BN;
      if (n0 + m < p.Cout) {
        const float tot = red[(0 * 2 + which) * BN + m] + red[(1 * 2 + which) * BN + m] +
                          red[(2 * 2 + which) * BN + m] + red[(3 * 2 + which) * BN + m];
        const long rep = (blockIdx.x + blockIdx.z) % p.nrep;
        if (BX && p.bx_mode == 1)
          atomicAdd(reinterpret_cast<float*>(p.stats) + (((long)which * p.nrep + rep) * p.bx_groups + grp) * p.Cout + n0 + m, tot);
        else
          atomicAdd(p.stats + rep * p.rep_stride + ((long)grp * 2 + which) * p.stats_ld + n0 + m, (double)tot);
      }
    }
  }
}

template <typename T, int TH, int TW, int BN, int KS, bool DMA, bool BX = false>
int launch_fast(const FastArgs& a, size_t lds, hipStream_t s) {
  auto kern = conv_fast_kernel<T, TH, TW, BN, KS, DMA, BX>;
  static bool attr_set = false;  // per instantiation
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv_fast: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  dim3 grid(sdhip_cdiv(a.Ho, TH) * sdhip_cdiv(a.Wo, TW), sdhip_cdiv(a.Mpad, BN), a.B * a.Do);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

template <typename T, int TH, int TW, int KS, bool DMA>
int launch_fast_bn(const FastArgs& a, int bn, size_t lds, hipStream_t s) {
  switch (bn) {
    case 16: return launch_fast<T, TH, TW, 16, KS, DMA>(a, lds, s);
    case 32: return launch_fast<T, TH, TW, 32, KS, DMA>(a, lds, s);
    case 64: return launch_fast<T, TH, TW, 64, KS, DMA>(a, lds, s);
    default: return launch_fast<T, TH, TW, 128, KS, DMA>(a, lds, s);
  }
}

template <typename T, bool DMA>
int launch_fast_tile(const FastArgs& a, bool big, int ks, int bn, size_t lds, hipStream_t s) {
  if (big) return ks == 2 ? launch_fast_bn<T, 8, 32, 2, DMA>(a, bn, lds, s) : launch_fast_bn<T, 8, 32, 1, DMA>(a, bn, lds, s);
  return ks == 2 ? launch_fast_bn<T, 4, 16, 2, DMA>(a, bn, lds, s) : launch_fast_bn<T, 4, 16, 1, DMA>(a, bn, lds, s);
}

template <int TH, int TW, int KS>
int launch_fast_bx_bn(const FastArgs& a, int bn, size_t lds, hipStream_t s) {   // bf16, LDS-DMA halo, BatchNorm-backward sums
  switch (bn) {
    case 16: return launch_fast<bf16_t, TH, TW, 16, KS, true, true>(a, lds, s);
    case 32: return launch_fast<bf16_t, TH, TW, 32, KS, true, true>(a, lds, s);
    case 64: return launch_fast<bf16_t, TH, TW, 64, KS, true, true>(a, lds, s);
    default: return launch_fast<bf16_t, TH, TW, 128, KS, true, true>(a, lds, s);
  }
}

template <typename T>
int launch_fast_any(const FastArgs& a, bool big, int ks, int bn, size_t lds, hipStream_t s) {
  if (a.bx) {
    if (!a.dma || sizeof(T) != 2) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_bnbwd: bf16 without an input prologue / channel tail only");
    if (big) return ks == 2 ? launch_fast_bx_bn<8, 32, 2>(a, bn, lds, s) : launch_fast_bx_bn<8, 32, 1>(a, bn, lds, s);
    return ks == 2 ? launch_fast_bx_bn<4, 16, 2>(a, bn, lds, s) : launch_fast_bx_bn<4, 16, 1>(a, bn, lds, s);
  }
  return a.dma ? launch_fast_tile<T, true>(a, big, ks, bn, lds, s) : launch_fast_tile<T, false>(a, big, ks, bn, lds, s);
}

}  // namespace
