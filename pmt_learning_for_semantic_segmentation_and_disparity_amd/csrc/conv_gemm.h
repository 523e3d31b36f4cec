// 1x1 convolutions as a streaming GEMM for gfx950 (bf16):  Y[p][n] = act( sum_k A'[p][k] * W[n][k] + bias[n] ).
//
// A 1x1 convolution over NHWC memory IS a row-major GEMM — a pixel is a row of K contiguous channels — and every one on
// the hot path (DenseNet bottlenecks and transitions, models/densenet.py:41-45,119-128; the `conv1d_*` fusion layers of
// models/dsnet_t2.py:1074-1124 and their data gradients) moves far more bytes than it multiplies (50-100 FLOP/B against
// a ridge of ~310): the job is to keep HBM busy, not the matrix cores.  The halo-tile kernel of conv_fast.h stages one
// channel chunk, waits, multiplies one tap, waits again — with one tap per chunk nothing hides the load latency.  Here:
//
//   * persistent 8-wave workgroups (one per CU) walk a contiguous range of (cout block, 256-pixel tile) pairs and run ONE
//     continuous software pipeline over all their (tile, 64-channel chunk) stages: an NS-deep LDS ring filled by LDS-DMA
//     (`global_load_lds_dwordx4`: no VGPR round trip), NS-1 stages in flight across tile boundaries and epilogues, a
//     counted `s_waitcnt vmcnt(n)` + raw `s_barrier` per stage (never a full drain);
//   * A rows are GATHERED: each lane of a DMA instruction computes its own source address, so a row may be the
//     concatenation of two tensors (torch.cat fused away) and either of them may be a nearest-neighbour upsampled map
//     (F.interpolate(scale_factor=2^k) fused away: models/dsnet_t2.py:1211-1216 never materialises the x8 map);
//   * the BatchNorm+ReLU prologue of the DenseNet (norm1/relu1 in front of conv1) is applied to the B fragments after
//     they are read from LDS (8 fma + 8 max per fragment, hidden under the MFMAs of an HBM-bound kernel), which keeps the
//     load path pure DMA; the per-channel scale/shift live in an LDS table;
//   * epilogue as in conv_fast.h: bias / activation, 8-byte NHWC stores, BatchNorm statistics of the stored values —
//     summed in registers across all tiles of a workgroup and flushed once (f64 atomics into a replica).
//
// LDS images use conv_fast.h's lane-linear layout with the swizzle on the source address (chunk slot = c ^ (row & 6)).
#pragma once
#include "conv_fast.h"

namespace {

struct GemmSeg {
  const void* p;   // first pixel of the segment's tensor
  int ld;          // pixel stride (elements)
  int c;           // channels taken from it
  int us;          // log2 of the nearest-neighbour upsampling factor (0: same grid as the output)
};

struct GemmArgs {
  GemmSeg seg[2];
  int nseg;
  const void* wp; void* y;
  const float* bias; const float* in_scale; const float* in_shift; double* stats;
  int ldy, Cout, Mpad, K;
  int in_relu, act;
  long M;                 // output pixels (B*H*W)
  int H, W;               // output grid
  long ppg;               // pixels per statistics group
  int stats_ld, nrep; long rep_stride;
  int ntiles, nblk;       // 128-pixel tiles, cout blocks
  int dbg;                // diagnostics (SDHIP_TUNE_GEMM_DBG): 1 no stores, 2 weights fetched once per workgroup, 4 no MFMA
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 512 threads = 8 waves: wave w owns pixel rows [32w, 32w + 32) of the 256-pixel tile and all NT output channels.
// Everything in the stage loop is 32-bit and incremental (no division, no 64-bit multiply): with one or two waves per SIMD
// the loop's own instruction stream is what competes with the DMA for time.
template <int NT, int NS>
__global__ __launch_bounds__(512) void gemm1x1_kernel(const GemmArgs p) {
  constexpr int MT = 256, NT_CO = NT / 16, NT_PIX = 2;
  constexpr int WROWS = NT < 64 ? 64 : NT;        // a DMA round is 64 rows: narrow blocks pad their weight tile to one round
  constexpr int A_BYTES = MT * 128, W_BYTES = WROWS * 128, ST_BYTES = A_BYTES + W_BYTES;
  constexpr int A_INSTR = MT / 64, W_INSTR = WROWS / 64, IPS = A_INSTR + W_INSTR;   // DMA instructions per wave per stage (same for every wave)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nq = (p.K + 63) >> 6;
  unsigned char* const ring = smem;
  float* const red = reinterpret_cast<float*>(smem + NS * ST_BYTES);   // [2][NT] statistics of the workgroup (LDS atomics)
  float* const tab = red + 2 * NT;                                      // [2][nq*64]: scale, shift of the current group (prologue only)

  // contiguous range of virtual tiles v = nb * ntiles + t for this workgroup
  const int nvt = p.ntiles * p.nblk;
  const int per = (nvt + (int)gridDim.x - 1) / (int)gridDim.x;
  const int v0 = (int)blockIdx.x * per;
  const int v1 = v0 + per < nvt ? v0 + per : nvt;
  if (v0 >= v1) return;
  const int nst = (v1 - v0) * nq;

  // ---- DMA side ----
  const int rsub = tid >> 3;                      // row inside a 64-row round
  const int c_l = (tid & 7) ^ (rsub & 6);         // logical 16-byte piece this lane fetches (swizzle on the source side)
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
  const bf16_t* const wpk = (const bf16_t*)p.wp;
  int i_nb = v0 / p.ntiles, i_t = v0 - i_nb * p.ntiles, i_q = 0, i_slot = 0;   // next stage to issue
  const bf16_t* arow0[A_INSTR];                   // this lane's source rows of the tile being issued (clamped to a valid pixel)
  const bf16_t* arow1[A_INSTR];
  auto tile_rows = [&](int t) {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      long pix = (long)t * MT + j * 64 + rsub;
      pix = pix < p.M ? pix : p.M - 1;           // rows past the last pixel re-read it: their results are never stored
#pragma unroll
      for (int sgi = 0; sgi < 2; ++sgi) {
        const bf16_t* r = (const bf16_t*)sdhip_zero16;
        if (sgi < p.nseg) {
          long sp = pix;
          const int us = p.seg[sgi].us;
          if (us) {
            const unsigned hw = (unsigned)(p.H * p.W);
            const unsigned b = (unsigned)pix / hw;
            const unsigned rr = (unsigned)pix - b * hw;
            const unsigned h = rr / (unsigned)p.W, w = rr - h * (unsigned)p.W;
            sp = ((long)b * (p.H >> us) + (h >> us)) * (p.W >> us) + (w >> us);
          }
          r = (const bf16_t*)p.seg[sgi].p + sp * p.seg[sgi].ld;
        }
        if (sgi == 0) arow0[j] = r; else arow1[j] = r;
      }
    }
  };
  tile_rows(i_t);
  const int c0 = p.seg[0].c, c1 = p.nseg > 1 ? p.seg[1].c : 0;
  auto issue = [&]() {
    const unsigned dst = (unsigned)(i_slot * ST_BYTES) + wave_lds;
    const int k0 = i_q * 64 + c_l * 8;
    const bool s1 = k0 >= c0;
    const int kk = s1 ? k0 - c0 : k0;
    const bool kin = s1 ? kk < c1 : true;        // pieces past the last channel read the zero block
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const bf16_t* row = s1 ? arow1[j] : arow0[j];
      const void* src = kin ? (const void*)(row + kk) : (const void*)sdhip_zero16;
      glds16(src, dst + j * 8192);
    }
    const int n0 = i_nb * NT;
    const int mvalid = p.Mpad - n0 < NT ? p.Mpad - n0 : NT;
    const bf16_t* wb = wpk + ((long)i_q * p.Mpad + n0) * 64 + c_l * 8;
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      int m = j * 64 + rsub;
      m = m < mvalid ? m : mvalid - 1;           // rows past Mpad are never stored: re-read a valid row
      if (!(p.dbg & 16)) glds16(wb + m * 64, dst + A_BYTES + j * 8192);
    }
    if (++i_slot == NS) i_slot = 0;
    if (++i_q == nq) {
      i_q = 0;
      if (++i_t == p.ntiles) { i_t = 0; ++i_nb; }
      tile_rows(i_t);
    }
  };

  // ---- compute side ----
  f32x4 acc[NT_CO][NT_PIX];
  const int a_off = LdsRow<2>::off(l15, lg);                       // weight rows mi*16 + l15
  const int b_off = (wave * 32) * 128 + LdsRow<2>::off(l15, lg);   // pixel rows wave*32 + ni*16 + l15
  const bool tailk = (p.K & 63) != 0;             // the last chunk holds pieces past K (zero block) or a partial piece
  int cur_grp = -1, st_grp = -1, st_nb = -1;      // group whose table is loaded; (group, block) of the pending statistics
  int c_nb = v0 / p.ntiles, c_t = v0 - c_nb * p.ntiles, c_q = 0, c_slot = 0;
  const int tiles_per_grp = p.ppg > 0 ? (int)(p.ppg / MT) : 0x7fffffff;

  auto load_table = [&](int g) {                 // all threads; callers bracket it with barriers
    for (int k = tid; k < nq * 64; k += 512) {
      const bool in = k < p.K;
      tab[k] = in ? p.in_scale[(long)g * p.K + k] : 0.f;
      tab[nq * 64 + k] = in ? p.in_shift[(long)g * p.K + k] : 0.f;
    }
  };
  // statistics: every tile's epilogue adds its column sums into `red` (LDS atomics); one global flush per (group, block)
  auto flush_stats = [&]() {                     // all threads
    __syncthreads();
    if (tid < 2 * NT) {
      const int which = tid / NT, m = tid - which * NT;
      const int co = st_nb * NT + m;
      if (co < p.Cout)
        atomicAdd(p.stats + (long)(blockIdx.x % p.nrep) * p.rep_stride + ((long)st_grp * 2 + which) * p.stats_ld + co, (double)red[tid]);
      red[tid] = 0.f;
    }
    __syncthreads();
  };
  if (p.stats) {
    if (tid < 2 * NT) red[tid] = 0.f;             // published by the first stage barrier
  }

  // ---- prime the ring ----
  {
    const int pre = nst < NS - 1 ? nst : NS - 1;
    for (int s = 0; s < pre; ++s) issue();
  }

  for (int st = 0; st < nst; ++st) {
    if (c_q == 0) {
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int grp = c_t >= tiles_per_grp ? c_t / tiles_per_grp : 0;
      if (p.stats && (grp != st_grp || c_nb != st_nb)) {
        if (st_grp >= 0) flush_stats();
        st_grp = grp; st_nb = c_nb;
      }
      if (p.in_scale && grp != cur_grp) {
        __syncthreads();                          // nobody still reads the previous group's table
        load_table(grp);
        cur_grp = grp;                            // published by the stage barrier below
      }
    }
    // stage st has landed when at most the DMA instructions of the younger in-flight stages are outstanding
    const int younger = nst - 1 - st < NS - 2 ? nst - 1 - st : NS - 2;
    if (p.dbg & 16) { if (younger >= 2) wait_vm<2 * A_INSTR>(); else if (younger == 1) wait_vm<A_INSTR>(); else wait_vm<0>(); }
    else if (younger >= 2) wait_vm<2 * IPS>(); else if (younger == 1) wait_vm<IPS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage st is in LDS; stage st-1's slot is free
    if (st + NS - 1 < nst) issue();

    const unsigned char* const abuf = ring + c_slot * ST_BYTES;
    const unsigned char* const wbuf = abuf + A_BYTES;
    const bool last = c_q == nq - 1;
    const bool xform = p.in_scale != nullptr || (tailk && last);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (p.dbg & 8) break;
      u32x4 af[NT_CO], bf[NT_PIX];
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(wbuf + ((a_off + mi * 2048) ^ (ks << 6)));
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(abuf + ((b_off + ni * 2048) ^ (ks << 6)));
      if (xform) {                                // wave-uniform
        const int kb = c_q * 64 + ks * 32 + lg * 8;
        float sc[8], sh[8];
        if (p.in_scale) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(tab + kb), a1 = *reinterpret_cast<const f32x4*>(tab + kb + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(tab + nq * 64 + kb), b1 = *reinterpret_cast<const f32x4*>(tab + nq * 64 + kb + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { sc[e] = a0[e]; sc[4 + e] = a1[e]; sh[e] = b0[e]; sh[4 + e] = b1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        }
        const int nv = p.K - kb;                  // valid channels of this 8-channel piece
        const bool whole = c_q * 64 + ks * 32 + 32 <= p.K;   // wave-uniform: no lane of this k-step holds channels past K
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) {
          float f[8];
          Chunk<bf16_t>::unpack(bf[ni], f);
          if (whole) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = fmaf(f[e], sc[e], sh[e]);
              f[e] = p.in_relu ? fmaxf(v, 0.f) : v;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = fmaf(f[e], sc[e], sh[e]);
              if (p.in_relu) v = fmaxf(v, 0.f);
              f[e] = e < nv ? v : 0.f;
            }
          }
          bf[ni] = Chunk<bf16_t>::pack(f);
        }
      }
      if (!(p.dbg & 4)) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni) Mma<bf16_t>::run(acc[mi][ni], af[mi], bf[ni]);
      }
    }
    if (++c_slot == NS) c_slot = 0;

    if (last) {
      // ---- epilogue of tile (c_nb, c_t) ----
      const int n0 = c_nb * NT;
      if (p.bias) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = n0 + mi * 16 + 4 * lg + r;
            const float bv = co < p.Cout ? p.bias[co] : 0.f;
#pragma unroll
            for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni][r] += bv;
          }
        }
      }
      if (p.act) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = acc[mi][ni][r];
              acc[mi][ni][r] = p.act == 1 ? fmaxf(v, 0.f) : 1.f / (1.f + __expf(-v));
            }
      }
      bf16_t* const yb = (bf16_t*)p.y;
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) {
        const long pix = (long)c_t * MT + wave * 32 + ni * 16 + l15;
        const bool pv = pix < p.M;
        bf16_t* const dst = yb + pix * p.ldy + n0 + 4 * lg;
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi) {
          const int co = n0 + mi * 16 + 4 * lg;
          f32x4 v = acc[mi][ni];
          const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          v = f32x4{bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1])};      // statistics of the STORED values
          if (pv && co + 3 < p.Cout) {
            if (!(p.dbg & 1)) *reinterpret_cast<u32x2*>(dst + mi * 16) = o;
          } else if (pv && co < p.Cout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (co + r < p.Cout) dst[mi * 16 + r] = f2bf(v[r]); else v[r] = 0.f;
            }
          } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          acc[mi][ni] = v;                         // what was stored (zero where nothing was)
        }
      }
      if (p.stats) {                              // uniform: column sums of the tile -> LDS
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v0 = acc[mi][0][r], v1 = acc[mi][1][r];
            const float a = row16_sum(v0 + v1), c2 = row16_sum(fmaf(v0, v0, v1 * v1));
            if (l15 == 0) {
              const int m = mi * 16 + 4 * lg + r;
              atomicAdd(red + m, a);
              atomicAdd(red + NT + m, c2);
            }
          }
      }
      c_q = 0;
      if (++c_t == p.ntiles) { c_t = 0; ++c_nb; }
    } else {
      ++c_q;
    }
  }
  if (p.stats && st_grp >= 0) flush_stats();
}

inline size_t gemm_lds_bytes(int nt, int ns, int nq, bool table) {
  return (size_t)ns * (256 * 128 + (nt < 64 ? 64 : nt) * 128) + (size_t)2 * nt * 4 + (table ? (size_t)2 * nq * 64 * 4 : 0);
}

template <int NT, int NS>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
  auto kern = gemm1x1_kernel<NT, NS>;
  const int nq = (a.K + 63) / 64;
  const size_t lds = gemm_lds_bytes(NT, NS, nq, a.in_scale != nullptr);
  if (lds > 160 * 1024) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "gemm1x1: %d input channels with a prologue do not fit the LDS table", a.K);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "gemm1x1: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  const long nvt = (long)a.ntiles * a.nblk;
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  long grid = 256L * (per_cu > 2 ? 2 : per_cu);
  if (grid > nvt) grid = nvt;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// Can this 1x1 convolution take the GEMM path?  (bf16, 16-byte aligned rows, whole statistics groups per tile range.)
inline bool gemm1x1_ok(const GemmArgs& a, int groups, bool any_size = false) {
  // measured (tools/gpu_gemmdiag.py): the persistent 256-pixel tiles need >= 128 of them to keep every CU streaming; with
  // fewer pixels the halo-tile kernel's many small workgroups win
  if (!any_size && (a.M < 32768 || (a.M < 65536 && a.in_scale))) return false;
  if (a.K > 4096 || a.Mpad > 2048) return false;
  for (int i = 0; i < a.nseg; ++i) {
    if (a.seg[i].ld % 8 || ((uintptr_t)a.seg[i].p & 15)) return false;
    if (a.seg[i].us && ((a.H & ((1 << a.seg[i].us) - 1)) || (a.W & ((1 << a.seg[i].us) - 1)))) return false;
  }
  if (a.nseg == 2 && (a.seg[0].c % 8)) return false;
  if (a.ldy % 4 || ((uintptr_t)a.y & 7)) return false;
  if (groups > 1 && (a.M % groups || (a.M / groups) % 256)) return false;   // a 256-pixel tile never straddles two groups
  if (a.M >= (1L << 31) / 2) return false;
  if (a.in_scale && a.K > 1024) return false;                               // prologue table: 8 KB next to a 144 KB ring
  return true;
}

inline int launch_gemm_any(GemmArgs& a, hipStream_t s) {
  a.ntiles = (int)((a.M + 255) / 256);
  a.dbg = sdhip_diag().tune_gemm_dbg;
  // cout block: these layers are bound by streaming the pixel rows, so one block should cover all output channels
  // whenever it can (every extra block re-reads the rows); MFMA rows wasted on padding are free
  const int nt = a.Mpad <= 32 ? 32 : (a.Mpad <= 64 ? 64 : 128);
  a.nblk = sdhip_cdiv(a.Mpad, nt);
  if (nt == 128) return launch_gemm<128, 3>(a, s);
  if (nt == 64) return launch_gemm<64, 3>(a, s);
  return launch_gemm<32, 3>(a, s);
}

}  // namespace
