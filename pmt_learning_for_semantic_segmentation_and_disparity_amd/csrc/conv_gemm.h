// 1x1 convolutions as a streaming GEMM for gfx950 (bf16):  Y[p][n] = act( sum_k A'[p][k] * W[n][k] + bias[n] ).
//
// A 1x1 convolution over NHWC memory IS a row-major GEMM — a pixel is a row of K contiguous channels — and every one on
// the hot path (DenseNet bottlenecks and transitions, models/densenet.py:41-45,119-128; the `conv1d_*` fusion layers of
// models/dsnet_t2.py:1074-1124 and their data gradients) moves far more bytes than it multiplies (50-100 FLOP/B against
// a ridge of ~310): the job is to keep HBM busy, not the matrix cores.  The halo-tile kernel of conv_fast.h stages one
// channel chunk, waits, multiplies one tap, waits again — with one tap per chunk nothing hides the load latency.  Here:
//
//   * persistent workgroups (one or two per CU) walk a contiguous range of (cout block, 128-pixel tile) pairs and run ONE
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
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NT, int NS>
__global__ __launch_bounds__(256) void gemm1x1_kernel(const GemmArgs p) {
  constexpr int MT = 128, NT_CO = NT / 16, NT_PIX = 2;
  constexpr int A_BYTES = MT * 128, W_BYTES = NT * 128, ST_BYTES = A_BYTES + W_BYTES;
  constexpr int A_INSTR = MT / 32, W_INSTR = NT / 32, IPS = A_INSTR + W_INSTR;   // DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nq = (p.K + 63) >> 6;
  unsigned char* const ring = smem;
  float* const tab = reinterpret_cast<float*>(smem + NS * ST_BYTES);   // [2][nq*64]: scale, shift of the current group
  float* const red = tab + 2 * nq * 64;                                 // [4 waves][2][NT]

  // contiguous range of virtual tiles v = nb * ntiles + t for this workgroup
  const long nvt = (long)p.ntiles * p.nblk;
  const long per = (nvt + gridDim.x - 1) / gridDim.x;
  const long v0 = (long)blockIdx.x * per;
  const long v1 = v0 + per < nvt ? v0 + per : nvt;
  if (v0 >= v1) return;
  const long nst = (v1 - v0) * nq;

  // ---- DMA side ----
  const int rsub = tid >> 3;                      // row inside a 32-row round
  const int c_l = (tid & 7) ^ (rsub & 6);         // logical 16-byte piece this lane fetches (swizzle on the source side)
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem) + wave * 1024);
  const bf16_t* const wpk = (const bf16_t*)p.wp;
  long iv = v0; int iq = 0;                       // next stage to issue
  const bf16_t* arow[2][A_INSTR];                 // per segment, per round: this lane's source row (nullptr: outside)
  long i_t = -1;
  auto tile_rows = [&](long t) {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const long pix = t * MT + j * 32 + rsub;
#pragma unroll
      for (int sgi = 0; sgi < 2; ++sgi) {
        arow[sgi][j] = nullptr;
        if (sgi < p.nseg && pix < p.M) {
          long sp = pix;
          const int us = p.seg[sgi].us;
          if (us) {
            const long hw = (long)p.H * p.W;
            const long b = pix / hw;
            const int r = (int)(pix - b * hw);
            const int h = r / p.W, w = r - h * p.W;
            sp = (b * (p.H >> us) + (h >> us)) * (p.W >> us) + (w >> us);
          }
          arow[sgi][j] = (const bf16_t*)p.seg[sgi].p + sp * p.seg[sgi].ld;
        }
      }
    }
  };
  auto issue = [&](long st) {
    const int nb = (int)(iv / p.ntiles);
    const long t = iv - (long)nb * p.ntiles;
    if (t != i_t) { tile_rows(t); i_t = t; }
    const unsigned dst = (unsigned)((st % NS) * ST_BYTES) + wave_lds;
    const int k0 = iq * 64 + c_l * 8;
    const int c0 = p.seg[0].c;
    const int sgi = k0 < c0 ? 0 : 1;
    const int kk = sgi ? k0 - c0 : k0;
    const bool kin = sgi < p.nseg && kk < p.seg[sgi].c;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const bf16_t* row = sgi ? arow[1][j] : arow[0][j];
      const void* src = (kin && row) ? (const void*)(row + kk) : (const void*)sdhip_zero16;
      glds16(src, dst + j * 4096);
    }
    const int n0 = nb * NT;
    const int mvalid = p.Mpad - n0 < NT ? p.Mpad - n0 : NT;
    const bf16_t* wb = wpk + ((long)iq * p.Mpad + n0) * 64 + c_l * 8;
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
      int m = j * 32 + rsub;
      m = m < mvalid ? m : mvalid - 1;           // rows past Mpad are never stored: re-read a valid row
      glds16(wb + m * 64, dst + A_BYTES + j * 4096);
    }
    if (++iq == nq) { iq = 0; ++iv; }
  };

  // ---- compute side ----
  f32x4 acc[NT_CO][NT_PIX];
  float s1[NT_CO][4], s2[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }
  const int a_off = LdsRow<2>::off(l15, lg);                    // weight rows mi*16 + l15
  const int b_off = (wave * 32) * 128 + LdsRow<2>::off(l15, lg);   // pixel rows wave*32 + ni*16 + l15
  const bool tailk = (p.K & 7) != 0;
  int cur_grp = -1, st_grp = -1, st_nb = -1;      // group whose table is loaded; (group, block) of the pending statistics
  long cv = v0; int cq = 0;

  auto load_table = [&](int g) {                 // all threads; callers bracket it with barriers
    for (int k = tid; k < nq * 64; k += 256) {
      const bool in = k < p.K;
      tab[k] = in ? p.in_scale[(long)g * p.K + k] : 0.f;
      tab[nq * 64 + k] = in ? p.in_shift[(long)g * p.K + k] : 0.f;
    }
  };
  auto flush_stats = [&]() {                     // all threads
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[mi][r]), c2 = row16_sum(s2[mi][r]);
        if (l15 == 0) {
          const int m = mi * 16 + 4 * lg + r;
          red[(wave * 2 + 0) * NT + m] = a;
          red[(wave * 2 + 1) * NT + m] = c2;
        }
        s1[mi][r] = 0.f; s2[mi][r] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * NT) {
      const int which = tid / NT, m = tid - which * NT;
      const int co = st_nb * NT + m;
      if (co < p.Cout) {
        const float tot = red[(0 * 2 + which) * NT + m] + red[(1 * 2 + which) * NT + m] + red[(2 * 2 + which) * NT + m] +
                          red[(3 * 2 + which) * NT + m];
        atomicAdd(p.stats + (long)(blockIdx.x % p.nrep) * p.rep_stride + ((long)st_grp * 2 + which) * p.stats_ld + co, (double)tot);
      }
    }
  };

  // ---- prime the ring ----
  {
    const long pre = nst < NS - 1 ? nst : NS - 1;
    for (long s = 0; s < pre; ++s) issue(s);
  }

  for (long st = 0; st < nst; ++st) {
    const int nb = (int)(cv / p.ntiles);
    const long t = cv - (long)nb * p.ntiles;
    if (cq == 0) {
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int grp = p.ppg > 0 ? (int)((t * MT) / p.ppg) : 0;
      if (p.stats && (grp != st_grp || nb != st_nb)) {
        if (st_grp >= 0) flush_stats();
        st_grp = grp; st_nb = nb;
      }
      if (p.in_scale && grp != cur_grp) {
        __syncthreads();                          // nobody still reads the previous group's table
        load_table(grp);
        cur_grp = grp;                            // published by the stage barrier below
      }
    }
    // stage st has landed when at most the DMA instructions of the younger in-flight stages are outstanding
    const long younger = nst - 1 - st < NS - 2 ? nst - 1 - st : NS - 2;
    if (younger >= 2) wait_vm<2 * IPS>(); else if (younger == 1) wait_vm<IPS>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of stage st is in LDS; stage st-1's slot is free
    if (st + NS - 1 < nst) issue(st + NS - 1);

    const unsigned char* const abuf = ring + (st % NS) * ST_BYTES;
    const unsigned char* const wbuf = abuf + A_BYTES;
    const bool xform = p.in_scale != nullptr || (tailk && cq == nq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[NT_CO], bf[NT_PIX];
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(wbuf + ((a_off + mi * 2048) ^ (ks << 6)));
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(abuf + ((b_off + ni * 2048) ^ (ks << 6)));
      if (xform) {                                // wave-uniform
        const int kb = cq * 64 + ks * 32 + lg * 8;
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
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) {
          float f[8];
          Chunk<bf16_t>::unpack(bf[ni], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);
            if (p.in_relu) v = fmaxf(v, 0.f);
            f[e] = e < nv ? v : 0.f;
          }
          bf[ni] = Chunk<bf16_t>::pack(f);
        }
      }
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) Mma<bf16_t>::run(acc[mi][ni], af[mi], bf[ni]);
    }

    if (cq == nq - 1) {
      // ---- epilogue of tile (nb, t) ----
      const int n0 = nb * NT;
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
        const long pix = t * MT + wave * 32 + ni * 16 + l15;
        const bool pv = pix < p.M;
        bf16_t* const dst = yb + pix * p.ldy + n0 + 4 * lg;
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi) {
          const int co = n0 + mi * 16 + 4 * lg;
          f32x4 v = acc[mi][ni];
          const u32x2 o = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          v = f32x4{bflo(o[0]), bfhi(o[0]), bflo(o[1]), bfhi(o[1])};      // statistics of the STORED values
          if (pv && co + 3 < p.Cout) {
            *reinterpret_cast<u32x2*>(dst + mi * 16) = o;
          } else if (pv && co < p.Cout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (co + r < p.Cout) dst[mi * 16 + r] = f2bf(v[r]); else v[r] = 0.f;
            }
          } else {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if (p.stats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] = fmaf(v[r], v[r], s2[mi][r]); }
          }
        }
      }
      cq = 0; ++cv;
    } else {
      ++cq;
    }
  }
  if (p.stats && st_grp >= 0) flush_stats();
}

template <int NT, int NS>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
  auto kern = gemm1x1_kernel<NT, NS>;
  const int nq = (a.K + 63) / 64;
  const size_t lds = (size_t)NS * (128 * 128 + NT * 128) + (size_t)2 * nq * 64 * 4 + (size_t)4 * 2 * NT * 4;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// Can this 1x1 convolution take the GEMM path?  (bf16, 16-byte aligned rows, whole statistics groups per tile range.)
inline bool gemm1x1_ok(const GemmArgs& a, int groups) {
  if (a.M < 4096 || a.K > 4096 || a.Mpad > 2048) return false;
  for (int i = 0; i < a.nseg; ++i) {
    if (a.seg[i].ld % 8 || ((uintptr_t)a.seg[i].p & 15)) return false;
    if (a.seg[i].us && ((a.H & ((1 << a.seg[i].us) - 1)) || (a.W & ((1 << a.seg[i].us) - 1)))) return false;
  }
  if (a.nseg == 2 && (a.seg[0].c % 8)) return false;
  if (a.ldy % 4 || ((uintptr_t)a.y & 7)) return false;
  if (groups > 1 && (a.M % groups || (a.M / groups) % 128)) return false;   // a 128-pixel tile never straddles two groups
  return true;
}

inline int launch_gemm_any(GemmArgs& a, hipStream_t s) {
  a.ntiles = (int)((a.M + 127) / 128);
  // cout block: the widest that wastes at most a quarter of its MFMA rows
  int nt = 32;
  for (int cand = 128; cand >= 32; cand >>= 1)
    if ((long)sdhip_cdiv(a.Mpad, cand) * cand * 4 <= (long)a.Mpad * 5) { nt = cand; break; }
  a.nblk = sdhip_cdiv(a.Mpad, nt);
  if (nt == 128) return launch_gemm<128, 4>(a, s);
  if (nt == 64) return launch_gemm<64, 4>(a, s);
  return launch_gemm<32, 4>(a, s);
}

}  // namespace
