// Direct 2-D convolution, forward / data-gradient form, for gfx950 (MI355X).
//
// Replaces ATen/cuDNN under `conv2dSame` (models/torch_model.py:236-281), the
// stride-1 `ConvTranspose2dSame` (models/torch_model.py:284-349, run as a
// correlation with flipped, transposed weights), the plain `nn.Conv2d`s of the
// DenseNet towers (models/densenet.py:25-93,218-245) and ASPP (models/aspp.py:7-32),
// and — with dgrad-packed weights — their input gradients.
//
// One 256-thread workgroup computes a TH x TW tile of output pixels for up to BN
// output channels of one image.  Per 128-byte channel chunk of the input it stages
// the halo tile ((TH-1)s+(kh-1)d+1) x ((TW-1)s+(kw-1)d+1) pixels ONCE in LDS and
// then walks all kh*kw taps over it (im2col-free; every input byte is read from
// HBM/L2 once per chunk), with the tap weights staged next to it.  Optional fused
// prologue (per-channel affine + ReLU on the input = the BatchNorm+ReLU that
// precedes the conv) and epilogue (bias, activation, accumulate, per-channel
// sum / sum-of-squares for the BatchNorm that follows).
#include "conv_common.h"
#include "conv_fast.h"
#include "conv_band.h"
#include "conv_gemm.h"
#include "conv_thin.h"

namespace {

struct FwdArgs {
  const void* x; const void* wp; void* y;
  const float* bias; const float* in_scale; const float* in_shift; double* stats;
  ConvGeom g;
  int Cin, ldx, Cout, Mpad, ldy;
  int in_relu, groups, act, accumulate;
  int tg, vec_in, vec_out, stats_ld, nrep, pf_halo, sh, per_tap;
  long rep_stride;
};

template <typename T, int TH, int TW, int BN>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const FwdArgs p) {
  constexpr int V = Chunk<T>::N;
  constexpr int CK = 8 * V;
  constexpr int TWT = TW / 16;             // pixel tiles per tile row
  constexpr int NPT = TH * TWT;            // pixel tiles per workgroup
  constexpr int NT_PIX = NPT / 4;          // per wave
  constexpr int NT_CO = BN / 16;
  static_assert(NPT % 4 == 0 && NT_PIX >= 1, "tile must give every wave at least one 16-pixel MFMA tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const ConvGeom& g = p.g;
  const int tiles_w = (g.Wo + TW - 1) / TW;
  const int oh0 = (blockIdx.x / tiles_w) * TH, ow0 = (blockIdx.x % tiles_w) * TW;
  const int n0 = blockIdx.y * BN;
  const int b = blockIdx.z / g.Do, dz = blockIdx.z - b * g.Do;   // image = (batch, output depth slice)
  const int grp = p.groups > 1 ? b / (g.B / p.groups) : 0;
  const int s = g.stride, d = g.dil;
  // per_tap (strongly dilated kernels, e.g. ASPP's 3x3 dil 6/12): the halo tile would be mostly holes, so every tap stages
  // just the shifted TH x TW tile instead and is treated as a 1x1 convolution.
  const int IH = (TH - 1) * s + (p.per_tap ? 0 : (g.kh - 1) * d) + 1, IW = (TW - 1) * s + (p.per_tap ? 0 : (g.kw - 1) * d) + 1;
  const int ih0 = oh0 * s - g.pad_t, iw0 = ow0 * s - g.pad_l;
  const int Tn = g.kh * g.kw;
  const int nq = (p.Cin + CK - 1) / CK;     // channel chunks per depth tap; the chunk loop runs over kd*nq "chunks"
  const int nqq = g.kd * nq;
  const int mvalid = min(BN, p.Mpad - n0);  // multiple of 16
  // LDS: [halo (x2 when its loads are register-prefetched)] [weights stage buffer 0] [weights stage buffer 1]
  const int halo_bytes = (IH * IW * 128 + 15) & ~15;
  const int wbuf_bytes = p.tg * BN * 128;
  unsigned char* halo0 = smem;
  unsigned char* wl0 = smem + halo_bytes * (((TH * TW <= 64) && p.pf_halo) ? 2 : 1);

  f32x4 acc[NT_CO][NT_PIX];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT_PIX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  int pbase[NT_PIX];  // halo pixel index of this lane's output pixel for tap (0,0)
#pragma unroll
  for (int ni = 0; ni < NT_PIX; ++ni) {
    const int pt = wave * NT_PIX + ni;
    pbase[ni] = ((pt / TWT) * s) * IW + ((pt % TWT) * 16 + l15) * s;
  }

  const T* xb0 = (const T*)p.x + (long)b * g.D * g.H * g.W * p.ldx;   // depth slice 0 of batch b
  const T* wpk = (const T*)p.wp;
  const int ntg = (Tn + p.tg - 1) / p.tg;
  const int S = nqq * ntg;   // pipeline stages: (depth tap, channel chunk, tap group)
  // chunk qq = (kdi, q): input depth slice din = dz*sd + kdi - pad_d (a slice outside the volume is zero padding)
  auto slice_of = [&](int qq) { return dz * g.sd + qq / nq - g.pad_d; };
  auto slice_ok = [&](int qq) { const int din = slice_of(qq); return din >= 0 && din < g.D; };
  auto xb_of = [&](int qq) { return xb0 + (long)slice_of(qq) * g.H * g.W * p.ldx; };

  // Software pipeline: the global loads of stage s+1 (its tap-group weights and, at a chunk boundary, its halo tile)
  // are issued into registers BEFORE the MFMAs of stage s and committed to the other LDS buffer after them, so the
  // L2/HBM latency hides behind the matrix work and a stage costs one barrier.
  // PF (small 4x16 tiles only): the halo tile is register-prefetched and double-buffered too; the large 8x32 tiles
  // stage their halo with a plain copy at chunk boundaries and spend the registers on accumulators instead.
  constexpr bool PF = (TH * TW <= 64);
  constexpr int HPF = PF ? 4 : 1, WPF = 4;   // 16-byte loads per lane kept in flight for the halo / the weights
  u32x4 rh[HPF], rw[WPF];
  const bool pf_halo = PF && p.pf_halo;

  // Per-lane staging plan, computed ONCE (it does not depend on the stage): where each of this lane's 16-byte loads
  // comes from (relative to the stage base) and where it lands in LDS.  p.sh is constant for the launch: 8 chunks per
  // 128-byte row, or 4 when the whole reduction fits one half row (Cin <= CK/2).
  const int sh = p.sh;
  const unsigned magic_iw = div_magic(IW);
  int h_src[HPF], h_lds[HPF], h_ch[HPF];   // halo: element offset inside the image (or -1), LDS byte offset, channel offset
  if (pf_halo) {
    const int total = (IH * IW) << sh;
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      const int i = tid + j * 256;
      h_src[j] = -1; h_lds[j] = -1; h_ch[j] = 0;
      if (i < total) {
        const int pix = i >> sh, c = i & ((1 << sh) - 1);
        const int ih = fast_div(pix, IW, magic_iw), iw = pix - ih * IW;
        const int gh = ih0 + ih, gw = iw0 + iw;
        h_lds[j] = lds_off(pix, c);
        h_ch[j] = c * V;
        if (gh >= 0 && gh < g.H && gw >= 0 && gw < g.W) h_src[j] = (gh * g.W + gw) * p.ldx + c * V;
      }
    }
  }
  int w_src[WPF], w_lds[WPF], w_tl[WPF];   // weights: element offset inside the stage's packed block, LDS offset, tap
  {
    const unsigned magic_m = div_magic(mvalid);
#pragma unroll
    for (int j = 0; j < WPF; ++j) {
      const int i = tid + j * 256;
      const int row = i >> sh, c = i & ((1 << sh) - 1);
      const int tl = fast_div(row, mvalid, magic_m), m = row - tl * mvalid;
      w_tl[j] = tl;
      w_src[j] = (tl * p.Mpad + m) * CK + c * V;
      w_lds[j] = lds_off(tl * BN + m, c);
    }
  }

  auto halo_issue = [&](int qq) {   // vector path only (host guarantees vec_in when pf_halo)
    const int q = qq % nq;
    const bool ok = slice_ok(qq);
    const T* xb = xb_of(qq);
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      rh[j] = u32x4{0u, 0u, 0u, 0u};
      if (ok && h_src[j] >= 0 && q * CK + h_ch[j] < p.Cin)
        rh[j] = *reinterpret_cast<const u32x4*>(xb + h_src[j] + q * CK);
    }
  };
  auto halo_commit = [&](int qq, unsigned char* dst) {
    const int q = qq % nq;
    const bool ok = slice_ok(qq);
#pragma unroll
    for (int j = 0; j < HPF; ++j) {
      if (h_lds[j] >= 0) {
        u32x4 raw = rh[j];
        if (p.in_scale && ok && h_src[j] >= 0) {
          const int ch0 = q * CK + h_ch[j];
          if (ch0 < p.Cin) {   // zero padding (spatial or channel) stays zero
            float f[V];
            Chunk<T>::unpack(raw, f);
            const float* sc = p.in_scale + grp * p.Cin + ch0;
            const float* sf = p.in_shift + grp * p.Cin + ch0;
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float v = fmaf(f[e], sc[e], sf[e]);
              f[e] = p.in_relu ? fmaxf(v, 0.f) : v;
            }
            raw = Chunk<T>::pack(f);
          }
        }
        *reinterpret_cast<u32x4*>(dst + h_lds[j]) = raw;
      }
    }
  };
  auto halo_sync_stage = [&](int qq, unsigned char* dst) {   // big halo tiles: plain staged copy (4 loads in flight)
    const int q = qq % nq;
    StageSrc ss;
    ss.base = xb_of(qq); ss.H = slice_ok(qq) ? g.H : 0; ss.W = g.W; ss.ld = p.ldx; ss.C = p.Cin;
    ss.h0 = ih0; ss.w0 = iw0; ss.IH = IH; ss.IW = IW;
    ss.scale = p.in_scale ? p.in_scale + grp * p.Cin : nullptr;
    ss.shift = p.in_scale ? p.in_shift + grp * p.Cin : nullptr;
    ss.relu = p.in_relu; ss.vec = p.vec_in; ss.magic_iw = magic_iw;
    stage_tile<T, 4>(dst, ss, q, sh, tid);
  };
  auto w_issue = [&](int qq, int t0) {   // packed weights are [kd][nq][T][Mpad][CK]: chunk index qq as is
    const int nt = min(p.tg, Tn - t0);
    const T* base = wpk + ((long)(qq * Tn + t0) * p.Mpad + n0) * CK;
#pragma unroll
    for (int j = 0; j < WPF; ++j)
      if (w_tl[j] < nt) rw[j] = *reinterpret_cast<const u32x4*>(base + w_src[j]);
  };
  auto w_commit = [&](int q, int t0, unsigned char* dst) {
    const int nt = min(p.tg, Tn - t0);
#pragma unroll
    for (int j = 0; j < WPF; ++j)
      if (w_tl[j] < nt) *reinterpret_cast<u32x4*>(dst + w_lds[j]) = rw[j];
  };
  // Fragment addressing.  lds_off(row, c) = row*128 + ((c ^ s(row)) << 4) with s(row) = (row>>1)&7 and c = 4*ks + lg, i.e.
  // row*128 + ((ks<<6) ^ (lg<<4) ^ (s(row)<<4)).  For the weight rows s() does not depend on the tap (tap stride BN*128 is
  // a multiple of 16 rows), so everything but the k-step bit is hoisted out of the tap loop.
  int a_off[NT_CO];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi) {
    const int row = mi * 16 + l15;
    a_off[mi] = row * 128 + ((lg << 4) ^ (((row >> 1) & 7) << 4));
  }
  auto compute = [&](int qq, int t0, const unsigned char* halo, const unsigned char* wl) {
    const int cq = min(CK, p.Cin - (qq % nq) * CK);
    const bool two = (cq + CK / 2 - 1) / (CK / 2) == 2;   // second 64-byte k-step present (wave-uniform)
    const int nt = min(p.tg, Tn - t0);
    int khi = t0 / g.kw, kwi = t0 - khi * g.kw;
    for (int tl = 0; tl < nt; ++tl) {
      const int toff = p.per_tap ? 0 : (khi * d) * IW + kwi * d;
      if (++kwi == g.kw) { kwi = 0; ++khi; }
      const unsigned char* wt = wl + tl * (BN * 128);
      int b_off[NT_PIX];
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) {
        const int row = pbase[ni] + toff;
        b_off[ni] = row * 128 + ((lg << 4) ^ (((row >> 1) & 7) << 4));
      }
      // both k-steps' fragments are requested up front: the second k-step's LDS latency hides behind the first's MFMAs
      u32x4 af0[NT_CO], bf0[NT_PIX], af1[NT_CO], bf1[NT_PIX];
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
        if (mi * 16 < mvalid) af0[mi] = *reinterpret_cast<const u32x4*>(wt + a_off[mi]);
#pragma unroll
      for (int ni = 0; ni < NT_PIX; ++ni) bf0[ni] = *reinterpret_cast<const u32x4*>(halo + b_off[ni]);
      if (two) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
          if (mi * 16 < mvalid) af1[mi] = *reinterpret_cast<const u32x4*>(wt + (a_off[mi] ^ 64));
#pragma unroll
        for (int ni = 0; ni < NT_PIX; ++ni) bf1[ni] = *reinterpret_cast<const u32x4*>(halo + (b_off[ni] ^ 64));
      }
#pragma unroll
      for (int mi = 0; mi < NT_CO; ++mi)
        if (mi * 16 < mvalid) {
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni) Mma<T>::run(acc[mi][ni], af0[mi], bf0[ni]);
        }
      if (two) {
#pragma unroll
        for (int mi = 0; mi < NT_CO; ++mi)
          if (mi * 16 < mvalid) {
#pragma unroll
            for (int ni = 0; ni < NT_PIX; ++ni) Mma<T>::run(acc[mi][ni], af1[mi], bf1[ni]);
          }
      }
    }
  };

  if (p.per_tap) {   // one stage per (chunk, tap): shifted tile + that tap's weights, no prefetch (rare, small layers)
    for (int st = 0; st < nqq * Tn; ++st) {
      const int q = st / Tn, t = st - q * Tn;   // q runs over (depth tap, channel chunk)
      const int khi = t / g.kw, kwi = t - khi * g.kw;
      __syncthreads();
      StageSrc ss;
      ss.base = xb_of(q); ss.H = slice_ok(q) ? g.H : 0; ss.W = g.W; ss.ld = p.ldx; ss.C = p.Cin;
      ss.h0 = ih0 + khi * d; ss.w0 = iw0 + kwi * d; ss.IH = IH; ss.IW = IW;
      ss.scale = p.in_scale ? p.in_scale + grp * p.Cin : nullptr;
      ss.shift = p.in_scale ? p.in_shift + grp * p.Cin : nullptr;
      ss.relu = p.in_relu; ss.vec = p.vec_in; ss.magic_iw = magic_iw;
      stage_tile<T, 4>(halo0, ss, q % nq, sh, tid);
      w_issue(q, t);
      w_commit(q, t, wl0);
      __syncthreads();
      compute(q, t, halo0, wl0);
    }
  } else {
  // prologue: stage 0
  if (pf_halo) halo_issue(0); else halo_sync_stage(0, halo0);
  w_issue(0, 0);
  for (int st = 0; st < S; ++st) {
    const int q = st / ntg, t0 = (st - q * ntg) * p.tg;
    unsigned char* halo = halo0 + ((pf_halo && (q & 1)) ? halo_bytes : 0);
    unsigned char* wl = wl0 + (st & 1) * wbuf_bytes;
    if (pf_halo && t0 == 0) halo_commit(q, halo);
    w_commit(q, t0, wl);
    __syncthreads();                      // stage st is visible; every wave has finished stage st-1
    const int sn = st + 1;
    int qn = 0, t0n = 0;
    if (sn < S) {
      qn = sn / ntg; t0n = (sn - qn * ntg) * p.tg;
      if (pf_halo && t0n == 0) halo_issue(qn);
      w_issue(qn, t0n);
    }
    compute(q, t0, halo, wl);
    if (!pf_halo && sn < S && t0n == 0) {   // chunk boundary with a single (large) halo buffer
      __syncthreads();
      halo_sync_stage(qn, halo0);
    }
  }
  }

  // ---- epilogue: bias / activation / accumulate / store / BN statistics ----
  float s1[NT_CO][4], s2[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }

  T* yb = (T*)p.y + (long)blockIdx.z * g.Ho * g.Wo * p.ldy;
#pragma unroll
  for (int ni = 0; ni < NT_PIX; ++ni) {
    const int pt = wave * NT_PIX + ni;
    const int oh = oh0 + pt / TWT, ow = ow0 + (pt % TWT) * 16 + l15;
    const bool valid = oh < g.Ho && ow < g.Wo;
    T* dst = yb + ((long)oh * g.Wo + ow) * p.ldy;
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
      if (mi * 16 >= mvalid) continue;
      const int co = n0 + mi * 16 + 4 * lg;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[mi][ni][r];
        if (p.bias && co + r < p.Cout) x += p.bias[co + r];
        if (p.act == 1) x = fmaxf(x, 0.f);
        else if (p.act == 2) x = 1.f / (1.f + __expf(-x));
        v[r] = x;
      }
      if (!valid) continue;
      if (p.vec_out && co + 3 < p.Cout) {
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
      } else {
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
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[mi][r] += v[r]; s2[mi][r] = fmaf(v[r], v[r], s2[mi][r]); }
    }
  }

  if (p.stats) {  // uniform
    __syncthreads();  // all fragment reads done: LDS is reused for the cross-wave reduction
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][BN]
#pragma unroll
    for (int mi = 0; mi < NT_CO; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[mi][r], c2 = s2[mi][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c2 += __shfl_xor(c2, o, 64); }
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
        atomicAdd(p.stats + (long)((blockIdx.x + blockIdx.z) % p.nrep) * p.rep_stride + ((long)grp * 2 + which) * p.stats_ld + n0 + m, (double)tot);
      }
    }
  }
}

template <typename T, int TH, int TW, int BN>
int launch(const FwdArgs& a, size_t lds, hipStream_t s) {
  auto kern = conv_fwd_kernel<T, TH, TW, BN>;
  static size_t attr_set = 0;  // per instantiation
  if (lds > 64 * 1024 && lds > attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv_fwd: cannot raise dynamic LDS limit");
    attr_set = 160 * 1024;
  }
  const ConvGeom& g = a.g;
  dim3 grid(sdhip_cdiv(g.Ho, TH) * sdhip_cdiv(g.Wo, TW), sdhip_cdiv(a.Mpad, BN), g.B * g.Do);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

template <typename T, int TH, int TW>
int launch_bn(const FwdArgs& a, int bn, size_t lds, hipStream_t s) {
  switch (bn) {
    case 16: return launch<T, TH, TW, 16>(a, lds, s);
    case 32: return launch<T, TH, TW, 32>(a, lds, s);
    case 64: return launch<T, TH, TW, 64>(a, lds, s);
    default: return launch<T, TH, TW, 128>(a, lds, s);
  }
}

size_t halo_bytes_of(const ConvGeom& g, int th, int tw, int per_tap) {
  const int eh = per_tap ? 0 : (g.kh - 1) * g.dil, ew = per_tap ? 0 : (g.kw - 1) * g.dil;
  const int IH = (th - 1) * g.stride + eh + 1, IW = (tw - 1) * g.stride + ew + 1;
  return ((size_t)IH * IW * 128 + 15) & ~(size_t)15;
}

}  // namespace

// consumer-side BatchNorm finalize of the input prologue (sdhip_conv2d_fwd_bnpro; FastArgs::pin_*)
struct ProStats {
  const double* stats; int ld, nrep;
  const float* gamma; const float* beta;
  float* scale; float* shift; float* mean; float* invstd; float* rmean; float* rvar;
  float eps, momentum; double count;
  const double* pend; double* fold; int pend_ld, pend_nrep, pend_c0, pend_n;
};

static int conv2d_fwd_impl(const void* x, const void* wpacked, void* y,
                           const float* bias, const float* in_scale, const float* in_shift,
                           double* stats, int stats_ld, int stats_nrep,
                           int B, int H, int W, int Cin, int ldx,
                           int Ho, int Wo, int Cout, int ldy,
                           int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                           int D, int Do, int kd, int sd, int pad_d,
                           int in_relu, int groups, int act, int accumulate,
                           int dtype, void* stream, int omul, int ooz, int ooy, int oox,
                           const void* bx = nullptr, int ldbx = 0, const float* bsc = nullptr, const float* bsh = nullptr,
                           const void* addend = nullptr, int ldadd = 0, int bx_mode = 0, const ProStats* ps = nullptr) {
  SDHIP_CHECK_ARG(x && wpacked && y, "conv2d_fwd: null pointer");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "conv2d_fwd: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "conv2d_fwd: empty tensor");
  SDHIP_CHECK_ARG(ldx >= Cin && ldy >= Cout, "conv2d_fwd: pixel stride smaller than channel count");
  SDHIP_CHECK_ARG(kh >= 1 && kw >= 1 && stride >= 1 && dil >= 1, "conv2d_fwd: bad kernel geometry");
  SDHIP_CHECK_ARG(groups >= 1 && B % groups == 0, "conv2d_fwd: batch %d not divisible by %d stat groups", B, groups);
  SDHIP_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd: in_scale/in_shift must come together");
  SDHIP_CHECK_ARG(((uintptr_t)wpacked & 15) == 0, "conv2d_fwd: packed weights must be 16-byte aligned");
  const int V = dtype == SDHIP_BF16 ? 8 : 4;
  const int es = conv_esize(dtype);
  FwdArgs a;
  a.x = x; a.wp = wpacked; a.y = y; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.stats = stats;
  SDHIP_CHECK_ARG(D >= 1 && Do >= 1 && kd >= 1 && sd >= 1 && (long)B * Do <= 65535, "conv2d_fwd: bad depth geometry");
  a.g = ConvGeom{B, H, W, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l, D, Do, kd, sd, pad_d};
  a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.Mpad = (Cout + 15) & ~15; a.ldy = ldy;
  a.in_relu = in_relu; a.groups = groups; a.act = act; a.accumulate = accumulate;
  a.stats_ld = stats_ld > 0 ? stats_ld : Cout;
  a.nrep = stats_nrep > 0 ? stats_nrep : 1;
  a.rep_stride = (long)groups * 2 * a.stats_ld;
  a.vec_in = (ldx % V == 0) && (((uintptr_t)x & 15) == 0);   // channel tails are masked in stage_tile
  a.vec_out = (ldy % 4 == 0) && (((uintptr_t)y % (4 * es)) == 0);
  // output channels per workgroup: the widest block that wastes at most a quarter of its MFMA rows on padding
  int bn = 16;
  for (int cand = 128; cand >= 32; cand >>= 1)
    if ((long)sdhip_cdiv(a.Mpad, cand) * cand * 4 <= (long)a.Mpad * 5) { bn = cand; break; }
  // short reductions (1x1 over <= 128 channels ...) are bound by staging the input tile, not by MFMA rows: one pass
  if ((long)kh * kw * kd * Cin <= 128 && a.Mpad <= 128) bn = a.Mpad > 64 ? 128 : (a.Mpad > 32 ? 64 : (a.Mpad > 16 ? 32 : 16));
  const int T = kh * kw;
  const long blocks_big = (long)sdhip_cdiv(Ho, 8) * sdhip_cdiv(Wo, 32) * B * Do * sdhip_cdiv(a.Mpad, bn);
  const SdhipDiag& dg = sdhip_diag();
  const int tune_big = dg.tune_big, tune_split = dg.tune_split;
  const int tune_ksoft_small = 80, tune_ksoft_big = 80;
  bool big = (blocks_big >= tune_big && Wo >= 24) || dg.conv_big;
  // stride-2 volumes (hourglass conv1 / conv3 and the adjoint of conv5 / conv6, stackhourglass.py:13-29): the 8x32 tile's halo is
  // 17 x 72 rows per depth tap — 78 KB single-buffered, one workgroup per CU, staging and arithmetic in turn (0.26 PFLOP/s);
  // the 4x16 tile's 9 x 40 rows double-buffer and leave room for a second workgroup
  if (stride == 2 && kd > 1 && !dg.conv_big && dg.tune_s2_small) big = false;
  if (!big) {
    // small feature maps (DenseNet blocks 2-4, pooled pyramids): the launch cannot fill 256 CUs with pixel tiles alone,
    // so split the output channels over more workgroups (the input tile is re-read from L2, the serial
    // chunk-by-chunk latency chain per workgroup gets shorter and more of them overlap per CU).
    const long px_blocks = (long)sdhip_cdiv(Ho, 4) * sdhip_cdiv(Wo, 16) * B * Do;
    while (bn > 32 && px_blocks * sdhip_cdiv(a.Mpad, bn) < tune_split) bn >>= 1;
  }
  const int rows = a.Mpad < bn ? a.Mpad : bn;
  const size_t kMax = 160 * 1024, kSoft = 80 * 1024;   // kSoft: two workgroups per CU
  const int CKh = (dtype == SDHIP_BF16 ? 64 : 32);
  const int chunks_per_row = Cin <= CKh / 2 ? 4 : 8;
  hipStream_t s = (hipStream_t)stream;
  const int per_tap_any = (kh > 1 || kw > 1) && dil >= 4 && kd == 1;
  // ---- thin path (conv_thin.h): <= 8 input channels -> 1 output channel on the vector ALUs ----
  if (!ps && !bx && !addend && omul == 1 && Cout == 1 && Cin <= V && ldx % V == 0 && ((uintptr_t)x & 15) == 0 && stride == 1 && kd == 1 && D == 1 && Do == 1 && !in_scale &&
      !accumulate && kh * kw <= kThinMaxT && Ho == H + 2 * pad_t - dil * (kh - 1) && Wo == W + 2 * pad_l - dil * (kw - 1) &&
      pad_t >= 0 && pad_l >= 0 && !dg.conv_no_thin) {
    ThinArgs t;
    t.x = x; t.wp = wpacked; t.y = y; t.bias = bias; t.stats = stats;
    t.B = B; t.H = H; t.W = W; t.Ho = Ho; t.Wo = Wo; t.kh = kh; t.kw = kw; t.dil = dil; t.pad_t = pad_t; t.pad_l = pad_l;
    t.Cin = Cin; t.ldx = ldx; t.ldy = ldy; t.Mpad = a.Mpad; t.act = act; t.bpg = B / groups;
    t.stats_ld = a.stats_ld; t.nrep = a.nrep; t.rep_stride = a.rep_stride;
    dim3 grid(sdhip_cdiv(Wo, 256), Ho, B);
    if (dtype == SDHIP_BF16) hipLaunchKernelGGL(conv_thin_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, t);
    else hipLaunchKernelGGL(conv_thin_fwd_kernel<float>, grid, dim3(256), 0, s, t);
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  // ---- 16..64 input channels into ONE output map (conv_thin.h): taps as the MFMA rows, fixed-order sum over taps ----
  if (!ps && !bx && !addend && omul == 1 && dtype == SDHIP_BF16 && fanin_ok(Cin, Cout, kh, kw, stride, dil, kd, sd, ldx, x) && !in_scale && !accumulate &&
      !stats && (long)H * W * ldx < (1L << 31) && !dg.conv_no_thin) {
    FaninArgs t;
    t.x = x; t.wp = wpacked; t.y = y; t.bias = bias;
    t.B = B; t.H = H; t.W = W; t.Ho = Ho; t.Wo = Wo; t.kh = kh; t.kw = kw; t.pad_t = pad_t; t.pad_l = pad_l;
    t.D = D; t.Do = Do; t.pad_d = pad_d;
    t.Cin = Cin; t.ldx = ldx; t.ldy = ldy; t.Mpad = a.Mpad; t.act = act;
    t.dpw = 1; t.zsegs = Do; t.npxp = 0; t.pitch = 0;
    const int rc = launch_fanin(t, kd, s);
    if (rc != 1) return rc;                     // 1: not launchable (LDS / grid limits) -> the kernels below
  }
  // ---- one input channel fanned out to <= 64 output channels (conv_thin.h): taps as the MFMA reduction axis ----
  if (!ps && !bx && !addend && omul == 1 && dtype == SDHIP_BF16 && fanout_ok(Cin, Cout, kh * kw, stride, kd, sd, ldy, y) && !in_scale &&
      !accumulate && !stats && !bias && act == 0 && (kh - 1) * dil <= 31 && (kw - 1) * dil <= 31 && (long)sdhip_cdiv(Ho, 8) * sdhip_cdiv(Wo, 32) < (1L << 31) &&
      B <= 65535 && !dg.conv_no_thin) {
    FanArgs t;
    t.x = x; t.wp = wpacked; t.y = y;
    t.B = B; t.H = H; t.W = W; t.Ho = Ho; t.Wo = Wo; t.kh = kh; t.kw = kw; t.dil = dil; t.pad_t = pad_t; t.pad_l = pad_l;
    t.ldx = ldx; t.Cout = Cout; t.Mpad = a.Mpad; t.ldy = ldy;
    t.D = D; t.Do = Do; t.kd = kd; t.pad_d = pad_d; t.dpw = 1; t.zsegs = Do;
    return launch_fanout(t, s);
  }
  // ---- 1x1 as a streaming GEMM (conv_gemm.h): bf16, plain stride-1 1x1 over whole images ----
  if (!ps && !bx && !addend && omul == 1 && dtype == SDHIP_BF16 && kh == 1 && kw == 1 && kd == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && D == 1 && Do == 1 &&
      Ho == H && Wo == W && !accumulate && !dg.conv_generic && !dg.conv_no_gemm) {
    GemmArgs g;
    g.seg[0] = GemmSeg{x, ldx, Cin, 0};
    g.seg[1] = GemmSeg{nullptr, 0, 0, 0};
    g.nseg = 1;
    g.wp = wpacked; g.y = y; g.bias = bias; g.in_scale = in_scale; g.in_shift = in_shift; g.stats = stats;
    g.ldy = ldy; g.Cout = Cout; g.Mpad = a.Mpad; g.K = Cin; g.in_relu = in_relu; g.act = act;
    g.M = (long)B * H * W; g.H = H; g.W = W; g.ppg = g.M / groups;
    g.stats_ld = a.stats_ld; g.nrep = a.nrep; g.rep_stride = a.rep_stride;
    if (gemm1x1_ok(g, (in_scale || stats) ? groups : 1)) return launch_gemm_any(g, s);
  }
  // ---- persistent whole-CU kernel (conv_band.h): the full-resolution 5x5 layers (<= 64 channels either side) and the 3x3
  //      layers with <= 32 input channels (weights resident in LDS), bf16 ----
  const bool band5 = kh == 5 && kw == 5 && (Cin <= 32 || Cin == 64);
  const bool band3 = kh == 3 && kw == 3 && Cin <= 32 && !dg.conv_no_band3;
  if (!ps && !bx && omul == 1 && dtype == SDHIP_BF16 && (band5 || band3) && kd == 1 && stride == 1 && dil == 1 && D == 1 && Do == 1 && !in_scale &&
      !((accumulate || addend) && stats) && !(accumulate && addend) && (!addend || (ldadd % 8 == 0 && ((uintptr_t)addend & 15) == 0)) && Cin % 8 == 0 && a.Mpad <= 64 && a.Mpad >= 32 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 &&
      (long)B * H * W * ldx * 2 < (long)kBandOob && band_ok(sdhip_cdiv(Ho, 16) * sdhip_cdiv(Wo, 32) * B) && !dg.conv_generic && !dg.conv_no_band) {
    BandArgs f;
    f.x = x; f.wp = wpacked; f.y = y; f.bias = bias; f.stats = stats;
    f.B = B; f.H = H; f.W = W; f.Ho = Ho; f.Wo = Wo; f.pad_t = pad_t; f.pad_l = pad_l;
    f.Cin = Cin; f.ldx = ldx; f.Cout = Cout; f.Mpad = a.Mpad; f.ldy = ldy;
    f.bpg = B / groups; f.act = act; f.stats_ld = a.stats_ld; f.nrep = a.nrep; f.rep_stride = a.rep_stride;
    f.res = addend ? addend : (accumulate ? y : nullptr); f.ldres = addend ? ldadd : ldy;
    f.D = 1; f.Do = 1; f.pad_d = 0;
    return band5 ? launch_band<5>(f, s) : launch_band<3>(f, s);
  }
  // ---- the same kernel over volumes: 3x3x3, <= 32 channels on both sides (PSMNet's 32 -> 32 stack, stackhourglass.py:59-102):
  //      every depth tap is a chunk of its tile (its own input slice and 3x3 weights), z runs fastest over a workgroup's tiles ----
  if (!ps && !bx && omul == 1 && dtype == SDHIP_BF16 && kh == 3 && kw == 3 && kd == 3 && sd == 1 && stride == 1 && dil == 1 && Cin <= 32 && Cin % 8 == 0 &&
      a.Mpad == 32 && !in_scale && !((accumulate || addend) && stats) && !(accumulate && addend) &&
      (!addend || (ldadd % 8 == 0 && ((uintptr_t)addend & 15) == 0)) && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 &&
      (long)B * D * H * W * ldx * 2 < (long)kBandOob && band_ok(sdhip_cdiv(Ho, 16) * sdhip_cdiv(Wo, 32) * B * Do) && !dg.conv_generic && !dg.conv_no_band && !dg.conv_no_band3) {
    BandArgs f;
    f.x = x; f.wp = wpacked; f.y = y; f.bias = bias; f.stats = stats;
    f.B = B; f.H = H; f.W = W; f.Ho = Ho; f.Wo = Wo; f.pad_t = pad_t; f.pad_l = pad_l;
    f.D = D; f.Do = Do; f.pad_d = pad_d;
    f.Cin = Cin; f.ldx = ldx; f.Cout = Cout; f.Mpad = a.Mpad; f.ldy = ldy;
    f.bpg = (B / groups) * Do; f.act = act; f.stats_ld = a.stats_ld; f.nrep = a.nrep; f.rep_stride = a.rep_stride;
    f.res = addend ? addend : (accumulate ? y : nullptr); f.ldres = addend ? ldadd : ldy;
    return launch_band<3, 3>(f, s);
  }
  if (addend && !bx) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_add: only the persistent 5x5 kernel adds a second tensor in its epilogue (bf16, <= 64 channels, >= 192 tiles of 16x32)");
  // ---- fast path (conv_fast.h): 16-byte-aligned pixels on both sides, halo-tile mode ----
  if (!per_tap_any && ldx % V == 0 && ((uintptr_t)x & 15) == 0 && a.vec_out &&
      (long)H * W * ldx < (1L << 31) && !dg.conv_generic) {
    const int ks = chunks_per_row == 4 ? 1 : 2;
    const int rb = 64 * ks, px_per_round = 256 / (4 * ks);
    FastArgs f;
    f.x = x; f.wp = wpacked; f.y = y; f.bias = bias; f.in_scale = in_scale; f.in_shift = in_shift; f.stats = stats;
    f.B = B; f.H = H; f.W = W; f.Ho = Ho; f.Wo = Wo; f.kh = kh; f.kw = kw; f.stride = stride; f.dil = dil; f.pad_t = pad_t; f.pad_l = pad_l;
    f.D = D; f.Do = Do; f.kd = kd; f.sd = sd; f.pad_d = pad_d;
    f.Cin = Cin; f.ldx = ldx; f.Cout = Cout; f.Mpad = a.Mpad; f.ldy = ldy;
    f.in_relu = in_relu; f.bpg = B / groups; f.act = act; f.accumulate = accumulate;
    f.stats_ld = a.stats_ld; f.nrep = a.nrep; f.rep_stride = a.rep_stride; f.tail = (Cin % V) != 0;
    f.omul = omul; f.ooz = ooz; f.ooy = ooy; f.oox = oox;
    f.bx = bx; f.ldbx = ldbx; f.bsc = bsc; f.bsh = bsh;
    f.res = bx ? addend : nullptr; f.ldres = ldadd; f.bx_mode = bx_mode; f.bx_groups = groups;
    f.dma = !in_scale && !f.tail && !ps;
    f.pin_stats = nullptr;
    if (ps) {
      f.pin_stats = ps->stats; f.pin_ld = ps->ld; f.pin_nrep = ps->nrep; f.pin_groups = groups; f.pin_gamma = ps->gamma; f.pin_beta = ps->beta;
      f.pin_scale = ps->scale; f.pin_shift = ps->shift; f.pin_mean = ps->mean; f.pin_invstd = ps->invstd; f.pin_rmean = ps->rmean; f.pin_rvar = ps->rvar;
      f.pin_eps = ps->eps; f.pin_momentum = ps->momentum; f.pin_count = ps->count;
      f.pend_stats = ps->pend; f.pin_fold = ps->fold; f.pend_ld = ps->pend_ld; f.pend_nrep = ps->pend_nrep; f.pend_c0 = ps->pend_c0; f.pend_n = ps->pend ? ps->pend_n : 0;
    }
    bool fbig = big;
    for (int attempt = 0; attempt < 2; ++attempt) {
      const int th = fbig ? 8 : 4, tw = fbig ? 32 : 16;
      const int IH = (th - 1) * stride + (kh - 1) * dil + 1, IW = (tw - 1) * stride + (kw - 1) * dil + 1;
      const int IWp = (IW + 7) & ~7;
      const int hrows = ((IH * IWp + px_per_round - 1) / px_per_round) * px_per_round;   // whole load rounds
      const int nqq_f = kd * (ks == 2 ? sdhip_cdiv(Cin, CKh) : 1);
      const size_t hb = (size_t)hrows * rb * ((fbig || nqq_f == 1) ? 1 : 2);   // small tiles double-buffer the halo across chunks
      if (!fbig && hrows > ((f.dma && ks == 1) ? 6 : 5) * px_per_round) { fbig = true; continue; }   // small-tile halo prefetch plan: 5 / 6 rounds (conv_fast.h HPF)
      auto wbuf = [&](int tgv) { return (size_t)(((tgv * bn + px_per_round - 1) / px_per_round) * px_per_round) * rb; };
      int tg = T;
      const size_t ksoft_f = (size_t)(fbig ? tune_ksoft_big : tune_ksoft_small) * 1024;
      size_t lds;
      if (nqq_f == 1 && hb + wbuf(T) <= ksoft_f) {
        lds = hb + wbuf(T);              // the whole kernel is ONE stage: all taps resident, no second weight buffer
      } else {
        while (tg > 1 && hb + 2 * wbuf(tg) > ksoft_f) --tg;
        lds = hb + 2 * wbuf(tg);
      }
      if (lds < 4096) lds = 4096;
      f.pin_off = 0; f.pin_cs = 0;
      if (ps) { f.pin_off = (int)lds; f.pin_cs = (Cin + 63) & ~63; lds += 2 * (size_t)f.pin_cs * sizeof(float); }   // the finalize table sits behind everything else
      if (lds > kMax) { if (!fbig) break; fbig = false; continue; }
      f.tg = tg;
      f.magic_iwp = IWp > 1 ? (unsigned)(0x100000000ULL / (unsigned)IWp) + 1u : 0u;
      f.tiles_w = sdhip_cdiv(Wo, tw);
      f.magic_tw = f.tiles_w > 1 ? (unsigned)(0x100000000ULL / (unsigned)f.tiles_w) + 1u : 0u;
      f.ntg = sdhip_cdiv(T, tg);
      if ((long)f.tiles_w * sdhip_cdiv(Ho, th) >= 65536) break;   // fast_div range: general kernel for gigantic images
      return dtype == SDHIP_BF16 ? launch_fast_any<bf16_t>(f, fbig, ks, bn, lds, s) : launch_fast_any<float>(f, fbig, ks, bn, lds, s);
    }
  }
  if (omul != 1) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_phase: interleaved output needs the aligned fast path (ldx %% 8, ldy %% 4)");
  if (ps) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_bnpro: needs the aligned fast path (ldx %% 8, ldy %% 4, halo tile within LDS)");
  if (bx) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_bnbwd: needs the aligned fast path (ldx %% 8, ldy %% 4, halo tile within LDS)");
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int th = big ? 8 : 4, tw = big ? 32 : 16;
    const int per_tap = per_tap_any;   // halo would be >= 4x the tile in each direction's holes
    const size_t halo = halo_bytes_of(a.g, th, tw, per_tap);
    const long halo_loads = (long)(halo / 128) * chunks_per_row;
    // register-prefetched (double-buffered) halo: small tiles, at most 4 loads per lane, vector loads only
    const int pf = !per_tap && !big && a.vec_in && (Cin % V == 0) && halo_loads <= 4 * 256 && 2 * halo + 2 * (size_t)bn * 128 <= kSoft;
    // weights: at most 4 loads per lane per stage => tg*rows*chunks_per_row <= 1024
    int tg = per_tap ? 1 : 1024 / (rows * chunks_per_row);
    if (tg < 1) tg = 1;
    if (tg > T) tg = T;
    const size_t hb = halo * (pf ? 2 : 1);
    while (tg > 1 && hb + 2 * (size_t)tg * bn * 128 > kSoft) --tg;
    size_t lds = hb + 2 * (size_t)tg * bn * 128;
    if (lds < 4096) lds = 4096;   // stats reduction scratch
    if (lds > kMax) {
      if (big) { big = false; continue; }
      SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd: halo tile of a %dx%d kernel with dilation %d does not fit LDS", kh, kw, dil);
    }
    a.tg = tg; a.pf_halo = pf; a.sh = chunks_per_row == 4 ? 2 : 3; a.per_tap = per_tap;
    if (dtype == SDHIP_BF16)
      return big ? launch_bn<bf16_t, 8, 32>(a, bn, lds, s) : launch_bn<bf16_t, 4, 16>(a, bn, lds, s);
    return big ? launch_bn<float, 8, 32>(a, bn, lds, s) : launch_bn<float, 4, 16>(a, bn, lds, s);
  }
  SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd: no tile configuration fits");
}

// 1x1 convolution over the channel concatenation of up to two tensors, either of which may be a nearest-neighbour
// upsampled map (see include/sdhip.h).  bf16 only: the f32 parity path materialises the concatenation.
extern "C" int sdhip_conv1x1_cat_fwd(const void* x0, int ld0, int c0, int us0, const void* x1, int ld1, int c1, int us1,
                                     const void* wpacked, void* y, int ldy, const float* bias,
                                     int B, int H, int W, int Cout, int act, int dtype, void* stream) {
  SDHIP_CHECK_ARG(x0 && wpacked && y && c0 > 0 && B > 0 && H > 0 && W > 0 && Cout > 0, "conv1x1_cat_fwd: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_BF16, "conv1x1_cat_fwd: bf16 only");
  SDHIP_CHECK_ARG((x1 == nullptr) == (c1 == 0) && us0 >= 0 && us1 >= 0 && us0 < 8 && us1 < 8, "conv1x1_cat_fwd: bad second segment");
  GemmArgs g;
  g.seg[0] = GemmSeg{x0, ld0, c0, us0};
  g.seg[1] = GemmSeg{x1, ld1, c1, us1};
  g.nseg = x1 ? 2 : 1;
  g.wp = wpacked; g.y = y; g.bias = bias; g.in_scale = nullptr; g.in_shift = nullptr; g.stats = nullptr;
  g.ldy = ldy; g.Cout = Cout; g.Mpad = (Cout + 15) & ~15; g.K = c0 + c1; g.in_relu = 0; g.act = act;
  g.M = (long)B * H * W; g.H = H; g.W = W; g.ppg = g.M;
  g.stats_ld = Cout; g.nrep = 1; g.rep_stride = 0;
  if (!gemm1x1_ok(g, 1, true)) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv1x1_cat_fwd: operands not 16-byte aligned / grid not divisible by the upsampling factor");
  return launch_gemm_any(g, (hipStream_t)stream);
}


extern "C" int sdhip_conv2d_fwd(const void* x, const void* wpacked, void* y,
                                const float* bias, const float* in_scale, const float* in_shift,
                                double* stats, int stats_ld, int stats_nrep,
                                int B, int H, int W, int Cin, int ldx,
                                int Ho, int Wo, int Cout, int ldy,
                                int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                                int D, int Do, int kd, int sd, int pad_d,
                                int in_relu, int groups, int act, int accumulate,
                                int dtype, void* stream) {
  return conv2d_fwd_impl(x, wpacked, y, bias, in_scale, in_shift, stats, stats_ld, stats_nrep, B, H, W, Cin, ldx, Ho, Wo, Cout, ldy,
                         kh, kw, stride, dil, pad_t, pad_l, D, Do, kd, sd, pad_d, in_relu, groups, act, accumulate, dtype, stream, 1, 0, 0, 0);
}

// y = conv(relu(BatchNorm_train(x))) with the BatchNorm finalized inside the launch (see include/sdhip.h).
extern "C" int sdhip_conv2d_fwd_bnpro(const void* x, const void* wpacked, void* y, double* out_stats, int out_stats_ld, int out_stats_nrep,
                                      double* in_stats, int in_stats_ld, int in_stats_nrep,
                                      const double* pend_stats, int pend_ld, int pend_nrep, int pend_c0, int pend_n,
                                      const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float* scale_out, float* shift_out, float* mean_out,
                                      float* invstd_out, double count, float eps, float momentum,
                                      int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                                      int kh, int kw, int pad_t, int pad_l, int groups, int dtype, void* stream) {
  SDHIP_CHECK_ARG(in_stats && scale_out && shift_out && mean_out && invstd_out && count >= 1. && in_stats_nrep >= 1 && in_stats_ld >= Cin,
                  "conv2d_fwd_bnpro: statistics / output vectors missing");
  SDHIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "conv2d_fwd_bnpro: running_mean / running_var come together");
  if (Cin > kPinMaxC || in_stats_nrep > 4 || (pend_stats && pend_nrep > 4))
    SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd_bnpro: at most %d input channels and 4 statistics replicas", kPinMaxC);
  SDHIP_CHECK_ARG(!pend_stats || (pend_n > 0 && pend_c0 >= 0 && pend_c0 + pend_n <= Cin && pend_ld >= pend_n && pend_nrep >= 1 && in_stats_nrep == 1),
                  "conv2d_fwd_bnpro: bad pending-statistics slice (folding needs single-replica slab statistics)");
  ProStats ps{in_stats, in_stats_ld, in_stats_nrep, gamma, beta, scale_out, shift_out, mean_out, invstd_out, running_mean, running_var,
              eps, momentum, count, pend_stats, in_stats, pend_ld, pend_nrep, pend_c0, pend_n};
  return conv2d_fwd_impl(x, wpacked, y, nullptr, nullptr, nullptr, out_stats, out_stats_ld, out_stats_nrep, B, H, W, Cin, ldx, Ho, Wo, Cout, ldy,
                         kh, kw, 1, 1, pad_t, pad_l, 1, 1, 1, 1, 0, 1, groups, 0, 0, dtype, stream, 1, 0, 0, 0, nullptr, 0, nullptr, nullptr,
                         nullptr, 0, 0, &ps);
}

// y = conv(x) + addend (see include/sdhip.h).
extern "C" int sdhip_conv2d_fwd_add(const void* x, const void* wpacked, void* y, const void* addend, int ldadd,
                                    int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                                    int kh, int kw, int pad_t, int pad_l, int dtype, void* stream) {
  SDHIP_CHECK_ARG(addend && ldadd >= Cout, "conv2d_fwd_add: addend missing / pixel stride smaller than Cout");
  return conv2d_fwd_impl(x, wpacked, y, nullptr, nullptr, nullptr, nullptr, 0, 1, B, H, W, Cin, ldx, Ho, Wo, Cout, ldy,
                         kh, kw, 1, 1, pad_t, pad_l, 1, 1, 1, 1, 0, 0, 1, 0, 0, dtype, stream, 1, 0, 0, 0, nullptr, 0, nullptr, nullptr, addend, ldadd);
}

// Data gradient of a stride-1 convolution whose input was relu(BatchNorm(u)): y = the gradient w.r.t. that input, and the
// epilogue also takes the two reductions of the BatchNorm backward over it (see include/sdhip.h).
extern "C" int sdhip_conv2d_fwd_bnbwd(const void* x, const void* wpacked, void* y, double* sums, int sums_ld, int sums_nrep,
                                      const void* u, int ldu, const float* scale, const float* shift, const void* addend, int ldadd,
                                      int B, int H, int W, int Cin, int ldx, int Ho, int Wo, int Cout, int ldy,
                                      int kh, int kw, int dil, int pad_t, int pad_l, int groups, int mode, int dtype, void* stream) {
  SDHIP_CHECK_ARG(mode == 0 || (mode == 1 && addend), "conv2d_fwd_bnbwd: mode 1 (apply) needs the tensor the result is added to");
  SDHIP_CHECK_ARG(dtype == SDHIP_BF16, "conv2d_fwd_bnbwd: bf16 only");
  SDHIP_CHECK_ARG(sums && u && scale && shift && ldu >= Cout && Cout % 4 == 0 && ldu % 4 == 0 && ((uintptr_t)u & 7) == 0 &&
                  ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
                  "conv2d_fwd_bnbwd: sums / u / scale / shift missing or misaligned (Cout %% 4, ldu %% 4)");
  SDHIP_CHECK_ARG(!addend || (ldadd >= Cout && ldadd % 4 == 0 && ((uintptr_t)addend & 7) == 0), "conv2d_fwd_bnbwd: addend misaligned");
  return conv2d_fwd_impl(x, wpacked, y, nullptr, nullptr, nullptr, sums, sums_ld, sums_nrep, B, H, W, Cin, ldx, Ho, Wo, Cout, ldy,
                         kh, kw, 1, dil, pad_t, pad_l, 1, 1, 1, 1, 0, 0, groups, 0, 0, dtype, stream, 1, 0, 0, 0, u, ldu, scale, shift,
                         addend, ldadd, mode);
}

// One sub-pixel phase of a stride-2 transposed convolution (see include/sdhip.h): a stride-1 correlation whose outputs are
// written to every second voxel of a volume twice as large in each of depth / height / width.
extern "C" int sdhip_conv2d_fwd_phase(const void* x, const void* wpacked, void* y, double* stats, int stats_ld, int stats_nrep,
                                      int B, int H, int W, int Cin, int ldx, int Cout, int ldy, int kh, int kw,
                                      int D, int kd, int groups, int off_d, int off_h, int off_w, int dtype, void* stream) {
  SDHIP_CHECK_ARG(off_d >= 0 && off_d < 2 && off_h >= 0 && off_h < 2 && off_w >= 0 && off_w < 2, "conv2d_fwd_phase: phase offsets are 0 or 1");
  return conv2d_fwd_impl(x, wpacked, y, nullptr, nullptr, nullptr, stats, stats_ld, stats_nrep, B, H, W, Cin, ldx, H, W, Cout, ldy,
                         kh, kw, 1, 1, 0, 0, D, D, kd, 1, 0, 0, groups, 0, 0, dtype, stream, 2, off_d, off_h, off_w);
}
