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

namespace {

struct FwdArgs {
  const void* x; const void* wp; void* y;
  const float* bias; const float* in_scale; const float* in_shift; double* stats;
  ConvGeom g;
  int Cin, ldx, Cout, Mpad, ldy;
  int in_relu, groups, act, accumulate;
  int tg, vec_in, vec_out, stats_ld, nrep;
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
  const int b = blockIdx.z;
  const int grp = p.groups > 1 ? b / (g.B / p.groups) : 0;
  const int s = g.stride, d = g.dil;
  const int IH = (TH - 1) * s + (g.kh - 1) * d + 1, IW = (TW - 1) * s + (g.kw - 1) * d + 1;
  const int ih0 = oh0 * s - g.pad_t, iw0 = ow0 * s - g.pad_l;
  const int Tn = g.kh * g.kw;
  const int nq = (p.Cin + CK - 1) / CK;
  const int mvalid = min(BN, p.Mpad - n0);  // multiple of 16
  unsigned char* halo = smem;
  unsigned char* wl = smem + ((IH * IW * 128 + 15) & ~15);

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

  const T* xb = (const T*)p.x + (long)b * g.H * g.W * p.ldx;
  const T* wpk = (const T*)p.wp;

  for (int q = 0; q < nq; ++q) {
    const int cin_q = min(CK, p.Cin - q * CK);
    const int nks = (cin_q + CK / 2 - 1) / (CK / 2);  // 64-byte k-steps in this chunk: 1 or 2
    const int sh = nks == 2 ? 3 : 2;                  // chunks per row that carry data: 8 or 4
    __syncthreads();                                  // previous chunk's fragments are consumed
    // ---- stage the halo tile (global -> LDS, fused prologue) ----
    {
      StageSrc ss;
      ss.base = xb; ss.H = g.H; ss.W = g.W; ss.ld = p.ldx; ss.C = p.Cin; ss.h0 = ih0; ss.w0 = iw0; ss.IH = IH; ss.IW = IW;
      ss.scale = p.in_scale ? p.in_scale + grp * p.Cin : nullptr;
      ss.shift = p.in_scale ? p.in_shift + grp * p.Cin : nullptr;
      ss.relu = p.in_relu; ss.vec = p.vec_in;
      stage_tile<T, 4>(halo, ss, q, sh, tid);
    }

    for (int t0 = 0; t0 < Tn; t0 += p.tg) {
      const int nt = min(p.tg, Tn - t0);
      if (t0 > 0) __syncthreads();  // previous tap group's weights are consumed
      // ---- stage the weights of taps [t0, t0+nt) for this channel chunk ----
      const int wtotal = (nt * mvalid) << sh;
      for (int i0 = tid; i0 < wtotal; i0 += 256 * 4) {
        u32x4 raw[4]; int off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = i0 + j * 256;
          const int row = i >> sh, c = i & ((1 << sh) - 1);
          const int tl = row / mvalid, m = row - tl * mvalid;
          off[j] = lds_off(tl * BN + m, c);
          if (i < wtotal) raw[j] = *reinterpret_cast<const u32x4*>(wpk + (((long)(q * Tn + t0 + tl) * p.Mpad + n0 + m) * CK + c * V));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i0 + j * 256 < wtotal) *reinterpret_cast<u32x4*>(wl + off[j]) = raw[j];
      }
      __syncthreads();
      // ---- MFMA over the taps of this group ----
      for (int tl = 0; tl < nt; ++tl) {
        const int t = t0 + tl;
        const int khi = t / g.kw, kwi = t - khi * g.kw;
        const int toff = (khi * d) * IW + kwi * d;
        for (int ks = 0; ks < nks; ++ks) {
          const int c = 4 * ks + lg;
          u32x4 af[NT_CO], bf[NT_PIX];
#pragma unroll
          for (int mi = 0; mi < NT_CO; ++mi)
            if (mi * 16 < mvalid) af[mi] = *reinterpret_cast<const u32x4*>(wl + lds_off(tl * BN + mi * 16 + l15, c));
#pragma unroll
          for (int ni = 0; ni < NT_PIX; ++ni)
            bf[ni] = *reinterpret_cast<const u32x4*>(halo + lds_off(pbase[ni] + toff, c));
#pragma unroll
          for (int mi = 0; mi < NT_CO; ++mi)
            if (mi * 16 < mvalid) {
#pragma unroll
              for (int ni = 0; ni < NT_PIX; ++ni) Mma<T>::run(acc[mi][ni], af[mi], bf[ni]);
            }
        }
      }
    }
  }

  // ---- epilogue: bias / activation / accumulate / store / BN statistics ----
  float s1[NT_CO][4], s2[NT_CO][4];
#pragma unroll
  for (int mi = 0; mi < NT_CO; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[mi][r] = 0.f; s2[mi][r] = 0.f; }

  T* yb = (T*)p.y + (long)b * g.Ho * g.Wo * p.ldy;
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
  dim3 grid(sdhip_cdiv(g.Ho, TH) * sdhip_cdiv(g.Wo, TW), sdhip_cdiv(a.Mpad, BN), g.B);
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

size_t lds_need(const ConvGeom& g, int th, int tw, int rows_per_tap, int tg) {
  const int IH = (th - 1) * g.stride + (g.kh - 1) * g.dil + 1, IW = (tw - 1) * g.stride + (g.kw - 1) * g.dil + 1;
  const size_t halo = ((size_t)IH * IW * 128 + 15) & ~(size_t)15;
  const size_t need = halo + (size_t)tg * rows_per_tap * 128;
  return need < 4096 ? 4096 : need;  // stats reduction scratch
}

}  // namespace

extern "C" int sdhip_conv2d_fwd(const void* x, const void* wpacked, void* y,
                                const float* bias, const float* in_scale, const float* in_shift,
                                double* stats, int stats_ld, int stats_nrep,
                                int B, int H, int W, int Cin, int ldx,
                                int Ho, int Wo, int Cout, int ldy,
                                int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                                int in_relu, int groups, int act, int accumulate,
                                int dtype, void* stream) {
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
  a.g = ConvGeom{B, H, W, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l};
  a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.Mpad = (Cout + 15) & ~15; a.ldy = ldy;
  a.in_relu = in_relu; a.groups = groups; a.act = act; a.accumulate = accumulate;
  a.stats_ld = stats_ld > 0 ? stats_ld : Cout;
  a.nrep = stats_nrep > 0 ? stats_nrep : 1;
  a.rep_stride = (long)groups * 2 * a.stats_ld;
  a.vec_in = (Cin % V == 0) && (ldx % V == 0) && (((uintptr_t)x & 15) == 0);
  a.vec_out = (ldy % 4 == 0) && (((uintptr_t)y % (4 * es)) == 0);
  int bn = a.Mpad > 64 ? 128 : (a.Mpad > 32 ? 64 : (a.Mpad > 16 ? 32 : 16));
  const int T = kh * kw;
  const long blocks_big = (long)sdhip_cdiv(Ho, 8) * sdhip_cdiv(Wo, 32) * B * sdhip_cdiv(a.Mpad, bn);
  bool big = blocks_big >= 512 && Wo >= 24;
  if (!big) {
    // small feature maps (DenseNet blocks 2-4, pooled pyramids): the launch cannot fill 256 CUs with pixel tiles alone,
    // so split the output channels over more workgroups (the input tile is re-read from L2, the serial
    // chunk-by-chunk latency chain per workgroup gets shorter and more of them overlap per CU).
    const long px_blocks = (long)sdhip_cdiv(Ho, 4) * sdhip_cdiv(Wo, 16) * B;
    while (bn > 32 && px_blocks * sdhip_cdiv(a.Mpad, bn) < 1024) bn >>= 1;
  }
  const int rows = a.Mpad < bn ? a.Mpad : bn;
  const size_t kMax = 160 * 1024, kSoft = 64 * 1024;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int th = big ? 8 : 4, tw = big ? 32 : 16;
    int tg = T;
    if (lds_need(a.g, th, tw, rows, tg) > kSoft) {
      const size_t base = lds_need(a.g, th, tw, rows, 0);
      tg = base < kSoft ? (int)((kSoft - base) / ((size_t)rows * 128)) : 1;
      if (tg < 1) tg = 1;
      if (tg > T) tg = T;
    }
    const size_t lds = lds_need(a.g, th, tw, rows, tg);
    if (lds > kMax) {
      if (big) { big = false; continue; }
      SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd: halo tile of a %dx%d kernel with dilation %d does not fit LDS", kh, kw, dil);
    }
    a.tg = tg;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SDHIP_BF16)
      return big ? launch_bn<bf16_t, 8, 32>(a, bn, lds, s) : launch_bn<bf16_t, 4, 16>(a, bn, lds, s);
    return big ? launch_bn<float, 8, 32>(a, bn, lds, s) : launch_bn<float, 4, 16>(a, bn, lds, s);
  }
  SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_fwd: no tile configuration fits");
}
