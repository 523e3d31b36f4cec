// Direct 2-D convolution, weight-gradient form, for gfx950 (MI355X).
//
//   dW[t][m][k] = sum over (b, oh, ow) of dY[b,oh,ow,m] * Xeff[b, oh*s + kh*d - pad_t, ow*s + kw*d - pad_l, k]
//
// (Xeff = X after the optional fused input prologue.)  This is the backward of
// conv2dSame / ConvTranspose2dSame / nn.Conv2d of the reference
// (models/torch_model.py:236-349, models/densenet.py:25-93) w.r.t. their weights.
//
// The contraction runs over PIXELS, so both MFMA operands must be read
// pixel-major from channel-major (NHWC) tiles.  The tiles are staged in LDS
// exactly as in the forward kernel ([pixel][channel], 128-byte rows) and the
// fragments are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16),
// so no transposed copy of the activations is ever made; tap shifts are plain
// row offsets into the shared halo tile.  A workgroup owns (output-channel block,
// input-channel chunk, <= 9 taps), keeps those partial sums in registers while it
// sweeps a strided share of all pixel tiles, and flushes once with f32 atomics
// into the packed f32 gradient buffer.
#include "conv_common.h"
#include "conv_wgrad_fast.h"
#include "conv_wgrad_half.h"
#include "conv_wgrad_single.h"
#include "conv_thin.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int kThreads = 512;  // 8 waves: 2 per SIMD, up to 256 VGPRs each

struct WgArgs {
  const void* x; const void* dy; float* dwp; float* dbias;
  const float* in_scale; const float* in_shift;
  ConvGeom g;
  int Cin, ldx, Cout, Mpad, lddy;
  int in_relu, groups;
  int tpb, ntg, nq;  // taps per workgroup, tap groups, channel chunks
  int vec_x, vec_dy;
};

typedef __attribute__((address_space(3))) bf16x4_t* lds_bf4_ptr;

// 16x16x32 bf16 operand fragment, pixel-major, from a [pixel][channel] LDS image (ds_read_b64_tr_b16 x2).
// row0/row1: the LDS rows (pixels) this lane addresses, i.e. pixels 8g + ((l&15)>>2) and +4; `ch` the first
// channel of the 16-channel tile.  EXEC must be full (the read crosses lanes).
__device__ __forceinline__ u32x4 tr_frag(unsigned char* base, int row0, int row1, int ch, int lane) {
  const int p = lane & 3;                      // 4-channel sub-block this lane addresses
  const int c = (ch >> 3) + (p >> 1);          // 16-byte chunk
  const int sub = (p & 1) << 3;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(base + lds_off(row0, c) + sub));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4_ptr)(base + lds_off(row1, c) + sub));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// One workgroup = (output-channel block of CK, input-channel chunk of CK, a group of <= MAXT taps).  Its 8 waves
// split the CK x CK x taps accumulator: wave -> (16 input channels, half of the output channels[, tap parity]),
// so that ALL taps of a 5x5 kernel (25 x 64 x 64 f32 partial sums = 400 KB) stay in registers while the workgroup
// sweeps its share of the pixel tiles: X and dY are staged once per tile, not once per tap group.
template <typename T, int TH, int TW, int MAXT, int MB>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(const WgArgs p) {
  constexpr int V = Chunk<T>::N;
  constexpr int CK = 8 * V;                 // channels per LDS row = input-channel chunk
  constexpr bool BF = sizeof(T) == 2;
  constexpr int NTILE = CK / 16;            // 16-channel MFMA tiles per input chunk: 4 (bf16) / 2 (f32)
  constexpr int NW = kThreads / 64;
  constexpr int NREST = NW / NTILE;         // waves per input-channel tile: 2 (bf16) / 4 (f32)
  constexpr int NCO = BF ? 2 : 1;           // output-channel tiles per wave
  constexpr int COG = (MB / 16) / NCO;      // output-channel groups across waves
  constexpr int TAPL = NREST / COG;         // waves sharing a (ci, co) block, interleaved over the taps
  constexpr int MAXTW = (MAXT + TAPL - 1) / TAPL;
  static_assert(COG >= 1 && TAPL >= 1 && COG * TAPL == NREST && MB <= CK, "wave decomposition");
  constexpr int KSTEP = BF ? 32 : 4;        // pixels per MFMA k-step
  static_assert((TH * TW) % 32 == 0, "tile must be whole k-steps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const ConvGeom& g = p.g;
  const int s = g.stride, d = g.dil;
  const int IH = (TH - 1) * s + (g.kh - 1) * d + 1, IW = (TW - 1) * s + (g.kw - 1) * d + 1;
  const int T_ = g.kh * g.kw;
  const unsigned magic_iw = div_magic(IW);
  int by = blockIdx.y;                       // -> (output-channel block, depth tap, channel chunk, tap group)
  const int tgi = by % p.ntg; by /= p.ntg;
  const int q = by % p.nq; by /= p.nq;
  const int kdi = by % g.kd;
  const int mb = by / g.kd;
  const int t0 = tgi * p.tpb, nt = min(p.tpb, T_ - t0);
  const int m0 = mb * MB;
  const int ci_tile = wave % NTILE;
  const int rest = wave / NTILE;
  const int co_tile0 = (rest % COG) * NCO;
  const int tap_lane = rest / COG;
  const int cin_q = min(CK, p.Cin - q * CK);
  const int cout_m = min(MB, p.Cout - m0);
  const int shx = (cin_q + CK / 2 - 1) / (CK / 2) == 2 ? 3 : 2;   // data chunks per X row: 8 or 4
  const int shy = (cout_m + CK / 2 - 1) / (CK / 2) == 2 ? 3 : 2;

  unsigned char* halo = smem;
  unsigned char* ytile = smem + ((IH * IW * 128 + 15) & ~15);

  f32x4 acc[MAXTW][NCO];
#pragma unroll
  for (int t = 0; t < MAXTW; ++t)
#pragma unroll
    for (int mi = 0; mi < NCO; ++mi) acc[t][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;  // bias gradient partial of channel (tid % CK), pixel stripe (tid / CK)

  const int tiles_w = (g.Wo + TW - 1) / TW, tiles_h = (g.Ho + TH - 1) / TH;
  const int ntiles = g.B * g.Do * tiles_h * tiles_w;   // "images" = (batch, output depth slice)
  const bool wave_active = ci_tile * 16 < cin_q && co_tile0 * 16 < cout_m;  // wave-uniform

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int img = tile / (tiles_h * tiles_w);
    const int tr = tile - img * tiles_h * tiles_w;
    const int oh0 = (tr / tiles_w) * TH, ow0 = (tr % tiles_w) * TW;
    const int b = img / g.Do, dz = img - b * g.Do;
    const int din = dz * g.sd + kdi - g.pad_d;       // input depth slice this depth tap reads
    if (din < 0 || din >= g.D) continue;             // zero padding in depth: no contribution (uniform per workgroup)
    const int grp = p.groups > 1 ? b / (g.B / p.groups) : 0;
    const T* xb = (const T*)p.x + ((long)b * g.D + din) * g.H * g.W * p.ldx;
    const T* yb = (const T*)p.dy + (long)img * g.Ho * g.Wo * p.lddy;
    __syncthreads();  // previous tile's fragments are consumed
    {   // X halo tile of channel chunk q (same LDS image as the forward kernel, fused prologue)
      StageSrc ss;
      ss.base = xb; ss.H = g.H; ss.W = g.W; ss.ld = p.ldx; ss.C = p.Cin;
      ss.h0 = oh0 * s - g.pad_t; ss.w0 = ow0 * s - g.pad_l; ss.IH = IH; ss.IW = IW; ss.magic_iw = magic_iw;
      ss.scale = p.in_scale ? p.in_scale + grp * p.Cin : nullptr;
      ss.shift = p.in_scale ? p.in_shift + grp * p.Cin : nullptr;
      ss.relu = p.in_relu; ss.vec = p.vec_x;
      stage_tile<T, 4, kThreads>(halo, ss, q, shx, tid);
    }
    {   // dY tile of output-channel block mb (TH x TW, no halo; channels offset by m0)
      StageSrc ss;
      ss.base = yb + m0; ss.H = g.Ho; ss.W = g.Wo; ss.ld = p.lddy; ss.C = p.Cout - m0;
      ss.h0 = oh0; ss.w0 = ow0; ss.IH = TH; ss.IW = TW; ss.magic_iw = div_magic(TW);
      ss.scale = nullptr; ss.shift = nullptr; ss.relu = 0;
      ss.vec = p.vec_dy;
      stage_tile<T, 2, kThreads>(ytile, ss, 0, shy, tid);
    }
    __syncthreads();

    if (p.dbias && q == 0 && tgi == 0 && kdi == g.pad_d) {  // uniform: bias gradient = column sums of the dY tile (once)
      const int ch = tid % CK, stripe = tid / CK;
      if (ch < cout_m) {
        const int c = ch / V, e = ch % V;
        for (int pix = stripe; pix < TH * TW; pix += kThreads / CK)
          bsum += Elem<T>::ld(reinterpret_cast<const T*>(ytile + lds_off(pix, c)) + e);
      }
    }

    // ---- MFMA over the tile's pixels (the contraction index) ----
    for (int k0 = 0; k0 < TH * TW; k0 += KSTEP) {
      if constexpr (BF) {
        const int pa = k0 + 8 * lg + (l15 >> 2);   // this lane addresses pixels pa and pa + 4
        const int pb = pa + 4;
        u32x4 af[NCO];
#pragma unroll
        for (int mi = 0; mi < NCO; ++mi) af[mi] = tr_frag(ytile, pa, pb, (co_tile0 + mi) * 16, lane);
        const int ha = ((pa / TW) * s) * IW + (pa % TW) * s;
        const int hb = ((pb / TW) * s) * IW + (pb % TW) * s;
#pragma unroll
        for (int tl = 0; tl < MAXTW; ++tl) {
          const int tt = tl * TAPL + tap_lane;
          if (tt < nt) {   // wave-uniform
            const int t = t0 + tt;
            const int khi = t / g.kw, kwi = t - khi * g.kw;
            const int toff = (khi * d) * IW + kwi * d;
            const u32x4 bf = tr_frag(halo, ha + toff, hb + toff, ci_tile * 16, lane);
#pragma unroll
            for (int mi = 0; mi < NCO; ++mi) Mma<T>::run(acc[tl][mi], af[mi], bf);
          }
        }
      } else {
        // f32: v_mfma_f32_16x16x4_f32, one float per lane per operand: A[m = l15][k = lg], B[k = lg][n = l15]
        const int pk = k0 + lg;
        const int hk = ((pk / TW) * s) * IW + (pk % TW) * s;
        float af[NCO];
#pragma unroll
        for (int mi = 0; mi < NCO; ++mi) {
          const int ch = (co_tile0 + mi) * 16 + l15;
          af[mi] = *reinterpret_cast<const float*>(ytile + lds_off(pk, ch >> 2) + ((ch & 3) << 2));
        }
        const int chx = ci_tile * 16 + l15;
#pragma unroll
        for (int tl = 0; tl < MAXTW; ++tl) {
          const int tt = tl * TAPL + tap_lane;
          if (tt < nt) {
            const int t = t0 + tt;
            const int khi = t / g.kw, kwi = t - khi * g.kw;
            const int row = hk + (khi * d) * IW + kwi * d;
            const float bv = *reinterpret_cast<const float*>(halo + lds_off(row, chx >> 2) + ((chx & 3) << 2));
#pragma unroll
            for (int mi = 0; mi < NCO; ++mi)
              acc[tl][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bv, acc[tl][mi], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- flush: f32 atomics into the packed gradient buffer [nq][T][Mpad][CK] ----
  if (wave_active) {
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
  if (p.dbias && q == 0 && tgi == 0 && kdi == g.pad_d) {
    const int ch = tid % CK;
    if (ch < cout_m) atomicAdd(p.dbias + m0 + ch, bsum);
  }
}

template <typename T, int TH, int TW, int MAXT, int MB>
int launch(const WgArgs& a, hipStream_t s) {
  auto kern = conv_wgrad_kernel<T, TH, TW, MAXT, MB>;
  const ConvGeom& g = a.g;
  const int IH = (TH - 1) * g.stride + (g.kh - 1) * g.dil + 1, IW = (TW - 1) * g.stride + (g.kw - 1) * g.dil + 1;
  const size_t lds = (((size_t)IH * IW * 128 + 15) & ~(size_t)15) + (size_t)TH * TW * 128;
  if (lds > 160 * 1024) return 1;  // caller falls back to a smaller tile
  static size_t attr_set = 0;
  if (lds > 64 * 1024 && attr_set == 0) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: cannot raise dynamic LDS limit");
    attr_set = 1;
  }
  const int nmb = sdhip_cdiv(a.Cout, MB);
  const int gy = nmb * g.kd * a.nq * a.ntg;
  const int ntiles = g.B * g.Do * sdhip_cdiv(g.Ho, TH) * sdhip_cdiv(g.Wo, TW);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  int gx = sdhip_cdiv(256 * per_cu, gy);
  if (gx > ntiles) gx = ntiles;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(kThreads), lds, s, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

template <typename T, int MAXT, int MB>
int launch_tile(const WgArgs& a, bool wide, hipStream_t s) {
  if (wide) {
    const int rc = launch<T, 8, 32, MAXT, MB>(a, s);
    if (rc != 1) return rc;
    const int rc2 = launch<T, 4, 32, MAXT, MB>(a, s);
    if (rc2 != 1) return rc2;
  }
  const int rc = launch<T, 4, 16, MAXT, MB>(a, s);
  if (rc == 1) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "conv2d_wgrad: halo tile does not fit LDS (k=%dx%d dil=%d)", a.g.kh, a.g.kw, a.g.dil);
  return rc;
}

// ---- weight (un)packing -------------------------------------------------------
// dst[q][t][m][c] = src[m*sm + (q*CK+c)*sk + (flip ? T-1-t : t)]   (zero padded)
template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int M, int Mpad, int K, int Tn,
                            long sm, long sk, int flip, long total) {
  constexpr int CK = 8 * Chunk<T>::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CK);
    long r = i / CK;
    const int m = (int)(r % Mpad); r /= Mpad;
    const int t = (int)(r % Tn);
    const int q = (int)(r / Tn);
    const int k = q * CK + c;
    float v = 0.f;
    if (m < M && k < K) v = src[m * sm + k * sk + (flip ? Tn - 1 - t : t)];
    Elem<T>::st(dst + i, v);
  }
}

// grad[m*sm + k*sk + tt] (+)= acc[q][t][m][c]
__global__ void unpack_kernel(const float* __restrict__ acc, float* __restrict__ grad, int M, int Mpad, int K, int Tn,
                              int CK, long sm, long sk, int flip, int accumulate, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // iterate over the destination so writes are coalesced: i = (m*K + k)*T + t for the natural (M,K,T) order
    const int t = (int)(i % Tn);
    long r = i / Tn;
    const int k = (int)(r % K);
    const int m = (int)(r / K);
    const int tt = flip ? Tn - 1 - t : t;
    const float v = acc[(((long)(k / CK) * Tn + tt) * Mpad + m) * CK + (k % CK)];
    float* dst = grad + m * sm + k * sk + t;
    *dst = accumulate ? *dst + v : v;
  }
}

}  // namespace

extern "C" long sdhip_conv_packed_elems(int M, int K, int T, int dtype) {
  const int ck = conv_ck(dtype);
  return (long)((K + ck - 1) / ck) * T * ((M + 15) & ~15) * ck;
}

extern "C" int sdhip_conv_pack_weights(const float* src, void* dst, int M, int K, int T,
                                       long stride_m, long stride_k, int flip, int dtype, void* stream) {
  SDHIP_CHECK_ARG(src && dst && M > 0 && K > 0 && T > 0, "conv_pack_weights: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "conv_pack_weights: unknown dtype %d", dtype);
  const long total = sdhip_conv_packed_elems(M, K, T, dtype);
  const int Mpad = (M + 15) & ~15;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SDHIP_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, M, Mpad, K, T, stride_m, stride_k, flip, total);
  else
    hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, M, Mpad, K, T, stride_m, stride_k, flip, total);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// One launch packs every weight of the network: desc[i] = {src ptr, dst ptr, M, K, T, stride_m, stride_k, flip}.
// Workgroups per descriptor: the launch lasts as long as its LARGEST layer (DenseNet's 1024 -> 512 transition, 524 K
// elements: 128 gather trips per lane with 16 workgroups, 113 us); surplus workgroups of small layers exit at once.
constexpr int kBatchBlocks = 64;
template <typename T>
__global__ void pack_batch_kernel(const long* __restrict__ desc) {
  constexpr int CK = 8 * Chunk<T>::N;
  const long* d = desc + (long)blockIdx.y * 8;
  const float* src = reinterpret_cast<const float*>(d[0]);
  T* dst = reinterpret_cast<T*>(d[1]);
  const int M = (int)d[2], K = (int)d[3], Tn = (int)d[4], flip = (int)d[7];
  const long sm = d[5], sk = d[6];
  const int Mpad = (M + 15) & ~15;
  // 32-bit index arithmetic throughout (a layer's packed image has at most a few million elements): the 64-bit divisions of the first version were most of this kernel's 120 us
  const unsigned total = (unsigned)((K + CK - 1) / CK) * Tn * Mpad * CK;
  const unsigned per_q = (unsigned)Tn * Mpad;                 // (tap, m) rows per channel chunk
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned c = i % CK;                                 // CK is a power of two
    const unsigned r = i / CK;
    const unsigned q = r / per_q, rr = r - q * per_q;
    const unsigned t = rr / Mpad, m = rr - t * Mpad;
    const unsigned k = q * CK + c;
    float v = 0.f;
    if (m < (unsigned)M && k < (unsigned)K) v = src[(long)m * sm + (long)k * sk + (flip ? Tn - 1 - (int)t : (int)t)];
    Elem<T>::st(dst + i, v);
  }
}

extern "C" int sdhip_conv_pack_batch(const long* desc, int ndesc, int dtype, void* stream) {
  SDHIP_CHECK_ARG(desc && ndesc > 0, "conv_pack_batch: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "conv_pack_batch: unknown dtype %d", dtype);
  if (dtype == SDHIP_BF16) hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3(kBatchBlocks, ndesc), dim3(256), 0, (hipStream_t)stream, desc);
  else hipLaunchKernelGGL(pack_batch_kernel<float>, dim3(kBatchBlocks, ndesc), dim3(256), 0, (hipStream_t)stream, desc);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// desc[i] = {acc ptr, grad ptr, M, K, T, stride_m, stride_k, flip}: grad(m,k,t) += acc[k/CK][tt][m][k%CK]
__global__ void unpack_batch_kernel(const long* __restrict__ desc, int CK) {
  const long* d = desc + (long)blockIdx.y * 8;
  const float* acc = reinterpret_cast<const float*>(d[0]);
  float* grad = reinterpret_cast<float*>(d[1]);
  const int M = (int)d[2], K = (int)d[3], Tn = (int)d[4], flip = (int)d[7];
  const long sm = d[5], sk = d[6];
  const int Mpad = (M + 15) & ~15;
  const unsigned total = (unsigned)M * K * Tn;                 // 32-bit index arithmetic (see pack_batch_kernel)
  const unsigned ckshift = CK == 64 ? 6 : 5;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / Tn, t = i - r * Tn;
    const unsigned m = r / K, k = r - m * K;
    const unsigned tt = flip ? Tn - 1 - t : t;
    grad[(long)m * sm + (long)k * sk + t] += acc[(((long)(k >> ckshift) * Tn + tt) * Mpad + m) * CK + (k & (CK - 1))];
  }
}

extern "C" int sdhip_conv_unpack_batch(const long* desc, int ndesc, int dtype, void* stream) {
  SDHIP_CHECK_ARG(desc && ndesc > 0, "conv_unpack_batch: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "conv_unpack_batch: unknown dtype %d", dtype);
  hipLaunchKernelGGL(unpack_batch_kernel, dim3(kBatchBlocks, ndesc), dim3(256), 0, (hipStream_t)stream, desc, conv_ck(dtype));
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_conv_unpack_wgrad(const float* acc, float* grad, int M, int K, int T,
                                       long stride_m, long stride_k, int flip, int accumulate, int dtype, void* stream) {
  SDHIP_CHECK_ARG(acc && grad && M > 0 && K > 0 && T > 0, "conv_unpack_wgrad: bad arguments");
  const long total = (long)M * K * T;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, grad, M, (M + 15) & ~15, K, T,
                     conv_ck(dtype), stride_m, stride_k, flip, accumulate, total);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// One layer.  plan == nullptr: launch now.  plan != nullptr: a layer of the bf16 fast path is only PLANNED (*planned = true,
// nothing launched) so that the caller can group it with others; every other path launches as usual.
static int wgrad_one(const void* x, const void* dy, float* dw_packed, float* dbias,
                     const float* in_scale, const float* in_shift,
                     int B, int H, int W, int Cin, int ldx,
                     int Ho, int Wo, int Cout, int lddy,
                     int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                     int D, int Do, int kd, int sd, int pad_d,
                     int in_relu, int groups, int prezeroed, int dtype, void* stream, WgfPlan* plan, bool* planned) {
  SDHIP_CHECK_ARG(x && dy && dw_packed, "conv2d_wgrad: null pointer");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "conv2d_wgrad: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0, "conv2d_wgrad: empty tensor");
  SDHIP_CHECK_ARG(ldx >= Cin && lddy >= Cout, "conv2d_wgrad: pixel stride smaller than channel count");
  SDHIP_CHECK_ARG(groups >= 1 && B % groups == 0, "conv2d_wgrad: batch %d not divisible by %d stat groups", B, groups);
  SDHIP_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv2d_wgrad: in_scale/in_shift must come together");
  const int V = dtype == SDHIP_BF16 ? 8 : 4, CK = 8 * V, T = kh * kw;
  WgArgs a;
  a.x = x; a.dy = dy; a.dwp = dw_packed; a.dbias = dbias; a.in_scale = in_scale; a.in_shift = in_shift;
  SDHIP_CHECK_ARG(D >= 1 && Do >= 1 && kd >= 1 && sd >= 1, "conv2d_wgrad: bad depth geometry");
  a.g = ConvGeom{B, H, W, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l, D, Do, kd, sd, pad_d};
  a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.Mpad = (Cout + 15) & ~15; a.lddy = lddy;
  a.in_relu = in_relu; a.groups = groups;
  a.nq = sdhip_cdiv(Cin, CK);
  const int maxt = T <= 9 ? 9 : 25;          // taps a workgroup keeps in registers
  a.ntg = sdhip_cdiv(T, maxt);
  a.tpb = sdhip_cdiv(T, a.ntg);              // balanced tap groups (49 -> 25, 24)
  a.ntg = sdhip_cdiv(T, a.tpb);
  a.vec_x = (ldx % V == 0) && (((uintptr_t)x & 15) == 0);      // channel tails are masked in stage_tile
  a.vec_dy = (lddy % V == 0) && (((uintptr_t)dy & 15) == 0);
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)kd * sdhip_conv_packed_elems(Cout, Cin, T, dtype);
  if (!prezeroed) {
    if (sdhip_zero_async(dw_packed, n * sizeof(float), s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: memset failed");
    if (dbias && sdhip_zero_async(dbias, (size_t)Cout * sizeof(float), s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad: memset failed");
  }
  // ---- thin path (conv_thin.h): <= 8 input channels -> 1 output channel ----
  if (Cout == 1 && Cin <= V && a.vec_x && stride == 1 && kd == 1 && D == 1 && Do == 1 && !in_scale && T <= 25 &&
      Ho == H + 2 * pad_t - dil * (kh - 1) && Wo == W + 2 * pad_l - dil * (kw - 1) && pad_t >= 0 && pad_l >= 0 &&
      !sdhip_diag().conv_no_thin) {
    ThinWgArgs t;
    t.x = x; t.dy = dy; t.dwp = dw_packed; t.dbias = dbias;
    t.B = B; t.H = H; t.W = W; t.Ho = Ho; t.Wo = Wo; t.kh = kh; t.kw = kw; t.dil = dil; t.pad_t = pad_t; t.pad_l = pad_l;
    t.Cin = Cin; t.ldx = ldx; t.lddy = lddy; t.Mpad = a.Mpad;
    const long npix = (long)B * Ho * Wo;
    // large maps, bf16: the LDS-tiled kernel (lanes = (tap, channel pair), 8 x 64 pixel tiles)
    {
      const int IHt = 8 + (kh - 1) * dil, IWt = 64 + (kw - 1) * dil, P = IWt + 2;
      const size_t lds = (size_t)IHt * P * 16 + 8 * 64 * 2;
      const bool no_tiled = sdhip_diag().thin_wgrad_reg;   // diagnostics: A/B against the register kernel
      if (dtype == SDHIP_BF16 && T > 9 && 8 * T <= 256 && lds <= 60 * 1024 && Wo >= 64 && Ho >= 8 && npix >= 65536 && !no_tiled) {
        const int th = sdhip_cdiv(Ho, 8), tw = sdhip_cdiv(Wo, 64);
        const int ntiles = B * th * tw;
        // measured at 8 x 256 x 512 (tools/gpu_thinwg.py): 128 / 256 / 512 / 1024 / 2048 workgroups -> 96 / 51 / 33 / 33 / 38 us:
        // the single-buffered tiles want several workgroups per CU; past 1024 the T x 8 flush atomics per workgroup show
        const int cap = sdhip_diag().tune_thin_blocks;
        const int blocks = ntiles < cap ? ntiles : cap;
        hipLaunchKernelGGL((conv_thin_wgrad_tiled_kernel<25>), dim3(blocks), dim3(256), lds, s, t, th, tw, IHt, IWt, P);
        SDHIP_LAUNCH_CHECK();
        return SDHIP_OK;
      }
    }
    const int blocks = (int)(npix / 256 < 1024 ? (npix + 255) / 256 : 1024);   // measured: 128 / 1024 workgroups -> 150 / 95 us
    if (dtype == SDHIP_BF16) {
      if (T <= 9) hipLaunchKernelGGL((conv_thin_wgrad_kernel<bf16_t, 9>), dim3(blocks), dim3(256), 0, s, t);
      else hipLaunchKernelGGL((conv_thin_wgrad_kernel<bf16_t, 25>), dim3(blocks), dim3(256), 0, s, t);
    } else {
      if (T <= 9) hipLaunchKernelGGL((conv_thin_wgrad_kernel<float, 9>), dim3(blocks), dim3(256), 0, s, t);
      else hipLaunchKernelGGL((conv_thin_wgrad_kernel<float, 25>), dim3(blocks), dim3(256), 0, s, t);
    }
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  // ---- one output map, 9..32 input channels, <= 32 taps (conv_wgrad_single.h): taps as the MFMA row axis ----
  if (dtype == SDHIP_BF16 && single_wgrad_ok(Cin, Cout, T, stride, dil, kd, sd, ldx, x) && !in_scale && !dbias && !sdhip_diag().conv_no_thin) {
    SingleWgArgs t;
    t.x = x; t.dy = dy; t.dwp = dw_packed;
    t.B = B; t.H = H; t.W = W; t.Ho = Ho; t.Wo = Wo; t.kh = kh; t.kw = kw; t.pad_t = pad_t; t.pad_l = pad_l;
    t.D = D; t.Do = Do; t.kd = kd; t.pad_d = pad_d;
    t.Cin = Cin; t.ldx = ldx; t.lddy = lddy; t.Mpad = a.Mpad; t.dpw = 1; t.zsegs = D;
    const int rc = launch_single_wgrad(t, s);
    if (rc != 1) return rc;                     // 1: halo does not fit -> the kernels below
  }
  // ---- bf16 fast path (conv_wgrad_fast.h): 16-byte-aligned pixels on both operands ----
  if (dtype == SDHIP_BF16 && a.vec_x && a.vec_dy && (long)H * W * ldx < (1L << 31) && (long)Ho * Wo * lddy < (1L << 31) &&
      !sdhip_diag().wgrad_generic) {
    WgfArgs f;
    f.x = x; f.dy = dy; f.dwp = dw_packed; f.dbias = dbias; f.in_scale = in_scale; f.in_shift = in_shift;
    f.B = B; f.H = H; f.W = W; f.Ho = Ho; f.Wo = Wo; f.kh = kh; f.kw = kw; f.stride = stride; f.dil = dil; f.pad_t = pad_t; f.pad_l = pad_l;
    f.D = D; f.Do = Do; f.kd = kd; f.sd = sd; f.pad_d = pad_d;
    f.Cin = Cin; f.ldx = ldx; f.Cout = Cout; f.Mpad = a.Mpad; f.lddy = lddy;
    f.in_relu = in_relu; f.bpg = B / groups;
    f.tpb = a.tpb; f.ntg = a.ntg; f.nq = a.nq;
    if (T == 1 && a.tpb == 1) { f.tpb = 1; f.ntg = 1; }
    f.qb = 0; f.qsh = 0; f.nq_tot = a.nq;
    // <= 32 channels on both sides, 3x3 (x3), stride 1: 64-byte LDS rows, all depth taps per workgroup (conv_wgrad_half.h)
    if (Cin <= 32 && Cout <= 32 && kh == 3 && kw == 3 && stride == 1 && dil == 1 && sd == 1 && (kd == 1 || kd == 3) && !in_scale && !dbias &&
        Wo >= 24 && !sdhip_diag().wgrad_no_half32) {
      WgfPlan pl;
      const int rc = kd == 3 ? plan_wg32<3>(f, pl) : plan_wg32<1>(f, pl);
      if (rc != SDHIP_OK) return rc;
      if (plan) { *plan = pl; *planned = true; return SDHIP_OK; }
      return launch_wgf_plan(pl, s);
    }
    // 1x1 with many input channels (DenseNet bottlenecks / transitions): pack 2 or 4 channel chunks per workgroup
    const bool no_pack = sdhip_diag().wgrad_no_pack;   // diagnostics: A/B against the unpacked kernel
    // Measured (tools/gpu_wgrad1x1_ab.sh): it pays on large maps and wide outputs (192->128 at 16x64x128: 88 -> 52 us,
    // 1024->512 at 16x16x32: 62 -> 54 us); on the small maps of the deep DenseNet blocks the sweep is bound by the
    // per-tile DMA latency either way and the unpacked kernel's smaller tiles win (1024->128 at 16x16x32: 21 vs 31 us).
    const bool pack_pays = Cout >= 256 || (long)B * H * W >= 100000;
    if (T == 1 && kd == 1 && stride == 1 && pad_t == 0 && pad_l == 0 && Ho == H && Wo == W && a.nq >= 2 && Cout > 32 && !no_pack &&
        (pack_pays || sdhip_diag().wgrad_force_pack)) {
      WgfArgs g = f;
      g.qb = a.nq >= 3 ? 4 : 2; g.qsh = g.qb == 4 ? 2 : 1;
      g.kh = 1; g.kw = g.qb; g.tpb = g.qb; g.ntg = 1; g.nq = sdhip_cdiv(a.nq, g.qb);
      const int rc = in_scale ? launch_wgf_packed<false>(g, s, plan) : launch_wgf_packed<true>(g, s, plan);
      if (rc != 1) { if (plan && rc == SDHIP_OK) *planned = true; return rc; }
    }
    // MB = 32 regroups the taps over two wave groups: the tap grouping must match the instantiation (see launch_wgf_taps)
    const int rc = in_scale ? launch_wgf_taps<false>(f, T, s, plan) : launch_wgf_taps<true>(f, T, s, plan);
    if (rc != 1) { if (plan && rc == SDHIP_OK) *planned = true; return rc; }   // 1: no tile fits LDS -> general kernel below
  }
  const bool wide = Wo >= 24;
  // bf16, 25 taps: 25 x (64 x 64) partial sums do not fit 8 waves' registers -> 32 output channels per workgroup
  if (dtype == SDHIP_BF16) return maxt == 9 ? launch_tile<bf16_t, 9, 64>(a, wide, s) : launch_tile<bf16_t, 25, 32>(a, wide, s);
  return maxt == 9 ? launch_tile<float, 9, 32>(a, wide, s) : launch_tile<float, 25, 32>(a, wide, s);
}

// One layer, possibly as two half layers: 64 input channels into <= 32 output channels, 3x3 (x3), stride 1 (dres0[0] of PSMNet,
// models_psmnet/stackhourglass.py:59: the 64-channel cost volume into 32) is the sum of two 32-channel layers over channel
// halves of the same pixels — x + 32 elements, the same pixel stride — whose gradients are the column halves [0,32) / [32,64)
// of the same packed rows [kd][1][9][Mpad][64]: both halves run on the 64-byte-row kernel (conv_wgrad_half.h) instead of the
// 128-byte-row one (2.04 -> 2 x 0.3 ms at 4 x 48 x 128 x 240).  plans == nullptr: launch now; otherwise fast-path layers are planned.
static int wgrad_layer(const void* x, const void* dy, float* dw_packed, float* dbias,
                       const float* in_scale, const float* in_shift,
                       int B, int H, int W, int Cin, int ldx,
                       int Ho, int Wo, int Cout, int lddy,
                       int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                       int D, int Do, int kd, int sd, int pad_d,
                       int in_relu, int groups, int prezeroed, int dtype, void* stream, std::vector<WgfPlan>* plans) {
  auto one = [&](const void* xx, float* dw, int cin, int prez) {
    WgfPlan pl;
    bool planned = false;
    const int rc = wgrad_one(xx, dy, dw, dbias, in_scale, in_shift, B, H, W, cin, ldx, Ho, Wo, Cout, lddy, kh, kw, stride, dil, pad_t, pad_l,
                             D, Do, kd, sd, pad_d, in_relu, groups, prez, dtype, stream, plans ? &pl : nullptr, plans ? &planned : nullptr);
    if (rc == SDHIP_OK && planned) plans->push_back(pl);
    return rc;
  };
  const bool halves = dtype == SDHIP_BF16 && x && dw_packed && Cin == 64 && Cout <= 32 && kh == 3 && kw == 3 && stride == 1 && dil == 1 && sd == 1 &&
                      (kd == 3 || (kd == 1 && sdhip_diag().wgrad_split2d)) && !in_scale && !dbias && Wo >= 24 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0 &&
                      !sdhip_diag().wgrad_no_half32;
  if (!halves) return one(x, dw_packed, Cin, prezeroed);
  const int rc = one(x, dw_packed, 32, prezeroed);           // (clears all 64 columns when asked to: one 64-channel chunk either way)
  if (rc != SDHIP_OK) return rc;
  return one((const bf16_t*)x + 32, dw_packed + 32, 32, 1);
}

extern "C" int sdhip_conv2d_wgrad(const void* x, const void* dy, float* dw_packed, float* dbias,
                                  const float* in_scale, const float* in_shift,
                                  int B, int H, int W, int Cin, int ldx,
                                  int Ho, int Wo, int Cout, int lddy,
                                  int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                                  int D, int Do, int kd, int sd, int pad_d,
                                  int in_relu, int groups, int prezeroed, int dtype, void* stream) {
  return wgrad_layer(x, dy, dw_packed, dbias, in_scale, in_shift, B, H, W, Cin, ldx, Ho, Wo, Cout, lddy, kh, kw, stride, dil, pad_t, pad_l,
                     D, Do, kd, sd, pad_d, in_relu, groups, prezeroed, dtype, stream, nullptr);
}

namespace {

// One grid for `k` (<= kWgfGroupMax) planned layers of one instantiation.  Tile shares per layer: every workgroup should
// sweep for about the same time tau; tau is chosen by a sweep over the cost model of plan_wgf (sweep time per tile, closing
// atomics per workgroup at the chip-wide atomic rate) against the number of workgroup slots.
int launch_wgf_group(std::vector<WgfPlan>& pls, size_t lo, size_t hi, hipStream_t s, int max_wg) {
  const int k = (int)(hi - lo);
  if (k == 1) {
    if (max_wg > 0 && pls[lo].gx * pls[lo].gy > max_wg) pls[lo].gx = max_wg / pls[lo].gy > 0 ? max_wg / pls[lo].gy : 1;
    return launch_wgf_plan(pls[lo], s);
  }
  size_t lds = 0;
  int occ = 2;
  for (size_t i = lo; i < hi; ++i) { if (pls[i].lds > lds) lds = pls[i].lds; if (pls[i].occ < occ) occ = pls[i].occ; }
  if (2 * lds > 160 * 1024) occ = 1;
  // max_wg > 0: the grid must stay within that many workgroups in ONE round, so that it leaves CUs free for the kernels of
  // another stream (the weight gradients of the decoder beside the latency-bound DenseNet backward chain)
  const int slots = max_wg > 0 && max_wg < 256 * occ ? max_wg : 256 * occ;
  // Tile shares: for 1..4 rounds of `slots` workgroups, the smallest sweep time tau per workgroup whose shares fit the
  // budget (bisection), then the candidate with the best modelled time.  The grid must not spill a few workgroups past a
  // round: with uniform sweep times, 288 workgroups on 256 slots take two rounds (measured: 17 full-resolution layers in
  // one grid 1.33 ms against 0.82 ms launched one by one, before this rule).
  double best = 1e30;
  int best_gx[kWgfGroupMax];
  auto shares = [&](double tau, int* gxs, double& tmax, double& atom) {
    long nwg = 0;
    tmax = 0.; atom = 0.;
    for (int i = 0; i < k; ++i) {
      const WgfPlan& p = pls[lo + i];
      int gx = (int)ceil(p.ntiles * p.t_tile_us / tau);
      const int cap = sdhip_cdiv(p.ntiles, 2);             // at least two tiles per workgroup: the pipeline overlaps them
      if (gx > cap) gx = cap;
      if (gx < 1) gx = 1;
      const double per = sdhip_cdiv(p.ntiles, gx) * p.t_tile_us + 2.5;   // + start-up and flush latency of a workgroup
      if (per > tmax) tmax = per;
      atom += (double)gx * p.gy * p.flush_us;
      gxs[i] = gx;
      nwg += (long)gx * p.gy;
    }
    return nwg;
  };
  for (int rounds = 1; rounds <= (max_wg > 0 ? 1 : 4); ++rounds) {
    const long budget = (long)slots * rounds;
    double tl = 0.05, th = 1e5, tmax, atom;
    int gxs[kWgfGroupMax];
    for (int it = 0; it < 48; ++it) {
      const double tm = sqrt(tl * th);
      if (shares(tm, gxs, tmax, atom) > budget) tl = tm; else th = tm;
    }
    const long nwg = shares(th, gxs, tmax, atom);
    const double t = (double)((nwg + slots - 1) / slots) * tmax + atom;
    if (t < best) { best = t; for (int i = 0; i < k; ++i) best_gx[i] = gxs[i]; }
  }
  // longest workgroups first: the tail of the grid is made of short ones
  int order[kWgfGroupMax];
  for (int i = 0; i < k; ++i) order[i] = i;
  std::sort(order, order + k, [&](int a, int b) {
    const double ta = sdhip_cdiv(pls[lo + a].ntiles, best_gx[a]) * pls[lo + a].t_tile_us;
    const double tb = sdhip_cdiv(pls[lo + b].ntiles, best_gx[b]) * pls[lo + b].t_tile_us;
    return ta > tb;
  });
  WgfGroup g = WgfGroup();
  g.n = k;
  int wg = 0;
  for (int j = 0; j < k; ++j) {
    const int i = order[j];
    g.wg0[j] = wg;
    g.gx[j] = best_gx[i];
    g.L[j] = pls[lo + i].a;
    wg += best_gx[i] * pls[lo + i].gy;
  }
  for (int j = k; j <= kWgfGroupMax; ++j) g.wg0[j] = wg;
  void* args[] = {&g};
  if (hipLaunchKernel(pls[lo].group, dim3(wg), dim3(512), args, lds, s) != hipSuccess)
    SDHIP_FAIL(SDHIP_ERR_LAUNCH, "conv2d_wgrad_group: launch failed: %s", hipGetErrorString(hipGetLastError()));
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

}  // namespace

extern "C" int sdhip_conv2d_wgrad_group(const SdhipWgradItem* items, int n, int max_workgroups, int dtype, void* stream) {
  SDHIP_CHECK_ARG(items && n > 0, "conv2d_wgrad_group: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  std::vector<WgfPlan> plans;
  plans.reserve(n);
  for (int i = 0; i < n; ++i) {
    const SdhipWgradItem& it = items[i];
    const int rc = wgrad_layer(it.x, it.dy, it.dw_packed, it.dbias, it.in_scale, it.in_shift, it.B, it.H, it.W, it.Cin, it.ldx, it.Ho, it.Wo,
                               it.Cout, it.lddy, it.kh, it.kw, it.stride, it.dil, it.pad_t, it.pad_l, it.D, it.Do, it.kd, it.sd, it.pad_d,
                               it.in_relu, it.groups, 1, dtype, stream, &plans);
    if (rc != SDHIP_OK) return rc;
  }
  // buckets = instantiations, in order of first appearance; <= kWgfGroupMax layers per grid
  std::vector<char> done(plans.size(), 0);
  std::vector<WgfPlan> bucket;
  for (size_t i = 0; i < plans.size(); ++i) {
    if (done[i]) continue;
    bucket.clear();
    for (size_t j = i; j < plans.size(); ++j)
      if (!done[j] && plans[j].group == plans[i].group) { bucket.push_back(plans[j]); done[j] = 1; }
    // even chunks (24 layers -> 12 + 12, not 16 + 8)
    const size_t nchunk = (bucket.size() + kWgfGroupMax - 1) / kWgfGroupMax;
    const size_t per = (bucket.size() + nchunk - 1) / nchunk;
    for (size_t lo = 0; lo < bucket.size(); lo += per) {
      const size_t hi = lo + per < bucket.size() ? lo + per : bucket.size();
      const int rc = launch_wgf_group(bucket, lo, hi, s, max_workgroups);
      if (rc != SDHIP_OK) return rc;
    }
  }
  return SDHIP_OK;
}
