// Spatial correlation sampler (kernel_size=1, stride=1, padding=0) for gfx950.
// Replaces the third-party `SpatialCorrelationSampler` op used at
// models/dsnet_t2.py:1078-1087,1188-1193,1233-1234 (1-D, patch (1,17)) and
// models/dsnet_t2.py:129-133,221-223 (2-D, patch (17,17)).
//
// The op is HBM/latency bound (AI ~ 4 FLOP/B): one wave64 owns one pixel, its
// lanes own 16-byte channel chunks, the inner product over channels is finished
// with a wave shuffle reduction.  NHWC, so every pixel's channels are one
// contiguous, coalesced read.
#include "conv_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

// dot product of one 16-byte chunk pair
template <typename T>
__device__ __forceinline__ float chunk_dot(const u32x4& a, const u32x4& b) {
  float fa[Chunk<T>::N], fb[Chunk<T>::N];
  Chunk<T>::unpack(a, fa);
  Chunk<T>::unpack(b, fb);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < Chunk<T>::N; ++i) s = fmaf(fa[i], fb[i], s);
  return s;
}

// VEC=true : C and ld are multiples of the 16-byte chunk, pointers 16-B aligned.
template <typename T, bool VEC>
__global__ __launch_bounds__(kWavesPerBlock * 64) void corr_fwd_kernel(
    const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out,
    int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);  // B*H*W < 2^31 (host-checked)
  if (pix >= B * H * W) return;  // wave-uniform
  const int w = pix % W;
  const int h = (pix / W) % H;
  const int b = pix / (W * H);
  const T* a_ptr = in1 + (long)pix * ld_in;
  T* o_ptr = out + (long)pix * ld_out;
  constexpr int V = VEC ? Chunk<T>::N : 1;
  const int rh = PH / 2, rw = PW / 2;

  for (int ph = 0; ph < PH; ++ph) {
    const int h2 = h + (ph - rh) * dil;
    for (int pw = 0; pw < PW; ++pw) {
      const int w2 = w + (pw - rw) * dil;
      float acc = 0.f;
      if (h2 >= 0 && h2 < H && w2 >= 0 && w2 < W) {  // wave-uniform
        const T* b_ptr = in2 + (long)((b * H + h2) * W + w2) * ld_in;
        for (int c = lane * V; c < C; c += 64 * V) {
          if constexpr (VEC) {
            u32x4 av = *reinterpret_cast<const u32x4*>(a_ptr + c);
            u32x4 bv = *reinterpret_cast<const u32x4*>(b_ptr + c);
            acc += chunk_dot<T>(av, bv);
          } else {
            acc = fmaf(Elem<T>::ld(a_ptr + c), Elem<T>::ld(b_ptr + c), acc);
          }
        }
        acc = wave_sum(acc);
      }
      if (lane == 0) Elem<T>::st(o_ptr + ph * PW + pw, acc);
    }
  }
}

// One wave per pixel computes both input gradients of that pixel:
//   gin1[pix,c] = sum_p gout[pix,p]       * in2[pix + d(p), c]
//   gin2[pix,c] = sum_p gout[pix - d(p),p] * in1[pix - d(p), c]
template <typename T, bool VEC>
__global__ __launch_bounds__(kWavesPerBlock * 64) void corr_bwd_kernel(
    const T* __restrict__ in1, const T* __restrict__ in2, const T* __restrict__ gout,
    T* __restrict__ gin1, T* __restrict__ gin2,
    int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pix >= B * H * W) return;
  const int w = pix % W;
  const int h = (pix / W) % H;
  const int b = pix / (W * H);
  constexpr int V = VEC ? Chunk<T>::N : 1;
  const int rh = PH / 2, rw = PW / 2;

  for (int c = lane * V; c < C; c += 64 * V) {
    float g1[V], g2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { g1[i] = 0.f; g2[i] = 0.f; }
    for (int ph = 0; ph < PH; ++ph) {
      for (int pw = 0; pw < PW; ++pw) {
        const int dy = (ph - rh) * dil, dx = (pw - rw) * dil;
        const int p = ph * PW + pw;
        // gin1: partner pixel in in2 at +d
        {
          const int h2 = h + dy, w2 = w + dx;
          if (h2 >= 0 && h2 < H && w2 >= 0 && w2 < W) {
            const float g = Elem<T>::ld(gout + (long)pix * ld_out + p);
            const T* src = in2 + (long)((b * H + h2) * W + w2) * ld_in + c;
            if constexpr (VEC) {
              float f[V];
              Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(src), f);
#pragma unroll
              for (int i = 0; i < V; ++i) g1[i] = fmaf(g, f[i], g1[i]);
            } else {
              g1[0] = fmaf(g, Elem<T>::ld(src), g1[0]);
            }
          }
        }
        // gin2: partner pixel in in1 at -d
        {
          const int h1 = h - dy, w1 = w - dx;
          if (h1 >= 0 && h1 < H && w1 >= 0 && w1 < W) {
            const long q = (b * H + h1) * W + w1;
            const float g = Elem<T>::ld(gout + q * ld_out + p);
            const T* src = in1 + q * ld_in + c;
            if constexpr (VEC) {
              float f[V];
              Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(src), f);
#pragma unroll
              for (int i = 0; i < V; ++i) g2[i] = fmaf(g, f[i], g2[i]);
            } else {
              g2[0] = fmaf(g, Elem<T>::ld(src), g2[0]);
            }
          }
        }
      }
    }
    if constexpr (VEC) {
      *reinterpret_cast<u32x4*>(gin1 + (long)pix * ld_in + c) = Chunk<T>::pack(g1);
      *reinterpret_cast<u32x4*>(gin2 + (long)pix * ld_in + c) = Chunk<T>::pack(g2);
    } else {
      Elem<T>::st(gin1 + (long)pix * ld_in + c, g1[0]);
      Elem<T>::st(gin2 + (long)pix * ld_in + c, g2[0]);
    }
  }
}

// =====================================================================================================================
// Tiled MFMA form (bf16, dilation_patch = 1): the displacement products of an 8 x 8 pixel tile are ONE matrix product of
// the tile's 64 pixels with the (8 + PH - 1) x (8 + PW - 1) halo of the other map, contraction over channels on the
// matrix cores.  About half of the products of a 17 x 17 patch lie outside the window and are discarded — still two orders
// of magnitude less work per output than the wave-per-pixel form, which runs PH * PW serial wave reductions per pixel
// (289 for the 2-D correlation of models/dsnet_t2.py:129-133,221-223: 510 us forward + 581 us backward at (8,C,32,64)).
// Both maps are staged once per 64-channel chunk in LDS (conv_common.h's swizzled [pixel][channel] image); the outputs of
// a tile are gathered in LDS and written as whole (B,H,W,PH*PW) rows.

typedef __attribute__((address_space(3))) bf16x4_t* corr_lds_bf4_p;

// [pixel][channel] image for TRANSPOSING fragment reads (ds_read_b64_tr_b16: the contraction of the backward pass runs
// over pixels): 128-byte rows, chunk slot = c ^ key(row), key = (row & 2) | ((row >> 1) & 4) — conflict-free for the row
// groups {r..r+3, r+8..r+11} a 32-lane half touches (the layout of conv_wgrad_fast.h).
struct TRow {
  static __device__ __forceinline__ int key(int row) { return (row & 2) | ((row >> 1) & 4); }
  static __device__ __forceinline__ int off(int row, int c) { return row * 128 + ((c ^ key(row)) << 4); }
};

__device__ __forceinline__ u32x4 corr_tr_pair(const unsigned char* lo, const unsigned char* hi) {
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((corr_lds_bf4_p)lo);
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((corr_lds_bf4_p)hi);
  const u32x2 l2 = __builtin_bit_cast(u32x2, a), h2 = __builtin_bit_cast(u32x2, b);
  return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

// One 512-thread workgroup per tile.  NT: 16-pixel halo tiles per wave QUARTET (waves = 4 halo-tile groups x 2 halves of the
// tile's pixels): 9 for a 17 x 17 patch (24 x 24 halo = 36 tiles), 3 for 1 x 17 (8 x 24), 4 for 1 x 21.  The next channel
// chunk travels through registers while the current one is multiplied (one wave per SIMD: nothing else hides the loads).
template <int NT>
__global__ __launch_bounds__(512) void corr_tile_fwd_kernel(const bf16_t* __restrict__ in1, const bf16_t* __restrict__ in2,
                                                            bf16_t* __restrict__ out, int B, int H, int W, int C, int ld,
                                                            int PH, int PW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTH = 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int tiles_w = (W + 7) >> 3, tiles_h = (H + 7) >> 3;
  int t = blockIdx.x;
  const int tx = t % tiles_w; t /= tiles_w;
  const int ty = t % tiles_h;
  const int b = t / tiles_h;
  const int h0 = ty * 8, w0 = tx * 8, rh = PH / 2, rw = PW / 2;
  const int IH = 8 + PH - 1, IW = 8 + PW - 1, NH = IH * IW;
  constexpr int NPAD = NT * 4 * 16;              // = NT * 64 rows: NT 16-byte loads per lane fill the halo image
  unsigned char* halo = smem;                    // NPAD rows of 128 bytes (rows >= NH hold zeros; their results are dropped)
  unsigned char* atile = smem + NPAD * 128;      // 64 rows
  const int mg = wave & 3, nh = wave >> 2;

  f32x4 acc[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long img = (long)b * H * W;
  const unsigned magic_iw = div_magic(IW);
  // per-lane source offsets of the lane's halo rows (the same for every channel chunk) and of its tile row
  int hoff[NT], aoff;
  const int cl = (tid & 7) * 8;                  // first channel of the lane's 16-byte chunk inside a 64-channel chunk
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int pix = (tid + j * NTH) >> 3;
    const int hr = fast_div(pix, IW, magic_iw), hc = pix - hr * IW;
    const int gh = h0 - rh + hr, gw = w0 - rw + hc;
    hoff[j] = (pix < NH && gh >= 0 && gh < H && gw >= 0 && gw < W) ? (gh * W + gw) * ld + cl : -1;
  }
  {
    const int n = tid >> 3, gh = h0 + (n >> 3), gw = w0 + (n & 7);
    aoff = (gh < H && gw < W) ? (gh * W + gw) * ld + cl : -1;
  }
  u32x4 rh_[NT], ra_;
  auto load = [&](int q) {
    const bool okc = q * 64 + cl < C;
    const bf16_t* s2 = in2 + img * ld + q * 64;
    const bf16_t* s1 = in1 + img * ld + q * 64;
#pragma unroll
    for (int j = 0; j < NT; ++j) rh_[j] = (okc && hoff[j] >= 0) ? *reinterpret_cast<const u32x4*>(s2 + hoff[j]) : u32x4{0u, 0u, 0u, 0u};
    ra_ = (okc && aoff >= 0) ? *reinterpret_cast<const u32x4*>(s1 + aoff) : u32x4{0u, 0u, 0u, 0u};
  };
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<u32x4*>(halo + lds_off((tid + j * NTH) >> 3, tid & 7)) = rh_[j];
    *reinterpret_cast<u32x4*>(atile + lds_off(tid >> 3, tid & 7)) = ra_;
  };

  const int nq = (C + 63) >> 6;
  load(0);
  store();
  __syncthreads();
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) load(q + 1);                 // in flight behind this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) bf[i] = *reinterpret_cast<const u32x4*>(atile + lds_off(16 * (2 * nh + i) + l15, 4 * ks + lg));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(halo + lds_off(16 * (mg * NT + j) + l15, 4 * ks + lg));
#pragma unroll
        for (int i = 0; i < 2; ++i) Mma<bf16_t>::run(acc[j][i], af, bf[i]);
      }
    }
    __syncthreads();                             // chunk q consumed by everybody
    if (q + 1 < nq) store();
    __syncthreads();                             // chunk q + 1 visible (last trip: the LDS is free for the output tile)
  }

  // D[m = halo pixel][n = tile pixel]: a lane holds halo pixels 4*lg .. 4*lg+3 of tile pixel l15.  Gather the tile's
  // (64, PH*PW) outputs in LDS, then write whole rows.
  bf16_t* ot = reinterpret_cast<bf16_t*>(smem);
  const int P = PH * PW;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * (mg * NT + j) + 4 * lg + r;
      const int hr = fast_div(m, IW, magic_iw), hc = m - hr * IW;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = 16 * (2 * nh + i) + l15;
        const int ph = hr - (n >> 3), pw = hc - (n & 7);
        if (m < NH && (unsigned)ph < (unsigned)PH && (unsigned)pw < (unsigned)PW) ot[n * P + ph * PW + pw] = f2bf(acc[j][i][r]);
      }
    }
  }
  __syncthreads();
  const int ncol = min(8, W - w0);
  for (int ay = 0; ay < 8 && h0 + ay < H; ++ay) {
    bf16_t* dst = out + ((long)(b * H + h0 + ay) * W + w0) * P;
    const bf16_t* src = ot + ay * 8 * P;
    const int n = ncol * P;
    if ((((uintptr_t)dst) & 15) == 0) {          // (8 * P bf16 per tile row = 16 * P bytes: aligned whenever W % 8 == 0)
      const int n16 = n >> 3;
      for (int i = tid; i < n16; i += NTH) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
      for (int i = (n16 << 3) + tid; i < n; i += NTH) dst[i] = src[i];
    } else {
      for (int i = tid; i < n; i += NTH) dst[i] = src[i];
    }
  }
}

// Backward of the tiled form.  For an 8 x 8 tile of pixels p of map 1 (pass 0):
//   gin1[p, c] = sum over halo pixels k of G[p][k] * in2[k, c],   G[p][k] = gout[p, k - p] inside the patch, 0 outside;
// for a tile of pixels q of map 2 (pass 1) the same product with the displacement mirrored:
//   gin2[q, c] = sum over halo pixels k of G'[q][k] * in1[k, c],  G'[q][k] = gout[k, q - k].
// G (64 x halo) is built once per tile in LDS; the halo of the other map is staged per 64-channel chunk as a [pixel][channel]
// image and read with the transposing LDS read (the contraction runs over PIXELS).  D[m = channel][n = tile pixel]: a lane
// ends up with 4 consecutive channels of one pixel (8-byte stores).  NKS: 32-pixel k-steps covering the halo.
template <int NKS>
__global__ __launch_bounds__(512) void corr_tile_bwd_kernel(const bf16_t* __restrict__ in1, const bf16_t* __restrict__ in2,
                                                            const bf16_t* __restrict__ gout, bf16_t* __restrict__ gin1,
                                                            bf16_t* __restrict__ gin2, int B, int H, int W, int C, int ld,
                                                            int PH, int PW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTH = 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int pass = blockIdx.y;
  const bf16_t* src = pass == 0 ? in2 : in1;
  bf16_t* dst = pass == 0 ? gin1 : gin2;
  const int tiles_w = (W + 7) >> 3, tiles_h = (H + 7) >> 3;
  int t = blockIdx.x;
  const int tx = t % tiles_w; t /= tiles_w;
  const int ty = t % tiles_h;
  const int b = t / tiles_h;
  const int h0 = ty * 8, w0 = tx * 8, rh = PH / 2, rw = PW / 2;
  const int IH = 8 + PH - 1, IW = 8 + PW - 1, NH = IH * IW, P = PH * PW;
  constexpr int NPAD = 32 * NKS;
  constexpr int GP = NPAD * 2 + 16;              // row pitch of G: 16 consecutive rows fall on distinct bank groups
  unsigned char* G = smem;                       // 64 rows
  unsigned char* halo = smem + 64 * GP;          // NPAD rows of 128 bytes (GP is a multiple of 16)

  // waves: channel tile = wave & 3 (16 channels), pixel tiles 2 * (wave >> 2) and + 1
  const int ctile = wave & 3, pt0 = (wave >> 2) * 2;
  const int p4 = lane & 3, r4 = l15 >> 2;
  const int ctr = ctile * 2 + (p4 >> 1);         // 16-byte chunk of this wave's 16-channel tile the lane addresses
  const int lo_off = TRow::off(8 * lg + r4, ctr) + ((p4 & 1) << 3);
  const int hi_off = TRow::off(8 * lg + r4 + 4, ctr) + ((p4 & 1) << 3);
  const unsigned magic_iw = div_magic(IW);
  const long img = (long)b * H * W;
  const int nq = (C + 63) >> 6;
  // halo rows of the lane (the same for every channel chunk); the next chunk travels through registers behind the MFMAs
  constexpr int NL = (NKS + 1) / 2;              // 16-byte loads per lane per chunk: NPAD * 8 / 512
  const int cl = (tid & 7) * 8;
  int hoff[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int pix = (tid + j * NTH) >> 3;
    const int hr = fast_div(pix, IW, magic_iw), hc = pix - hr * IW;
    const int gh = h0 - rh + hr, gw = w0 - rw + hc;
    hoff[j] = (pix < NH && gh >= 0 && gh < H && gw >= 0 && gw < W) ? (gh * W + gw) * ld + cl : -1;
  }
  u32x4 rr[NL];
  auto load = [&](int q) {
    const bool okc = q * 64 + cl < C;
    const bf16_t* sp = src + img * ld + q * 64;
#pragma unroll
    for (int j = 0; j < NL; ++j) rr[j] = (okc && hoff[j] >= 0) ? *reinterpret_cast<const u32x4*>(sp + hoff[j]) : u32x4{0u, 0u, 0u, 0u};
  };
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = tid + j * NTH;
      if (i < NPAD * 8) *reinterpret_cast<u32x4*>(halo + TRow::off(i >> 3, i & 7)) = rr[j];
    }
  };
  load(0);                                       // chunk 0 of the halo is in flight while G is built

  // ---- G: zero, then place the patch values ----
  for (int i = tid; i < (64 * GP) / 16; i += NTH) reinterpret_cast<u32x4*>(G)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  constexpr int UB = 8;                          // loads in flight per lane (the fill is a chain of L2 round trips otherwise)
  if (pass == 0) {
    // e = (tile pixel n, displacement d): consecutive lanes read consecutive elements of one gout row
    const unsigned magic_p = div_magic(P), magic_pw = div_magic(PW);
    for (int e0 = tid; e0 < 64 * P; e0 += NTH * UB) {
      bf16_t v[UB];
      int off[UB];
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        const int e = e0 + j * NTH;
        const int n = fast_div(e, P, magic_p), d = e - n * P;
        const int ph = fast_div(d, PW, magic_pw), pw = d - ph * PW;
        const int ay = n >> 3, ax = n & 7;
        const int gh = h0 + ay, gw = w0 + ax;
        const bool ok = e < 64 * P && gh < H && gw < W;
        off[j] = ok ? n * GP + ((ay + ph) * IW + ax + pw) * 2 : -1;
        v[j] = ok ? gout[(img + (long)gh * W + gw) * P + d] : (bf16_t)0;
      }
#pragma unroll
      for (int j = 0; j < UB; ++j)
        if (off[j] >= 0) *reinterpret_cast<bf16_t*>(G + off[j]) = v[j];
    }
  } else {
    // G'[q][k] = gout[k, q - k]: enumerate (tile row ay, displacement row ph, halo column hc, j) with pw = PW-1-hc+j and
    // tile column ax = j — the 8 lanes of a (ay, ph, hc) group read 8 consecutive elements of ONE gout row (the row of the
    // map-1 pixel in halo column hc), instead of 8 different rows 578 bytes apart
    const int per_ay = PH * IW * 8, total = 8 * per_ay;
    const unsigned magic_a = div_magic(per_ay), magic_h = div_magic(IW * 8);
    for (int e0 = tid; e0 < total; e0 += NTH * UB) {
      bf16_t v[UB];
      int off[UB];
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        const int e = e0 + j * NTH;
        const int ay = fast_div(e, per_ay, magic_a), r1 = e - ay * per_ay;
        const int ph = fast_div(r1, IW * 8, magic_h), r2 = r1 - ph * IW * 8;
        const int hc = r2 >> 3, ax = r2 & 7;
        const int pw = PW - 1 - hc + ax;
        const int gh = h0 + ay, gw = w0 + ax;
        const int sh = gh + rh - ph, sw = gw + rw - pw;        // the map-1 pixel whose displacement (ph, pw) lands on (gh, gw)
        const bool ok = e < total && (unsigned)pw < (unsigned)PW && gh < H && gw < W && sh >= 0 && sh < H && sw >= 0 && sw < W;
        off[j] = ok ? (ay * 8 + ax) * GP + ((ay + (PH - 1) - ph) * IW + hc) * 2 : -1;
        v[j] = ok ? gout[(img + (long)sh * W + sw) * P + ph * PW + pw] : (bf16_t)0;
      }
#pragma unroll
      for (int j = 0; j < UB; ++j)
        if (off[j] >= 0) *reinterpret_cast<bf16_t*>(G + off[j]) = v[j];
    }
  }

  store();
  __syncthreads();                               // G complete, chunk 0 visible
  // G's fragments are the same for every channel chunk: read once, kept in registers (the chunk loop is otherwise bound by
  // LDS reads: two 16-byte G reads per transposing halo read)
  u32x4 gfr[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) gfr[ks][i] = *reinterpret_cast<const u32x4*>(G + (16 * (pt0 + i) + l15) * GP + (ks * 32 + 8 * lg) * 2);
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) load(q + 1);
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const u32x4 af = corr_tr_pair(halo + lo_off + ks * 4096, halo + hi_off + ks * 4096);
#pragma unroll
      for (int i = 0; i < 2; ++i) Mma<bf16_t>::run(acc[i], af, gfr[ks][i]);
    }
    const int ch = q * 64 + 16 * ctile + 4 * lg;
    if (ch < C) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = 16 * (pt0 + i) + l15;
        const int gh = h0 + (n >> 3), gw = w0 + (n & 7);
        if (gh < H && gw < W)
          *reinterpret_cast<u32x2*>(dst + (img + (long)gh * W + gw) * ld + ch) = u32x2{pack2bf(acc[i][0], acc[i][1]), pack2bf(acc[i][2], acc[i][3])};
      }
    }
    __syncthreads();                             // chunk q consumed by everybody
    if (q + 1 < nq) store();
    __syncthreads();                             // chunk q + 1 visible
  }
}

// tiles per wave of the forward kernel / k-steps of the backward kernel for a patch, 0 = no tiled instantiation
inline int corr_tile_plan(int PH, int PW, int* nks) {
  const int nh = (8 + PH - 1) * (8 + PW - 1);
  const int tiles = (nh + 15) / 16;
  *nks = (nh + 31) / 32;
  const int nt = (tiles + 3) / 4;
  return (nt == 3 || nt == 4 || nt == 9) ? nt : 0;
}

template <typename K>
int corr_raise_lds(K kern, size_t lds) {
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    SDHIP_FAIL(SDHIP_ERR_LAUNCH, "corr: cannot raise the dynamic LDS limit");
  return SDHIP_OK;
}

bool corr_tiled_ok(const void* a, const void* b, const void* c, const void* d, int C, int ld, int PH, int PW, int dil, int ld_out, int dtype) {
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
  int nks;
  return dtype == SDHIP_BF16 && dil == 1 && C % 8 == 0 && ld % 8 == 0 && al(a) && al(b) && al(c) && al(d) && ld_out == PH * PW &&
         PH * PW >= 9 && corr_tile_plan(PH, PW, &nks) != 0 && !sdhip_diag().corr_no_tiled;
}

template <typename T>
bool vec_ok(const void* a, const void* b, const void* c, const void* d, int C, int ld) {
  const int n = Chunk<T>::N;
  auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
  return (C % n == 0) && (ld % n == 0) && al(a) && al(b) && al(c) && al(d);
}

int check_common(int B, int H, int W, int C, int ld_in, int PH, int PW, int dil, int ld_out, int dtype) {
  SDHIP_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0, "corr: empty tensor B=%d H=%d W=%d C=%d", B, H, W, C);
  SDHIP_CHECK_ARG(PH > 0 && PW > 0 && (PH & 1) && (PW & 1), "corr: patch size must be odd, got (%d,%d)", PH, PW);
  SDHIP_CHECK_ARG(dil >= 1, "corr: dilation_patch must be >= 1");
  SDHIP_CHECK_ARG(ld_in >= C && ld_out >= PH * PW, "corr: pixel stride smaller than channel count");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "corr: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG((long)B * H * W < (1L << 31), "corr: more than 2^31 pixels");
  return 0;
}

}  // namespace

extern "C" int sdhip_corr_fwd(const void* in1, const void* in2, void* out, int B, int H, int W, int C,
                              int ld_in, int PH, int PW, int dil_patch, int ld_out, int dtype, void* stream) {
  if (int rc = check_common(B, H, W, C, ld_in, PH, PW, dil_patch, ld_out, dtype)) return rc;
  SDHIP_CHECK_ARG(in1 && in2 && out, "corr_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  if (corr_tiled_ok(in1, in2, nullptr, nullptr, C, ld_in, PH, PW, dil_patch, ld_out, dtype)) {
    int nks;
    const int nt = corr_tile_plan(PH, PW, &nks);
    const int tiles = B * sdhip_cdiv(H, 8) * sdhip_cdiv(W, 8);
    const size_t lds = (size_t)(nt * 64 + 64) * 128;
#define TFWD(NT) do { if (int rc = corr_raise_lds(corr_tile_fwd_kernel<NT>, lds)) return rc; \
    hipLaunchKernelGGL((corr_tile_fwd_kernel<NT>), dim3(tiles), dim3(512), lds, s, (const bf16_t*)in1, (const bf16_t*)in2, (bf16_t*)out, \
                       B, H, W, C, ld_in, PH, PW); } while (0)
    if (nt == 3) TFWD(3); else if (nt == 4) TFWD(4); else TFWD(9);
#undef TFWD
    SDHIP_LAUNCH_CHECK();
    return SDHIP_OK;
  }
  dim3 grid(sdhip_cdiv(npix, kWavesPerBlock)), block(kWavesPerBlock * 64);
#define LAUNCH(T, VEC) hipLaunchKernelGGL((corr_fwd_kernel<T, VEC>), grid, block, 0, s, (const T*)in1, (const T*)in2, \
                                          (T*)out, B, H, W, C, ld_in, PH, PW, dil_patch, ld_out)
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(in1, in2, nullptr, nullptr, C, ld_in)) LAUNCH(float, true); else LAUNCH(float, false);
  } else {
    if (vec_ok<bf16_t>(in1, in2, nullptr, nullptr, C, ld_in)) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
  }
#undef LAUNCH
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_corr_bwd(const void* in1, const void* in2, const void* gout, void* gin1, void* gin2,
                              int B, int H, int W, int C, int ld_in, int PH, int PW, int dil_patch,
                              int ld_out, int dtype, void* stream) {
  if (int rc = check_common(B, H, W, C, ld_in, PH, PW, dil_patch, ld_out, dtype)) return rc;
  SDHIP_CHECK_ARG(in1 && in2 && gout && gin1 && gin2, "corr_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long npix = (long)B * H * W;
  if (corr_tiled_ok(in1, in2, gin1, gin2, C, ld_in, PH, PW, dil_patch, ld_out, dtype) && (((uintptr_t)gout) & 1) == 0) {
    int nks;
    corr_tile_plan(PH, PW, &nks);
    const int tiles = B * sdhip_cdiv(H, 8) * sdhip_cdiv(W, 8);
    const size_t lds = (size_t)64 * (nks * 64 + 16) + (size_t)nks * 32 * 128;
#define TBWD(NKS) do { if (int rc = corr_raise_lds(corr_tile_bwd_kernel<NKS>, lds)) return rc; \
    hipLaunchKernelGGL((corr_tile_bwd_kernel<NKS>), dim3(tiles, 2), dim3(512), lds, s, (const bf16_t*)in1, (const bf16_t*)in2, \
                       (const bf16_t*)gout, (bf16_t*)gin1, (bf16_t*)gin2, B, H, W, C, ld_in, PH, PW); } while (0)
    if (nks == 6) { TBWD(6); SDHIP_LAUNCH_CHECK(); return SDHIP_OK; }
    if (nks == 7) { TBWD(7); SDHIP_LAUNCH_CHECK(); return SDHIP_OK; }
    if (nks == 18) { TBWD(18); SDHIP_LAUNCH_CHECK(); return SDHIP_OK; }
#undef TBWD
  }
  dim3 grid(sdhip_cdiv(npix, kWavesPerBlock)), block(kWavesPerBlock * 64);
#define LAUNCH(T, VEC) hipLaunchKernelGGL((corr_bwd_kernel<T, VEC>), grid, block, 0, s, (const T*)in1, (const T*)in2, \
                                          (const T*)gout, (T*)gin1, (T*)gin2, B, H, W, C, ld_in, PH, PW, dil_patch, ld_out)
  if (dtype == SDHIP_F32) {
    if (vec_ok<float>(in1, in2, gin1, gin2, C, ld_in)) LAUNCH(float, true); else LAUNCH(float, false);
  } else {
    if (vec_ok<bf16_t>(in1, in2, gin1, gin2, C, ld_in)) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
  }
#undef LAUNCH
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
