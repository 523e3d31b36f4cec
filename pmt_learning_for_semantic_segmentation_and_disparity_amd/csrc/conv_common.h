// Shared pieces of the direct (im2col-free) convolution kernels.
//
// Data layout
//   activations : NHWC, pixel stride ld (elements)           -> one pixel = one contiguous run
//   LDS "row"   : 128 bytes = 8 chunks of 16 B = CK channels  (CK = 64 bf16 / 32 f32)
//   packed W    : [nq][T][Mpad][CK]  (nq = ceil(K/CK) channel chunks, T = kh*kw taps,
//                 Mpad = output channels rounded up to 16); row (q,t,m) holds the CK
//                 reduction-channel weights of output channel m for tap t, zero padded.
//
// MFMA orientation: D[M = out channel][N = pixel] = A(weights) x B(pixels); a lane ends up with
// 4 consecutive output channels of one pixel, i.e. a contiguous 8/16-byte NHWC store.
// Both operand fragments are one 16-byte LDS read per lane (ds_read_b128):
//   bf16: v_mfma_f32_16x16x32_bf16, lane (r = l&15, g = l>>4) holds k = 8g..8g+7
//   f32 : 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the chunk (k = channel 4g+i in MFMA i);
//         exact f32 FMA chain, used for the 1e-3 parity path.
#pragma once
#include "sdhip_common.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
  }
};

// XOR swizzle of the 16-byte chunk index inside a 128-byte LDS row: 16 consecutive rows read at
// the same logical chunk land on 16 distinct 16-byte bank slots (conflict-free ds_read_b128).
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + (swz(row, chunk) << 4); }

struct ConvGeom {
  int B, H, W;        // input  spatial
  int Ho, Wo;         // output spatial
  int kh, kw, stride, dil, pad_t, pad_l;
  // third (depth) axis of the 3-D convolutions of the PSMNet cost-volume network (models_psmnet/submodule.py:16-19):
  // a volume is stored [B][D][H][W][C]; the depth taps are folded into the channel-chunk loop (chunk = (kd, 128-byte
  // channel chunk)), so one launch accumulates all kd*kh*kw taps in registers.  2-D convs: D = Do = kd = sd = 1, pad_d = 0.
  int D, Do, kd, sd, pad_d;
};

// Stage one input tile (npx = IH*IW pixels, `1<<sh` 16-byte chunks per pixel of channel chunk q) into LDS with
// the fused prologue.  Loads are issued NB at a time before any LDS store so that NB global loads are in flight
// per lane (a load->store->load chain would expose the full HBM latency once per chunk).
struct StageSrc {
  const void* base;        // first pixel of the image
  int H, W, ld, C;         // image extent, pixel stride, channels
  int h0, w0;              // image coordinates of tile pixel (0,0)
  int IH, IW;              // tile extent
  const float* scale;      // per-channel prologue (already offset to the statistics group) or nullptr
  const float* shift;
  int relu, vec;
  unsigned magic_iw;       // fast_div(IW): see div_magic()
};

// n / d for n < 2^16 via one v_mul_hi_u32: magic = floor(2^32 / d) + 1 (d >= 2); d == 1 is the identity.
__device__ __forceinline__ unsigned div_magic(int d) { return d > 1 ? (unsigned)(0x100000000ULL / (unsigned)d) + 1u : 0u; }
__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) { return d > 1 ? (int)__umulhi((unsigned)n, magic) : n; }

template <typename T, int NB, int NT = 256>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const StageSrc& s, int q, int sh, int tid) {
  constexpr int V = Chunk<T>::N;
  constexpr int CK = 8 * V;
  const int total = (s.IH * s.IW) << sh;
  const T* xb = (const T*)s.base;
  // The lane's 16-byte chunk inside a pixel is the same for all its loads (NT is a multiple of the chunks per pixel):
  // its prologue coefficients are fetched ONCE, all loads in flight together (fetched per element inside the loop they
  // form a chain of dependent global loads — tens of microseconds per tile).
  float psc[V], psf[V];
  if (s.scale) {
    const int chl = q * CK + (tid & ((1 << sh) - 1)) * V;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const bool okc = chl + e < s.C;
      psc[e] = okc ? s.scale[okc ? chl + e : 0] : 0.f;
      psf[e] = okc ? s.shift[okc ? chl + e : 0] : 0.f;
    }
  }
  for (int i0 = tid; i0 < total; i0 += NT * NB) {
    u32x4 raw[NB];
    int ch[NB], off[NB];
    bool in[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = i0 + j * NT;
      const int pix = i >> sh, c = i & ((1 << sh) - 1);
      const int ih = fast_div(pix, s.IW, s.magic_iw), iw = pix - ih * s.IW;
      const int gh = s.h0 + ih, gw = s.w0 + iw;
      ch[j] = q * CK + c * V;
      off[j] = lds_off(pix, c);
      in[j] = i < total && gh >= 0 && gh < s.H && gw >= 0 && gw < s.W && ch[j] < s.C;
      raw[j] = u32x4{0u, 0u, 0u, 0u};
      if (in[j]) {
        const T* src = xb + ((long)gh * s.W + gw) * s.ld + ch[j];
        if (s.vec) {
          raw[j] = *reinterpret_cast<const u32x4*>(src);
          if (ch[j] + V > s.C) {   // odd channel count inside a padded pixel stride: the chunk's tail is not data
            float f[V];
            Chunk<T>::unpack(raw[j], f);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = (ch[j] + e < s.C) ? f[e] : 0.f;
            raw[j] = Chunk<T>::pack(f);
          }
        } else {
          float f[V];
#pragma unroll
          for (int e = 0; e < V; ++e) f[e] = (ch[j] + e < s.C) ? Elem<T>::ld(src + e) : 0.f;
          raw[j] = Chunk<T>::pack(f);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (i0 + j * NT < total) {
        if (s.scale && in[j]) {
          float f[V];
          Chunk<T>::unpack(raw[j], f);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            if (ch[j] + e < s.C) {
              const float v = fmaf(f[e], psc[e], psf[e]);
              f[e] = s.relu ? fmaxf(v, 0.f) : v;
            }
          }
          raw[j] = Chunk<T>::pack(f);
        }
        *reinterpret_cast<u32x4*>(lds + off[j]) = raw[j];
      }
    }
  }
}

// number of statistics replicas the conv epilogue / reduction kernels spread their atomics over: thousands of
// workgroups adding into ONE 256-byte line serialise at the memory side (~25 ns per wave-instruction).
#define SDHIP_NREP 32

// host-side: channels per 128-byte row
static inline int conv_ck(int dtype) { return dtype == SDHIP_BF16 ? 64 : 32; }
static inline int conv_esize(int dtype) { return dtype == SDHIP_BF16 ? 2 : 4; }
