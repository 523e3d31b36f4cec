// sdhip — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Everything here is wave64 / MFMA specific; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/sdhip.h"

#define SDHIP_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits (storage type)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing (thread-local last error, never abort) -----------------
void sdhip_set_error(const char* fmt, ...);
#define SDHIP_FAIL(code, ...) do { sdhip_set_error(__VA_ARGS__); return (code); } while (0)
#define SDHIP_CHECK_ARG(cond, ...) do { if (!(cond)) SDHIP_FAIL(SDHIP_ERR_ARG, __VA_ARGS__); } while (0)
#define SDHIP_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "%s: launch failed: %s", __func__, hipGetErrorString(e__)); } while (0)

// ---- diagnostic / tuning switches (runtime.hip): environment read once at library load ----
struct SdhipDiag {
  bool conv_generic, conv_big, conv_no_thin, conv_no_gemm, conv_no_band, conv_no_band3, wgrad_generic, wgrad_no_pack, wgrad_force_pack, wgrad_no_half, thin_wgrad_reg, corr_no_tiled, wgrad_no_half32, wgrad_split2d;
  int tune_s2_small;
  int tune_big, tune_split, tune_thin_blocks, tune_fused_blocks, tune_gemm_dbg, tune_band_dbg;
  double tune_atomic_tbs;
};
const SdhipDiag& sdhip_diag();

// ---- bf16 <-> f32 ------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, ((unsigned int)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }  // RNE, NaN-safe (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPer16B = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16) and back
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  // NB: bit_cast the WHOLE vector; __builtin_bit_cast on a vector-element lvalue (u[i]) reads element 0 (hipcc 7.2).
  static __device__ __forceinline__ void unpack(const u32x4& u, float* f) {
    const f32x4 v = __builtin_bit_cast(f32x4, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[i];
    return __builtin_bit_cast(u32x4, v);
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const u32x4& u, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(u[i]); f[2 * i + 1] = bfhi(u[i]); }
  }
  static __device__ __forceinline__ u32x4 pack(const float* f) {
    u32x4 u;
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return u;
  }
};

// ---- wave64 reductions (DPP-free, shuffle based) --------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// one "unit" of a row-wise elementwise kernel: a 16-byte chunk (VEC) or a single element
template <typename T, bool VEC> struct Unit {
  static constexpr int N = VEC ? Chunk<T>::N : 1;
  static __device__ __forceinline__ void load(const T* p, float* f) {
    if constexpr (VEC) Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(p), f);
    else f[0] = Elem<T>::ld(p);
  }
  static __device__ __forceinline__ void store(T* p, const float* f) {
    if constexpr (VEC) *reinterpret_cast<u32x4*>(p) = Chunk<T>::pack(f);
    else Elem<T>::st(p, f[0]);
  }
};

static inline int sdhip_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- zero fill as a KERNEL ------------------------------------------------------------------------------------------
// hipMemsetAsync becomes a memset node when the training step is captured into a hipGraph, and on ROCm 7.2 such a node
// was observed to stop clearing its 512-byte target once unrelated allocations had been made between two replays (the
// Lovasz class counters then accumulated from replay to replay: tests/diag/gpu_lovasz_graph.py).  Everything the
// library clears on a stream is therefore cleared by this kernel: a captured step consists of kernel nodes only.
namespace {
__global__ __launch_bounds__(256) void sdhip_zero_kernel(unsigned char* __restrict__ p, size_t bytes) {
  // any alignment, any length: byte head up to the first 16-byte boundary, 16-byte body, byte tail (a 2-byte-aligned bf16
  // buffer with an odd element count must not fall back to a memset node either)
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t head = (size_t)((16 - (((uintptr_t)p) & 15)) & 15);
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) >> 4;
  const size_t tail0 = head + (n16 << 4);
  u32x4* p16 = reinterpret_cast<u32x4*>(p + head);
  for (size_t j = i; j < n16; j += stride) p16[j] = u32x4{0u, 0u, 0u, 0u};
  if (i < head) p[i] = 0;
  if (tail0 + i < bytes && i < 16) p[tail0 + i] = 0;
}
inline hipError_t sdhip_zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  size_t blocks = (bytes / 16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sdhip_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned char*)ptr, bytes);
  return hipGetLastError();
}
}  // namespace
