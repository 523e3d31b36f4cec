// Pixel rows through LDS for the per-pixel class loops of the many-class loss kernels (19 Cityscapes classes: bf16 rows of 48
// bytes, f32 target rows of 76): a thread that walks its own row in global memory touches a new cache line every iteration
// and 64 lanes keep 64 lines busy for one element each (ce_kernel: 394 us for 340 MB; more workgroups made it SLOWER).  A
// workgroup's 256 pixels are one contiguous region of the NHWC tensor: it is copied to LDS with 16-byte vectors, the
// threads walk their rows there (row pitches of 12 / 19 words are conflict-free or nearly so), results go back the same way.
#pragma once
#include "sdhip_common.h"

namespace {

// `bytes` (even; the global side 4-byte aligned) from global to LDS / back, all 256 threads; 16-byte vectors when the global side allows
__device__ __forceinline__ void rows_to_lds(const void* g, void* l, int bytes, int tid) {
  if ((((unsigned long long)g) & 15ull) == 0 && (bytes & 15) == 0) {
    const u32x4* gs = reinterpret_cast<const u32x4*>(g);
    u32x4* ld = reinterpret_cast<u32x4*>(l);
    for (int i = tid; i < (bytes >> 4); i += 256) ld[i] = gs[i];
  } else {
    const unsigned int* gs = reinterpret_cast<const unsigned int*>(g);
    unsigned int* ld = reinterpret_cast<unsigned int*>(l);
    for (int i = tid; i < (bytes >> 2); i += 256) ld[i] = gs[i];
    if ((bytes & 2) && tid == 0)
      reinterpret_cast<unsigned short*>(l)[(bytes >> 1) - 1] = reinterpret_cast<const unsigned short*>(g)[(bytes >> 1) - 1];
  }
}
__device__ __forceinline__ void rows_from_lds(void* g, const void* l, int bytes, int tid) {
  if ((((unsigned long long)g) & 15ull) == 0 && (bytes & 15) == 0) {
    u32x4* gd = reinterpret_cast<u32x4*>(g);
    const u32x4* ls = reinterpret_cast<const u32x4*>(l);
    for (int i = tid; i < (bytes >> 4); i += 256) gd[i] = ls[i];
  } else {
    unsigned int* gd = reinterpret_cast<unsigned int*>(g);
    const unsigned int* ls = reinterpret_cast<const unsigned int*>(l);
    for (int i = tid; i < (bytes >> 2); i += 256) gd[i] = ls[i];
    if ((bytes & 2) && tid == 0)
      reinterpret_cast<unsigned short*>(g)[(bytes >> 1) - 1] = reinterpret_cast<const unsigned short*>(l)[(bytes >> 1) - 1];
  }
}

// LDS bytes of a workgroup's 256 rows of `ld` elements of size es
inline size_t rows_lds_bytes(int ld, int es) { return ((size_t)256 * ld * es + 15) & ~(size_t)15; }

}  // namespace
