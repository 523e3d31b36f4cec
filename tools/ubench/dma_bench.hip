// Micro-benchmark: how fast can one MI355X stream a 256 MB buffer through (a) 16-byte register loads, (b) LDS-DMA
// (global_load_lds_dwordx4) with a counted-wait ring, for several workgroup shapes / depths?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void reg_stream(const u32x4* __restrict__ src, unsigned* out, long n16, int unroll) {
  u32x4 acc = {0, 0, 0, 0};
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  for (; i < n16; i += stride) acc ^= src[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each workgroup streams a contiguous slice in stages of STAGE_KB; NS ring slots; NW waves; with / without a barrier per stage
template <int NW, int STAGE_KB, int NS, bool BARRIER, bool SWZ>
__global__ __launch_bounds__(NW * 64) void dma_stream(const unsigned char* __restrict__ src, unsigned* out, long bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IPS = STAGE_KB / NW;     // 1 KiB instructions per wave per stage
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long per = bytes / gridDim.x;
  const unsigned char* base = src + (long)blockIdx.x * per;
  const int nst = (int)(per / (STAGE_KB * 1024));
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem) + wave * 1024;
  const int lane_off = SWZ ? (((lane & 7) ^ ((lane >> 3) & 6)) * 16 + (lane >> 3) * 128) : lane * 16;
  auto issue = [&](int st) {
    const unsigned char* s = base + (long)st * STAGE_KB * 1024 + wave * 1024 + lane_off;
    const unsigned dst = (st % NS) * STAGE_KB * 1024 + lds0;
#pragma unroll
    for (int j = 0; j < IPS; ++j) glds16(s + j * NW * 1024, __builtin_amdgcn_readfirstlane(dst + j * NW * 1024));
  };
  for (int s = 0; s < NS - 1 && s < nst; ++s) issue(s);
  unsigned acc = 0;
  for (int st = 0; st < nst; ++st) {
    const int younger = nst - 1 - st < NS - 2 ? nst - 1 - st : NS - 2;
    if (younger >= 3) wait_vm<3 * IPS>(); else if (younger == 2) wait_vm<2 * IPS>(); else if (younger == 1) wait_vm<IPS>(); else wait_vm<0>();
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (st + NS - 1 < nst) issue(st + NS - 1);
    acc ^= *reinterpret_cast<const unsigned*>(smem + (st % NS) * STAGE_KB * 1024 + tid * 4);
  }
  if (acc == 0x12345678u) out[0] = 1;
}

template <typename F> float time_it(F f) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) f();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / 10;
}

template <int NW, int STAGE_KB, int NS, bool BARRIER, bool SWZ>
void run_dma(const unsigned char* src, unsigned* out, long bytes, int grid) {
  auto k = dma_stream<NW, STAGE_KB, NS, BARRIER, SWZ>;
  const size_t lds = (size_t)NS * STAGE_KB * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, src, out, bytes); });
  printf("dma  waves %d stage %3d KB ring %d (%3zu KB LDS) barrier %d swz %d grid %4d : %7.1f us  %5.2f TB/s\n", NW, STAGE_KB, NS, lds / 1024, (int)BARRIER,
         (int)SWZ, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const long bytes = 256L << 20;
  unsigned char* src; unsigned* out;
  CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(src, 1, bytes));
  for (int grid : {256, 512, 1024, 2048}) {
    float ms = time_it([&] { hipLaunchKernelGGL(reg_stream, dim3(grid), dim3(512), 0, 0, (const u32x4*)src, out, bytes / 16, 8); });
    printf("reg  16B loads x8 in flight, 512 threads, grid %4d : %7.1f us  %5.2f TB/s\n", grid, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  }
  run_dma<8, 32, 3, true, false>(src, out, bytes, 256);
  run_dma<8, 32, 3, true, true>(src, out, bytes, 256);
  run_dma<8, 32, 4, true, false>(src, out, bytes, 256);
  run_dma<8, 16, 5, true, false>(src, out, bytes, 256);
  run_dma<8, 16, 8, true, false>(src, out, bytes, 256);
  run_dma<8, 32, 3, false, false>(src, out, bytes, 256);
  run_dma<4, 16, 4, true, false>(src, out, bytes, 256);
  run_dma<4, 16, 4, true, false>(src, out, bytes, 512);
  run_dma<4, 16, 4, true, false>(src, out, bytes, 1024);
  run_dma<4, 16, 3, true, false>(src, out, bytes, 768);
  run_dma<8, 32, 2, true, false>(src, out, bytes, 512);
  run_dma<16, 64, 2, true, false>(src, out, bytes, 256);
  run_dma<4, 8, 8, true, false>(src, out, bytes, 512);
  return 0;
}
