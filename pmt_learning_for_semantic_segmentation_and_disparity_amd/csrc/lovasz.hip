// Lovasz-softmax loss (util/lovasz_losses.py:153-199 with classes='present', per_image=False, applied to
// softmax(log_softmax(logits)) = softmax(logits) and labels = argmax(one-hot), losses/multiLosses.py:70-72), forward
// value and gradient w.r.t. the logits in one call.  HBM bound: per class one descending radix sort of (|fg - p|, index)
// over all B*H*W pixels (rocPRIM device primitives), an inclusive scan of the sorted foreground flags, and two
// elementwise passes.  The Jaccard gradient is a constant w.r.t. the errors (as in the reference, which detaches it),
// so d loss / d p_c[i] = -sign(fg - p) * lovasz_grad(fg_sorted)[rank(i)].
// ignore=<void label> (flatten_probas :202-216; cityscapes: losses/multiLosses.py:19-21 drops the 20th one-hot channel and
// passes ignore=19): a pixel whose target row has no positive entry is VOID.  Void pixels get the key -1 (every real error
// is >= 0), so the descending sort parks them behind the nvalid real entries; the Jaccard walk stops at nvalid and they
// receive no gradient — the same as removing them from the flattened arrays.
// ignore=None (roses / garden, losses/multiLosses.py:11-17): nothing is void — the label of an all-zero row is argmax = class 0
// and the pixel counts like any other.  `ignore_void` selects between the two rules.
#include "sdhip_common.h"
#include <cstring>
#include <string.h>
using std::memset;
#include <rocprim/rocprim.hpp>

namespace {

// rocPRIM sorts up to 1 Mi keys with its merge sort and larger inputs with the one-sweep radix sort, which clears its
// histograms with hipMemsetAsync — memset nodes once the step is captured, and those are not reliable under hipGraph
// replay on ROCm 7.2 (sdhip_common.h, sdhip_zero_async).  The merge path launches kernels only: it is used at every size.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)1 << 40>;

struct FgFlag {
  __device__ __host__ unsigned int operator()(unsigned int v) const { return v >> 31; }
};

// keys[c][i] = |fg - p_c|, vals[c][i] = i | fg << 31, counts[c] += fg
template <typename T>
__global__ __launch_bounds__(256) void lovasz_errors_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                            float* __restrict__ keys, unsigned int* __restrict__ vals,
                                                            unsigned int* __restrict__ counts, long npix, int C, int ignore_void) {
  // per-class pixel counts: LDS histogram per workgroup, one global atomic per class per workgroup (a global atomic per
  // pixel on C addresses serialises: ~10 ms for 1M pixels)
  __shared__ unsigned int hist[64];
  __shared__ unsigned int nvoid;
  if (threadIdx.x < 64) hist[threadIdx.x] = 0u;
  if (threadIdx.x == 0) nvoid = 0u;
  __syncthreads();
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const T* yp = y + p * ldy;
    float mx = -INFINITY, tbest = -INFINITY;
    int label = 0;
    for (int c = 0; c < C; ++c) {
      mx = fmaxf(mx, Elem<T>::ld(yp + c));
      const float tv = t[p * ldt + c];
      if (tv > tbest) { tbest = tv; label = c; }   // argmax, first maximum wins (torch.argmax)
    }
    if (ignore_void && !(tbest > 0.f)) {   // void pixel
      for (int c = 0; c < C; ++c) {
        keys[(long)c * npix + p] = -1.f;
        vals[(long)c * npix + p] = (unsigned int)p;
      }
      atomicAdd(&nvoid, 1u);
      continue;
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(Elem<T>::ld(yp + c) - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
      const float pc = __expf(Elem<T>::ld(yp + c) - mx) * inv;
      const unsigned int fg = label == c ? 1u : 0u;
      keys[(long)c * npix + p] = fabsf((float)fg - pc);
      vals[(long)c * npix + p] = (unsigned int)p | (fg << 31);
    }
    if (C <= 64) atomicAdd(hist + label, 1u); else atomicAdd(counts + label, 1u);
  }
  __syncthreads();
  if (C <= 64 && threadIdx.x < C && hist[threadIdx.x]) atomicAdd(counts + threadIdx.x, hist[threadIdx.x]);
  if (threadIdx.x == 0 && nvoid) atomicAdd(counts + C, nvoid);   // counts[C] = number of void pixels
}

// per class (blockIdx.y): g_i = jaccard(i) - jaccard(i-1) over the sorted order; loss_c += err_i * g_i; gerr[c][orig] = g_i
__global__ __launch_bounds__(256) void lovasz_grad_kernel(const float* __restrict__ keys_sorted, const unsigned int* __restrict__ vals_sorted,
                                                          const unsigned int* __restrict__ cum, const unsigned int* __restrict__ counts,
                                                          float* __restrict__ gerr, double* __restrict__ lossc, long npix, int C) {
  __shared__ float sh[4];
  const int c = blockIdx.y;
  const long nvalid = npix - (long)counts[C];
  const double gts = (double)counts[c];
  const float* ks = keys_sorted + (long)c * npix;
  const unsigned int* vs = vals_sorted + (long)c * npix;
  const unsigned int* cu = cum + (long)c * npix;
  float part = 0.f;
  if (gts > 0.) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvalid; i += (long)gridDim.x * 256) {
      const double ci = (double)cu[i];
      const double jac = 1.0 - (gts - ci) / (gts + (double)(i + 1) - ci);
      double prev = 0.0;
      if (i > 0) { const double cp = (double)cu[i - 1]; prev = 1.0 - (gts - cp) / (gts + (double)i - cp); }
      const float gi = (float)(jac - prev);
      part = fmaf(ks[i], gi, part);
      gerr[(long)c * npix + (vs[i] & 0x7fffffffu)] = gi;
    }
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(lossc + c, (double)(sh[0] + sh[1] + sh[2] + sh[3]));
}

// gy[p,k] += w/n_present * p_k * (g_k - sum_c g_c p_c), g_c = -sign(fg_c - p_c) * gerr[c][p]  (present classes only)
template <typename T>
__global__ __launch_bounds__(256) void lovasz_backward_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                              const float* __restrict__ gerr, const unsigned int* __restrict__ counts,
                                                              const double* __restrict__ lossc, T* __restrict__ gy, int ldg,
                                                              double* __restrict__ loss, long npix, int C, float weight, int ignore_void) {
  int npres = 0;
  for (int c = 0; c < C; ++c) npres += counts[c] > 0u ? 1 : 0;
  const float w = weight / (float)(npres > 0 ? npres : 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double tot = 0.;
    for (int c = 0; c < C; ++c) if (counts[c] > 0u) tot += lossc[c];
    atomicAdd(loss, tot * (double)w);
  }
  if (!gy) return;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const T* yp = y + p * ldy;
    float mx = -INFINITY, tbest = -INFINITY;
    int label = 0;
    for (int c = 0; c < C; ++c) {
      mx = fmaxf(mx, Elem<T>::ld(yp + c));
      const float tv = t[p * ldt + c];
      if (tv > tbest) { tbest = tv; label = c; }
    }
    if (ignore_void && !(tbest > 0.f)) continue;   // void pixel: not part of the loss
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(Elem<T>::ld(yp + c) - mx);
    const float inv = 1.f / se;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      if (counts[c] == 0u) continue;
      const float pc = __expf(Elem<T>::ld(yp + c) - mx) * inv;
      const float diff = (label == c ? 1.f : 0.f) - pc;
      const float g = (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) * gerr[(long)c * npix + p];
      dot = fmaf(g, pc, dot);
    }
    for (int k = 0; k < C; ++k) {
      const float pk = __expf(Elem<T>::ld(yp + k) - mx) * inv;
      float g = 0.f;
      if (counts[k] > 0u) {
        const float diff = (label == k ? 1.f : 0.f) - pk;
        g = (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) * gerr[(long)k * npix + p];
      }
      T* dst = gy + p * ldg + k;
      Elem<T>::st(dst, Elem<T>::ld(dst) + w * pk * (g - dot));
    }
  }
}

struct Layout {
  size_t keys_in, keys_out, vals_in, vals_out, cum, gerr, counts, lossc, temp, temp_bytes, total;
};

Layout layout(long npix, int C) {
  Layout L;
  const size_t n = (size_t)npix * C;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  L.keys_in = off; off = al(off + n * 4);
  L.keys_out = off; off = al(off + n * 4);
  L.vals_in = off; off = al(off + n * 4);
  L.vals_out = off; off = al(off + n * 4);
  L.cum = off; off = al(off + n * 4);
  L.gerr = off; off = al(off + n * 4);
  L.counts = off; off = al(off + (size_t)(C + 1) * 4);
  L.lossc = off; off = al(off + (size_t)C * 8);
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_pairs_desc<SortConfig>(nullptr, t1, (const float*)nullptr, (float*)nullptr, (const unsigned int*)nullptr,
                                 (unsigned int*)nullptr, (size_t)npix);
  auto it = rocprim::make_transform_iterator((const unsigned int*)nullptr, FgFlag());
  (void)rocprim::inclusive_scan(nullptr, t2, it, (unsigned int*)nullptr, (size_t)npix, rocprim::plus<unsigned int>());
  L.temp_bytes = t1 > t2 ? t1 : t2;
  L.temp = off; off = al(off + L.temp_bytes);
  L.total = off;
  return L;
}

// every kernel here ends with a few atomics per workgroup on the same counters (class counts, per-class f64 sums): 2048
// workgroups queued ~30 us of same-address atomics behind 10 us of streaming, so the grid is kept at two per CU
inline dim3 grid_for(long items) {
  long b = (items + 255) / 256;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

inline dim3 grid_stream(long items) {   // no closing atomics: fill the chip
  long b = (items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

}  // namespace

extern "C" long sdhip_lovasz_workspace_bytes(long npix, int C) {
  if (npix <= 0 || C <= 0) return 0;
  return (long)layout(npix, C).total;
}

extern "C" int sdhip_lovasz_softmax(const void* logits, int ldy, const float* target, int ldt, void* grad, int ldg,
                                    double* loss, long npix, int C, float weight, void* workspace, long workspace_bytes,
                                    int ignore_void, int dtype, void* stream) {
  SDHIP_CHECK_ARG(logits && target && loss && workspace && npix > 0 && C > 0 && ldy >= C && ldt >= C && (!grad || ldg >= C),
                  "lovasz_softmax: bad arguments");
  SDHIP_CHECK_ARG(npix < (1L << 31), "lovasz_softmax: more than 2^31 pixels");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "lovasz_softmax: unknown dtype %d", dtype);
  const Layout L = layout(npix, C);
  SDHIP_CHECK_ARG((size_t)workspace_bytes >= L.total, "lovasz_softmax: workspace too small (%ld < %zu)", workspace_bytes, L.total);
  hipStream_t s = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  float* keys_in = (float*)(ws + L.keys_in); float* keys_out = (float*)(ws + L.keys_out);
  unsigned int* vals_in = (unsigned int*)(ws + L.vals_in); unsigned int* vals_out = (unsigned int*)(ws + L.vals_out);
  unsigned int* cum = (unsigned int*)(ws + L.cum);
  float* gerr = (float*)(ws + L.gerr);
  unsigned int* counts = (unsigned int*)(ws + L.counts);
  double* lossc = (double*)(ws + L.lossc);
  // (a kernel, not hipMemsetAsync: the step is replayed from a hipGraph, and everything in it is kept to kernel nodes)
  if (sdhip_zero_async(ws + L.counts, L.temp - L.counts, s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "lovasz_softmax: clearing the counters failed");
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(lovasz_errors_kernel<float>, grid_for(npix), dim3(256), 0, s, (const float*)logits, ldy, target, ldt, keys_in, vals_in, counts, npix, C, ignore_void);
  else
    hipLaunchKernelGGL(lovasz_errors_kernel<bf16_t>, grid_for(npix), dim3(256), 0, s, (const bf16_t*)logits, ldy, target, ldt, keys_in, vals_in, counts, npix, C, ignore_void);
  for (int c = 0; c < C; ++c) {
    size_t tb = L.temp_bytes;
    const size_t o = (size_t)c * npix;
    if (rocprim::radix_sort_pairs_desc<SortConfig>(ws + L.temp, tb, keys_in + o, keys_out + o, vals_in + o, vals_out + o, (size_t)npix, 0, 32, s) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "lovasz_softmax: radix sort failed");
    tb = L.temp_bytes;
    auto it = rocprim::make_transform_iterator((const unsigned int*)(vals_out + o), FgFlag());
    if (rocprim::inclusive_scan(ws + L.temp, tb, it, cum + o, (size_t)npix, rocprim::plus<unsigned int>(), s) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "lovasz_softmax: scan failed");
  }
  dim3 g2 = grid_for(npix); g2.y = C;
  hipLaunchKernelGGL(lovasz_grad_kernel, g2, dim3(256), 0, s, keys_out, vals_out, cum, counts, gerr, lossc, npix, C);
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(lovasz_backward_kernel<float>, grid_stream(npix), dim3(256), 0, s, (const float*)logits, ldy, target, ldt, gerr, counts, lossc, (float*)grad, ldg, loss, npix, C, weight, ignore_void);
  else
    hipLaunchKernelGGL(lovasz_backward_kernel<bf16_t>, grid_stream(npix), dim3(256), 0, s, (const bf16_t*)logits, ldy, target, ldt, gerr, counts, lossc, (bf16_t*)grad, ldg, loss, npix, C, weight, ignore_void);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
