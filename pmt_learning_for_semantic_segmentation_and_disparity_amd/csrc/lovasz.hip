// Lovasz-softmax loss (util/lovasz_losses.py:153-199 with classes='present', per_image=False, applied to
// softmax(log_softmax(logits)) = softmax(logits) and labels = argmax(one-hot), losses/multiLosses.py:70-72), forward
// value and gradient w.r.t. the logits in one call.  HBM bound: ONE segmented descending sort of (|fg - p|, index) — a
// segment per class, all B*H*W pixels each — a prefix count of the sorted foreground flags, and two elementwise passes.
// The Jaccard gradient is a constant w.r.t. the errors (as in the reference, which detaches it), so
// d loss / d p_c[i] = -sign(fg - p) * lovasz_grad(fg_sorted)[rank(i)].
// ignore=<void label> (flatten_probas :202-216; cityscapes: losses/multiLosses.py:19-21 drops the 20th one-hot channel and
// passes ignore=19): a pixel whose target row has no positive entry is VOID.  Void pixels get the largest key, so the sort
// parks them behind the nvalid real entries; the Jaccard walk stops at nvalid and they receive no gradient — the same as
// removing them from the flattened arrays.
// ignore=None (roses / garden, losses/multiLosses.py:11-17): nothing is void — the label of an all-zero row is argmax = class 0
// and the pixel counts like any other.  `ignore_void` selects between the two rules.
//
// The sort (the step's only torch.sort): a least-significant-digit radix sort written for this use, every
// class in the same launches (grid.y = class) — 12 launches per loss where a library sort per class took 26 x C (494 of
// a 19-class step's 1595 kernels, 5.0 ms), and kernels only (the library's one-sweep path clears its histograms with
// memset nodes, which hipGraph replay on ROCm 7.2 does not honour reliably).
//   key    0x3F800000 - bits(err), err in [0, 1]: ascending key = descending error; void = 0x3F800001.  30 significant bits:
//          four passes of 8 bits.
//   unit   a WAVE owns 2048 consecutive elements (32 rounds of 64) and a column of the digit table [class][digit][column]:
//          sort_hist counts, sort_scan turns the table into exclusive offsets (digit-major, then column: a stable order),
//          sort_scatter ranks each round's lanes among their equal-digit peers (8 ballots -> peer mask -> popcount of the
//          lanes below) and adds the column's running offset.  No barrier inside the round loop, nothing shared between waves:
//          the result is the same on every run.
#include "sdhip_common.h"
#include "rows_lds.h"

namespace {

constexpr unsigned int kKeyOne = 0x3F800000u;       // key of err = 0 is kKeyOne, of err = 1 is 0
constexpr unsigned int kKeyVoid = 0x3F800001u;
constexpr int kSortKPW = 2048;                       // elements per wave
constexpr int kSortRounds = kSortKPW / 64;
constexpr int kChunk = 4096;                         // sorted elements per workgroup of the Jaccard walk

__device__ __forceinline__ unsigned int wave_incl_scan(unsigned int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}

// bh[class][digit][column] = number of elements of the column's 2048 whose digit (bits shift .. shift+7 of the key) is `digit`
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint2* __restrict__ kv, unsigned int* __restrict__ bh,
                                                        long npix, int cols, int shift) {
  __shared__ unsigned int h[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.y;
  const int col = blockIdx.x * 4 + wave;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[wave][lane * 4 + j] = 0u;
  __syncthreads();
  if (col < cols) {
    const uint2* k = kv + (long)c * npix;
    const long base = (long)col * kSortKPW;
    for (int r0 = 0; r0 < kSortRounds; r0 += 8) {     // eight loads in flight
      unsigned int kk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long i = base + (r0 + j) * 64 + lane;
        kk[j] = i < npix ? k[i].x : 0u;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long i = base + (r0 + j) * 64 + lane;
        if (i < npix) atomicAdd(&h[wave][(kk[j] >> shift) & 255u], 1u);
      }
    }
  }
  __syncthreads();
  if (col < cols) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = lane * 4 + j;
      bh[((long)c * 256 + d) * cols + col] = h[wave][d];
    }
  }
}

// one wave per (class, digit) row of the table: the row's counts -> exclusive prefix over the columns (in place), the row's
// total -> tot[class][digit].  (The prefix over the digits is 256 values: every scatter wave forms it itself.)
__global__ __launch_bounds__(256) void sort_scan_kernel(unsigned int* __restrict__ bh, unsigned int* __restrict__ tot, int cols, int nrows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;              // class * 256 + digit
  if (row >= nrows) return;
  unsigned int* t = bh + (long)row * cols;
  unsigned int carry = 0u;
  for (int i0 = 0; i0 < cols; i0 += 64) {
    const int i = i0 + lane;
    const unsigned int v = i < cols ? t[i] : 0u;
    const unsigned int inc = wave_incl_scan(v, lane);
    if (i < cols) t[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) tot[row] = carry;
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint2* __restrict__ kvin, uint2* __restrict__ kvout,
                                                           const unsigned int* __restrict__ bh, const unsigned int* __restrict__ tot,
                                                           long npix, int cols, int shift) {
  __shared__ unsigned int off[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.y;
  const int col = blockIdx.x * 4 + wave;
  if (col >= cols) return;                           // (whole waves; no workgroup barrier below)
  {
    // digit d starts at the sum of the totals of the digits below it, plus this column's share of the digit's row
    unsigned int tv[4], sum = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) { tv[j] = tot[c * 256 + lane * 4 + j]; sum += tv[j]; }
    unsigned int run = wave_incl_scan(sum, lane) - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = lane * 4 + j;
      off[wave][d] = run + bh[((long)c * 256 + d) * cols + col];
      run += tv[j];
    }
  }
  const long seg = (long)c * npix, base = (long)col * kSortKPW;
  const unsigned long long below = (1ull << lane) - 1ull;
  constexpr int G = 8;                               // rounds whose loads are in flight together
  uint2 kb[2][G];
  auto fetch = [&](int g, uint2 (&k)[G]) {
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const long i = base + (g * G + j) * 64 + lane;
      k[j] = i < npix ? kvin[seg + i] : make_uint2(0u, 0u);
    }
  };
  fetch(0, kb[0]);
#pragma unroll
  for (int g = 0; g < kSortRounds / G; ++g) {
    if (g + 1 < kSortRounds / G) fetch(g + 1, kb[(g + 1) & 1]);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const long i = base + (g * G + j) * 64 + lane;
      const bool valid = i < npix;
      const unsigned int k = kb[g & 1][j].x;
      const unsigned int d = (k >> shift) & 255u;
      unsigned long long peers = __ballot(valid);    // lanes of this round with the same digit
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
      }
      const unsigned int rank = (unsigned int)__popcll(peers & below), cnt = (unsigned int)__popcll(peers);
      volatile unsigned int* const o = &off[wave][d];  // (every lane reads before the group's last lane writes: one wave, in order)
      const unsigned int pos = *o + rank;
      if (valid && rank + 1u == cnt) *o = pos + 1u;
      if (valid) kvout[seg + pos] = kb[g & 1][j];
    }
  }
}

// foreground elements of every chunk of kChunk sorted positions: bfg[class][chunk]
__global__ __launch_bounds__(256) void lovasz_fgcount_kernel(const uint2* __restrict__ kv_sorted, unsigned int* __restrict__ bfg,
                                                             long npix, int nchunk) {
  __shared__ unsigned int sh[4];
  const int c = blockIdx.y;
  const long i0 = (long)blockIdx.x * kChunk;
  const uint2* vs = kv_sorted + (long)c * npix;
  unsigned int n = 0u;
  for (int j = 0; j < kChunk / 256; ++j) {
    const long i = i0 + j * 256 + threadIdx.x;
    if (i < npix) n += vs[i].y >> 31;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) bfg[(long)c * nchunk + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// kv[c][i] = (sort key of |fg - p_c| (see above), i | fg << 31), counts[c] += fg
// ROWS: the workgroup's 256 pixel rows of logits and targets travel through LDS (rows_lds.h; many classes)
template <typename T, bool ROWS>
__global__ __launch_bounds__(256) void lovasz_errors_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                            uint2* __restrict__ kv,
                                                            unsigned int* __restrict__ counts, long npix, int C, int ignore_void, int off_t) {
  // per-class pixel counts: LDS histogram per workgroup, one global atomic per class per workgroup (a global atomic per
  // pixel on C addresses serialises: ~10 ms for 1M pixels)
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  __shared__ unsigned int hist[64];
  __shared__ unsigned int nvoid;
  if (threadIdx.x < 64) hist[threadIdx.x] = 0u;
  if (threadIdx.x == 0) nvoid = 0u;
  __syncthreads();
  auto pixel = [&](long p, const T* yp, const float* tp) {
    float mx = -INFINITY, tbest = -INFINITY;
    int label = 0;
    for (int c = 0; c < C; ++c) {
      mx = fmaxf(mx, Elem<T>::ld(yp + c));
      const float tv = tp[c];
      if (tv > tbest) { tbest = tv; label = c; }   // argmax, first maximum wins (torch.argmax)
    }
    if (ignore_void && !(tbest > 0.f)) {   // void pixel
      for (int c = 0; c < C; ++c) kv[(long)c * npix + p] = make_uint2(kKeyVoid, (unsigned int)p);
      atomicAdd(&nvoid, 1u);
      return;
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(Elem<T>::ld(yp + c) - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
      const float pc = __expf(Elem<T>::ld(yp + c) - mx) * inv;
      const unsigned int fg = label == c ? 1u : 0u;
      kv[(long)c * npix + p] = make_uint2(kKeyOne - __float_as_uint(fminf(fabsf((float)fg - pc), 1.f)), (unsigned int)p | (fg << 31));
    }
    if (C <= 64) atomicAdd(hist + label, 1u); else atomicAdd(counts + label, 1u);
  };
  if constexpr (ROWS) {
    T* const ly = reinterpret_cast<T*>(rsm);
    float* const lt = reinterpret_cast<float*>(rsm + off_t);
    const long ntiles = (npix + 255) / 256;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long p0 = tile * 256;
      const int n = (int)min(256L, npix - p0);
      rows_to_lds(y + p0 * ldy, ly, n * ldy * (int)sizeof(T), threadIdx.x);
      rows_to_lds(t + p0 * ldt, lt, n * ldt * 4, threadIdx.x);
      __syncthreads();
      if ((int)threadIdx.x < n) pixel(p0 + threadIdx.x, ly + threadIdx.x * ldy, lt + threadIdx.x * ldt);
      __syncthreads();
    }
  } else {
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) pixel(p, y + p * ldy, t + p * ldt);
  }
  __syncthreads();
  if (C <= 64 && threadIdx.x < C && hist[threadIdx.x]) atomicAdd(counts + threadIdx.x, hist[threadIdx.x]);
  if (threadIdx.x == 0 && nvoid) atomicAdd(counts + C, nvoid);   // counts[C] = number of void pixels
}

// per class (blockIdx.y) and chunk of kChunk sorted positions (blockIdx.x): cum_i = foreground elements among positions
// 0..i (the chunks before this one from bfg, inside the chunk a scan: 16 consecutive positions per thread), g_i = jaccard(i) -
// jaccard(i-1); loss_c += err_i * g_i; gerr[c][orig] = g_i
__global__ __launch_bounds__(256) void lovasz_grad_kernel(const uint2* __restrict__ kv_sorted,
                                                          const unsigned int* __restrict__ bfg, const unsigned int* __restrict__ counts,
                                                          float* __restrict__ gerr, double* __restrict__ lossc, long npix, int C, int nchunk) {
  __shared__ unsigned int shu[4];
  __shared__ float sh[4];
  const int c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nvalid = npix - (long)counts[C];
  const double gts = (double)counts[c];
  if (!(gts > 0.) || (long)blockIdx.x * kChunk >= nvalid) return;          // class absent / only void entries here (uniform)
  const uint2* kvs = kv_sorted + (long)c * npix;
  // foreground elements in the chunks before this one
  unsigned int before = 0u;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) before += bfg[(long)c * nchunk + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) shu[wave] = before;
  __syncthreads();
  before = shu[0] + shu[1] + shu[2] + shu[3];
  __syncthreads();
  // this thread's 16 consecutive positions
  constexpr int PT = kChunk / 256;
  const long i0 = (long)blockIdx.x * kChunk + (long)threadIdx.x * PT;
  unsigned int v[PT], k[PT], mine = 0u;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const long i = i0 + j;
    const uint2 e = i < nvalid ? kvs[i] : make_uint2(kKeyOne, 0u);
    v[j] = e.y; k[j] = e.x;
    mine += v[j] >> 31;
  }
  const unsigned int inc = wave_incl_scan(mine, lane);
  if (lane == 63) shu[wave] = inc;
  __syncthreads();
  unsigned int cum = before + inc - mine;                                  // foreground elements before position i0
  for (int w = 0; w < wave; ++w) cum += shu[w];
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const long i = i0 + j;
    const double cp = (double)cum;                                         // cum_{i-1}
    cum += v[j] >> 31;
    if (i < nvalid) {
      const double ci = (double)cum;
      const double jac = 1.0 - (gts - ci) / (gts + (double)(i + 1) - ci);
      const double prev = i > 0 ? 1.0 - (gts - cp) / (gts + (double)i - cp) : 0.0;
      const float gi = (float)(jac - prev);
      part = fmaf(__uint_as_float(kKeyOne - k[j]), gi, part);
      gerr[(long)c * npix + (v[j] & 0x7fffffffu)] = gi;
    }
  }
  part = wave_sum(part);
  if (lane == 0) sh[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(lossc + c, (double)(sh[0] + sh[1] + sh[2] + sh[3]));
}

// gy[p,k] += w/n_present * p_k * (g_k - sum_c g_c p_c), g_c = -sign(fg_c - p_c) * gerr[c][p]  (present classes only)
// CM > 0: C <= CM, the pixel's probabilities and gradient factors stay in registers (one exp per class instead of three)
template <typename T, int CM, bool ROWS = false>
__global__ __launch_bounds__(256) void lovasz_backward_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ t, int ldt,
                                                              const float* __restrict__ gerr, const unsigned int* __restrict__ counts,
                                                              const double* __restrict__ lossc, T* __restrict__ gy, int ldg,
                                                              double* __restrict__ loss, long npix, int C, float weight, int ignore_void,
                                                              int off_t = 0, int off_g = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  int npres = 0;
  unsigned long long present = 0ull;                  // (CM <= 64)
  for (int c = 0; c < C; ++c)
    if (counts[c] > 0u) { ++npres; if (c < 64) present |= 1ull << c; }
  const float w = weight / (float)(npres > 0 ? npres : 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double tot = 0.;
    for (int c = 0; c < C; ++c) if (counts[c] > 0u) tot += lossc[c];
    atomicAdd(loss, tot * (double)w);
  }
  if (!gy) return;
  if constexpr (ROWS) {
    // the workgroup's 256 pixel rows of logits, targets and the gradient (read, added to, written back) through LDS (rows_lds.h)
    static_assert(CM > 0, "rows path keeps the pixel in registers");
    T* const ly = reinterpret_cast<T*>(rsm);
    float* const lt = reinterpret_cast<float*>(rsm + off_t);
    T* const lg = reinterpret_cast<T*>(rsm + off_g);
    const long ntiles = (npix + 255) / 256;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long p0 = tile * 256;
      const int n = (int)min(256L, npix - p0);
      rows_to_lds(y + p0 * ldy, ly, n * ldy * (int)sizeof(T), threadIdx.x);
      rows_to_lds(t + p0 * ldt, lt, n * ldt * 4, threadIdx.x);
      rows_to_lds(gy + p0 * ldg, lg, n * ldg * (int)sizeof(T), threadIdx.x);
      __syncthreads();
      if ((int)threadIdx.x < n) {
        const long p = p0 + threadIdx.x;
        const T* yp = ly + threadIdx.x * ldy;
        const float* tp = lt + threadIdx.x * ldt;
        float pc[CM], g[CM];
        float mx = -INFINITY, tbest = -INFINITY;
        int label = 0;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
          pc[c] = c < C ? Elem<T>::ld(yp + c) : -INFINITY;
          mx = fmaxf(mx, pc[c]);
          const float tv = c < C ? tp[c] : -INFINITY;
          if (tv > tbest) { tbest = tv; label = c; }
        }
        if (!(ignore_void && !(tbest > 0.f))) {        // (void pixel: not part of the loss)
          float se = 0.f;
#pragma unroll
          for (int c = 0; c < CM; ++c) { pc[c] = c < C ? __expf(pc[c] - mx) : 0.f; se += pc[c]; }
          const float inv = 1.f / se;
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < CM; ++c) {
            pc[c] *= inv;
            g[c] = 0.f;
            if (c < C && ((present >> c) & 1ull)) {
              const float diff = (label == c ? 1.f : 0.f) - pc[c];
              g[c] = (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) * gerr[(long)c * npix + p];
              dot = fmaf(g[c], pc[c], dot);
            }
          }
          T* gp = lg + threadIdx.x * ldg;
#pragma unroll
          for (int k = 0; k < CM; ++k)
            if (k < C) Elem<T>::st(gp + k, Elem<T>::ld(gp + k) + w * pc[k] * (g[k] - dot));
        }
      }
      __syncthreads();
      rows_from_lds(gy + p0 * ldg, lg, n * ldg * (int)sizeof(T), threadIdx.x);
      __syncthreads();
    }
    return;
  }
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const T* yp = y + p * ldy;
    if constexpr (CM > 0) {
      float pc[CM], g[CM];
      float mx = -INFINITY, tbest = -INFINITY;
      int label = 0;
#pragma unroll
      for (int c = 0; c < CM; ++c) {
        pc[c] = c < C ? Elem<T>::ld(yp + c) : -INFINITY;
        mx = fmaxf(mx, pc[c]);
        const float tv = c < C ? t[p * ldt + c] : -INFINITY;
        if (tv > tbest) { tbest = tv; label = c; }
      }
      if (ignore_void && !(tbest > 0.f)) continue;   // void pixel: not part of the loss
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c) { pc[c] = c < C ? __expf(pc[c] - mx) : 0.f; se += pc[c]; }
      const float inv = 1.f / se;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c) {
        pc[c] *= inv;
        g[c] = 0.f;
        if (c < C && ((present >> c) & 1ull)) {
          const float diff = (label == c ? 1.f : 0.f) - pc[c];
          g[c] = (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) * gerr[(long)c * npix + p];
          dot = fmaf(g[c], pc[c], dot);
        }
      }
#pragma unroll
      for (int k = 0; k < CM; ++k) {
        if (k < C) {
          T* dst = gy + p * ldg + k;
          Elem<T>::st(dst, Elem<T>::ld(dst) + w * pc[k] * (g[k] - dot));
        }
      }
    } else {
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
}

template <typename T>
void launch_lovasz_backward(dim3 grid, hipStream_t s, const T* y, int ldy, const float* t, int ldt, const float* gerr, const unsigned int* counts,
                            const double* lossc, T* gy, int ldg, double* loss, long npix, int C, float weight, int ignore_void) {
  const size_t by = rows_lds_bytes(ldy, sizeof(T)), bt = rows_lds_bytes(ldt, 4), bg = rows_lds_bytes(ldg, sizeof(T));
  if (C > 4 && C <= 32 && gy && by + bt + bg <= 60 * 1024 && ((uintptr_t)y & 3) == 0 && ((uintptr_t)gy & 3) == 0) {
    long b = (npix + 255) / 256;
    if (b > 768) b = 768;
    hipLaunchKernelGGL((lovasz_backward_kernel<T, 32, true>), dim3((unsigned)b), dim3(256), by + bt + bg, s, y, ldy, t, ldt, gerr, counts, lossc, gy, ldg, loss,
                       npix, C, weight, ignore_void, (int)by, (int)(by + bt));
    return;
  }
  if (C <= 4) hipLaunchKernelGGL((lovasz_backward_kernel<T, 4>), grid, dim3(256), 0, s, y, ldy, t, ldt, gerr, counts, lossc, gy, ldg, loss, npix, C, weight, ignore_void, 0, 0);
  else if (C <= 32) hipLaunchKernelGGL((lovasz_backward_kernel<T, 32>), grid, dim3(256), 0, s, y, ldy, t, ldt, gerr, counts, lossc, gy, ldg, loss, npix, C, weight, ignore_void, 0, 0);
  else hipLaunchKernelGGL((lovasz_backward_kernel<T, 0>), grid, dim3(256), 0, s, y, ldy, t, ldt, gerr, counts, lossc, gy, ldg, loss, npix, C, weight, ignore_void, 0, 0);
}

template <typename T>
void launch_lovasz_errors(hipStream_t s, const T* y, int ldy, const float* t, int ldt, uint2* kv, unsigned int* counts, long npix, int C, int ignore_void) {
  const size_t by = rows_lds_bytes(ldy, sizeof(T)), bt = rows_lds_bytes(ldt, 4);
  long b = (npix + 255) / 256;
  if (C > 4 && by + bt <= 60 * 1024 && ((uintptr_t)y & 3) == 0) {
    if (b > 768) b = 768;
    hipLaunchKernelGGL((lovasz_errors_kernel<T, true>), dim3((unsigned)b), dim3(256), by + bt, s, y, ldy, t, ldt, kv, counts, npix, C, ignore_void, (int)by);
    return;
  }
  if (b > 512) b = 512;     // (closing atomics per workgroup: see grid_for)
  hipLaunchKernelGGL((lovasz_errors_kernel<T, false>), dim3((unsigned)b), dim3(256), 0, s, y, ldy, t, ldt, kv, counts, npix, C, ignore_void, 0);
}

struct Layout {
  size_t kv_a, kv_b, gerr, counts, lossc, bfg, tot, table, total;
  int cols, nchunk;
};

Layout layout(long npix, int C) {
  Layout L;
  const size_t n = (size_t)npix * C;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.cols = (int)((npix + kSortKPW - 1) / kSortKPW);
  L.nchunk = (int)((npix + kChunk - 1) / kChunk);
  size_t off = 0;
  L.kv_a = off; off = al(off + n * 8);
  L.kv_b = off; off = al(off + n * 8);
  L.gerr = off; off = al(off + n * 4);
  L.counts = off; off = al(off + (size_t)(C + 1) * 4);
  L.lossc = off; off = al(off + (size_t)C * 8);
  L.bfg = off; off = al(off + (size_t)C * L.nchunk * 4);
  L.tot = off; off = al(off + (size_t)C * 256 * 4);
  L.table = off; off = al(off + (size_t)C * 256 * L.cols * 4);
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
  uint2* kv_a = (uint2*)(ws + L.kv_a); uint2* kv_b = (uint2*)(ws + L.kv_b);       // (key, index | fg << 31) pairs: one 8-byte store per element and pass
  float* gerr = (float*)(ws + L.gerr);
  unsigned int* counts = (unsigned int*)(ws + L.counts);
  double* lossc = (double*)(ws + L.lossc);
  unsigned int* bfg = (unsigned int*)(ws + L.bfg);
  unsigned int* table = (unsigned int*)(ws + L.table);
  unsigned int* tot = (unsigned int*)(ws + L.tot);
  SDHIP_CHECK_ARG(C <= 65535, "lovasz_softmax: more than 65535 classes");
  // (a kernel, not hipMemsetAsync: the step is replayed from a hipGraph, and everything in it is kept to kernel nodes)
  if (sdhip_zero_async(ws + L.counts, L.bfg - L.counts, s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "lovasz_softmax: clearing the counters failed");
  if (dtype == SDHIP_F32) launch_lovasz_errors<float>(s, (const float*)logits, ldy, target, ldt, kv_a, counts, npix, C, ignore_void);
  else launch_lovasz_errors<bf16_t>(s, (const bf16_t*)logits, ldy, target, ldt, kv_a, counts, npix, C, ignore_void);
  // four stable passes over bits 0-7, 8-15, 16-23, 24-31 (keys < 2^30); a -> b -> a -> b -> a: the sorted arrays end in `a`
  const dim3 gs((unsigned)((L.cols + 3) / 4), (unsigned)C);
  for (int pass = 0; pass < 4; ++pass) {
    const uint2* kin = (pass & 1) ? kv_b : kv_a;
    uint2* kout = (pass & 1) ? kv_a : kv_b;
    hipLaunchKernelGGL(sort_hist_kernel, gs, dim3(256), 0, s, kin, table, npix, L.cols, 8 * pass);
    hipLaunchKernelGGL(sort_scan_kernel, dim3((unsigned)(C * 64)), dim3(256), 0, s, table, tot, L.cols, C * 256);
    hipLaunchKernelGGL(sort_scatter_kernel, gs, dim3(256), 0, s, kin, kout, (const unsigned int*)table, (const unsigned int*)tot, npix, L.cols, 8 * pass);
  }
  const dim3 gc((unsigned)L.nchunk, (unsigned)C);
  hipLaunchKernelGGL(lovasz_fgcount_kernel, gc, dim3(256), 0, s, (const uint2*)kv_a, bfg, npix, L.nchunk);
  hipLaunchKernelGGL(lovasz_grad_kernel, gc, dim3(256), 0, s, (const uint2*)kv_a, (const unsigned int*)bfg,
                     (const unsigned int*)counts, gerr, lossc, npix, C, L.nchunk);
  if (dtype == SDHIP_F32)
    launch_lovasz_backward<float>(grid_stream(npix), s, (const float*)logits, ldy, target, ldt, gerr, counts, lossc, (float*)grad, ldg, loss, npix, C, weight, ignore_void);
  else
    launch_lovasz_backward<bf16_t>(grid_stream(npix), s, (const bf16_t*)logits, ldy, target, ldt, gerr, counts, lossc, (bf16_t*)grad, ldg, loss, npix, C, weight, ignore_void);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
