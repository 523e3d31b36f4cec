// On-device step metrics for gfx950.
//
// Replaces the numpy / sklearn block that the reference runs on the host after every training and validation step
// (three device->host copies of full-resolution maps, a bincount and four sklearn scores per step):
//   SegAccuracyNp        util/utilTorchLoss.py:221-236   (called from losses/multiLosses.py:120)
//   GetSegMetricsNp      util/utilTorchLoss.py:251-303   (losses/multiLosses.py:122)
//   unnormalizedErrorNP  util/utilTorchLoss.py:363-370   (losses/multiLosses.py:150)
//   GetDispMetricsNp     util/utilTorchLoss.py:318-343   (losses/multiLosses.py:152)
// One pass over the network outputs and the targets produces integer counters and f64 sums; the scores the reference
// reports are ratios of those (host side: metrics.py).  The counters are ACCUMULATED, so an epoch's confusion matrix
// (util/torch_implementation.py sums the per-step matrices) needs one device->host copy of a few hundred bytes.
//
// HBM-bound: every input byte is read exactly once (L logits + Ct target values + 2 disparities per pixel).
#include "sdhip_common.h"

namespace {

constexpr int kMaxL = 32;           // up to 32 classes: the L*L histogram lives in LDS
constexpr int kNC = SDHIP_METRIC_COUNTS, kNS = SDHIP_METRIC_SUMS;

__device__ __forceinline__ long wave_sum_l(long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One workgroup walks 256-pixel tiles.  With more than a few channels per pixel a lane-per-pixel read touches one cache
// line per lane and instruction (the texture-address unit, not HBM, then sets the pace: 1.5 TB/s measured at L=19), so
// STAGE copies the tile's logits and targets lane-linearly (fully coalesced) into LDS first and the per-pixel scan
// reads from there.  With 2 classes (roses) the direct reads are already coalesced and staging is skipped.
template <typename T, bool STAGE>
__global__ __launch_bounds__(256) void step_metrics_kernel(
    const T* __restrict__ seg, int lds, const float* __restrict__ tgt, int ldt, int Ct, const T* __restrict__ disp,
    const float* __restrict__ dtgt, unsigned long long* __restrict__ counts, double* __restrict__ sums, long npix, long hw,
    int L, float max_disp, int mask_invalid, int nrep, int rep_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ unsigned int hist[kMaxL * kMaxL];
  __shared__ long wc[4][kNC];
  __shared__ double wsum[4][kNS];
  float* stg = reinterpret_cast<float*>(smem);                      // [256][ldt] targets
  T* slg = reinterpret_cast<T*>(smem + (size_t)256 * ldt * 4);      // [256][lds] logits
  for (int i = threadIdx.x; i < L * L; i += 256) hist[i] = 0;
  __syncthreads();

  long c[kNC];
  double s[kNS];
#pragma unroll
  for (int i = 0; i < kNC; ++i) c[i] = 0;
#pragma unroll
  for (int i = 0; i < kNS; ++i) s[i] = 0.0;

  const bool small = L * L <= 16;
  const int lane = threadIdx.x & 63;
  unsigned int hreg = 0;
  for (long t0 = (long)blockIdx.x * 256; t0 < npix; t0 += (long)gridDim.x * 256) {
    const long p = t0 + threadIdx.x;
    const bool valid = p < npix;                   // tail lanes stay in the loop: lane k owns bin k of the ballots below
    const bool first = valid && p < hw;            // the single-image scores look at image 0 only (`gt[0][1]`, `outputs[0][1]`)
    const T* sp = seg + p * lds;
    const float* tp = tgt + p * ldt;
    if constexpr (STAGE) {
      const long n = npix - t0 < 256 ? npix - t0 : 256;
      __syncthreads();                              // the previous tile has been consumed
      if (tgt) for (long i = threadIdx.x; i < n * ldt; i += 256) stg[i] = tgt[t0 * ldt + i];
      if (seg) for (long i = threadIdx.x; i < n * lds; i += 256) slg[i] = seg[t0 * lds + i];
      __syncthreads();
      sp = slg + (long)threadIdx.x * lds;
      tp = stg + (long)threadIdx.x * ldt;
    }
    float l1 = 0.f, g1 = 0.f;
    int bin = -1;
    if (seg && valid) {
      // argmax with the first maximum winning, as numpy.argmax (NaN wins too)
      float bv = Elem<T>::ld(sp);
      int pred = 0;
      l1 = L > 1 ? Elem<T>::ld(sp + 1) : 0.f;
      for (int k = 1; k < L; ++k) {
        const float v = Elem<T>::ld(sp + k);
        if ((v > bv || v != v) && bv == bv) { bv = v; pred = k; }
      }
      float gv = tp[0];
      int gt = 0;
      g1 = Ct > 1 ? tp[1] : 0.f;
      for (int k = 1; k < Ct; ++k) {
        const float v = tp[k];
        if ((v > gv || v != v) && gv == gv) { gv = v; gt = k; }
      }
      bin = gt != L ? (gt < L ? gt : L - 1) * L + pred : -1;   // gt > L cannot happen with Ct <= L + 1
      if (first) {
        // GetSegMetricsNp: channel-1 logit thresholded at 0 (`>0 -> 1`, `<0 -> 0`, 0 stays 0) against the one-hot channel 1
        const bool pp = l1 > 0.f, gp = g1 != 0.f;
        c[0] += pp && gp; c[1] += pp && !gp; c[2] += !pp && gp; c[3] += !pp && !gp;
        // branch mask: ground-truth branch or a raw logit that is exactly 1.0 (the mask is taken BEFORE thresholding);
        // f1 'micro' over the 1-D selection is the fraction of matching values
        if (g1 == 1.f || l1 == 1.f) {
          const float pv = l1 > 0.f ? 1.f : (l1 < 0.f ? 0.f : l1);
          c[5] += 1; c[4] += pv == g1;
        }
      }
    }
    if (small) {
      // few classes (roses: 2): every lane of a wave hits the same handful of LDS words, so count with wave ballots
      // instead of serialised LDS atomics; lane k keeps bin k
      for (int k = 0; k < L * L; ++k) {
        const unsigned int n = (unsigned int)__popcll(__ballot(bin == k));
        hreg += lane == k ? n : 0u;
      }
    } else if (bin >= 0) {
      atomicAdd(&hist[bin], 1u);
    }
    if (disp && valid) {
      float dp = Elem<T>::ld(disp + p), dg = dtgt[p];
      if (mask_invalid) { const float z = dg > 0.f ? 1.f : 0.f; dp *= z; dg *= z; }   // `zeros = (disp > 0)` of lossDisp_fn
      // unnormalizedErrorNP: |pred*max - gt*max| > 3 where gt > 0
      const float th = dg > 0.f ? 1.f : 0.f;
      const float e = fabsf(dp * max_disp - dg * max_disp) * th;
      c[6] += e > 3.f; c[7] += dg > 0.f;
      if (first) {
        c[9] += 1;                                    // pixels behind the image-0 means (replayed launches count themselves)
        const float d = dg - dp;
        const float sq = d * d;                       // float32 like the numpy expression
        const float rel = sq / dg;                    // GetSqRel: inf / nan where gt == 0, as in the reference
        s[0] += (double)sq; s[1] += (double)rel;
        if (tgt && Ct > 1 && tp[1] == 1.f) { c[8] += 1; s[2] += (double)sq; s[3] += (double)rel; }
      }
    }
  }

  const int wave = threadIdx.x >> 6;
  if (small && lane < L * L && hreg) atomicAdd(&hist[lane], hreg);
#pragma unroll
  for (int i = 0; i < kNC; ++i) { const long v = wave_sum_l(c[i]); if (lane == 0) wc[wave][i] = v; }
#pragma unroll
  for (int i = 0; i < kNS; ++i) { const double v = wave_sum_d(s[i]); if (lane == 0) wsum[wave][i] = v; }
  __syncthreads();   // also orders the histogram updates above before the flush below
  // memory-side atomics on one cache line serialise: the workgroups spread over `nrep` replicas (one 256-byte line
  // each for the f64 sums, rep_stride counters each), summed by the caller
  const int r = blockIdx.x % nrep;
  unsigned long long* cr = counts + (long)r * rep_stride;
  double* sr = sums + (long)r * SDHIP_METRIC_SUM_STRIDE;
  if (threadIdx.x < kNC) {
    const long v = wc[0][threadIdx.x] + wc[1][threadIdx.x] + wc[2][threadIdx.x] + wc[3][threadIdx.x];
    if (v) atomicAdd(&cr[L * L + threadIdx.x], (unsigned long long)v);
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + kNS) {
    const int i = threadIdx.x - 64;
    const double v = wsum[0][i] + wsum[1][i] + wsum[2][i] + wsum[3][i];
    if (v != 0.0) atomicAdd(&sr[i], v);
  }
  for (int i = threadIdx.x; i < L * L; i += 256)
    if (hist[i]) atomicAdd(&cr[i], (unsigned long long)hist[i]);
}

template <typename T>
int launch_metrics(const void* seg, int lds, const float* seg_target, int ldt, int Ct, const void* disp, const float* disp_target,
                   long* counts, double* sums, long npix, long hw, int L, float max_disp, int mask_invalid, int nrep,
                   int rep_stride, hipStream_t stream) {
  long blocks = (npix + 255) / 256;
  if (blocks > 1024) blocks = 1024;       // 4 workgroups per CU
  if (!seg_target) ldt = 0;               // the kernel lays its LDS tile out from these
  if (!seg) lds = 0;
  const bool stage = (seg && lds > 4) || (seg_target && ldt > 4);
  const size_t lds_bytes = stage ? (size_t)256 * ((seg_target ? ldt : 0) * 4 + (seg ? lds : 0) * sizeof(T)) : 0;
  auto k = stage ? step_metrics_kernel<T, true> : step_metrics_kernel<T, false>;
  if (lds_bytes > 150 * 1024) SDHIP_FAIL(SDHIP_ERR_UNSUPPORTED, "step_metrics: pixel strides %d / %d too wide for the LDS tile", lds, ldt);
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "step_metrics: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, (const T*)seg, lds, seg_target, ldt, Ct, (const T*)disp,
                     disp_target, (unsigned long long*)counts, sums, npix, hw, L, max_disp, mask_invalid, nrep, rep_stride);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

}  // namespace

extern "C" int sdhip_step_metrics(const void* seg, int lds, const float* seg_target, int ldt, int Ct, const void* disp,
                                  const float* disp_target, long* counts, double* sums, int nrep, int rep_stride, int B,
                                  long hw, int L, float max_disp, int mask_invalid, int dtype, void* stream) {
  SDHIP_CHECK_ARG(counts && sums && B > 0 && hw > 0, "step_metrics: bad arguments");
  SDHIP_CHECK_ARG(seg || disp, "step_metrics: neither a segmentation nor a disparity output was given");
  SDHIP_CHECK_ARG(L >= 1 && L <= kMaxL, "step_metrics: 1 <= labels <= %d expected, got %d", kMaxL, L);
  SDHIP_CHECK_ARG(nrep >= 1 && rep_stride >= L * L + kNC, "step_metrics: nrep %d / rep_stride %d (need >= %d counters per replica)",
                  nrep, rep_stride, L * L + kNC);
  if (seg) SDHIP_CHECK_ARG(seg_target && lds >= L, "step_metrics: segmentation target missing or pixel stride %d < labels %d", lds, L);
  if (seg_target) SDHIP_CHECK_ARG((Ct == L || Ct == L + 1) && ldt >= Ct,
                                  "step_metrics: target must hold labels or labels+1 (ignore) one-hot channels (L=%d Ct=%d ldt=%d)", L, Ct, ldt);
  if (disp) SDHIP_CHECK_ARG(disp_target, "step_metrics: disparity target missing");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "step_metrics: unknown dtype %d", dtype);
  const long npix = (long)B * hw;
  if (dtype == SDHIP_F32)
    return launch_metrics<float>(seg, lds, seg_target, ldt, Ct, disp, disp_target, counts, sums, npix, hw, L, max_disp, mask_invalid,
                                 nrep, rep_stride, (hipStream_t)stream);
  return launch_metrics<bf16_t>(seg, lds, seg_target, ldt, Ct, disp, disp_target, counts, sums, npix, hw, L, max_disp, mask_invalid,
                                nrep, rep_stride, (hipStream_t)stream);
}
