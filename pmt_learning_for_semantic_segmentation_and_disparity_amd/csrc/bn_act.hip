// BatchNorm (train-mode batch statistics) + activation, decomposed for fusion, for gfx950.
//
// Replaces nn.BatchNorm2d (+ nn.ReLU / skip add) as used by convbn / deconvbn /
// Conv2DownUp (models/dsnet_t2.py:16-117), the DenseNet layers
// (models/densenet.py:25-93,119-128) and ASPP (models/aspp.py:7-32); semantics of
// the statistics as restated in sync_batchnorm/batchnorm.py:114-126 (sum, sum of
// squares, count -> mean, biased variance for normalisation, unbiased for
// running_var, momentum 0.1).
//
//   statistics  S[g] = (sum x, sum x^2) per channel and statistics group g (f64; produced by the
//               conv epilogue or by sdhip_channel_stats; a "group" is a sub-batch that the
//               reference normalises separately, e.g. the left and the right tower pass)
//   finalize    scale = gamma * invstd, shift = beta - mean * scale           (tiny, per channel)
//   apply       y = act(x * scale + shift) (+ residual)                        (HBM bound, 16 B/lane)
//   backward    gx = gy * act' * scale,  dscale = sum gy*act'*x,  dshift = sum gy*act'   (one pass)
//               finalize_bwd: (dscale, dshift) -> dgamma, dbeta, dS
//               stats_fix   : g += dS1 + 2 * x * dS2      (gradient through the statistics)
//
// All elementwise kernels view a tensor as [pixels][C] rows with pixel stride ld.
#include "sdhip_common.h"

namespace {

struct RowGeom {  // how a 256-thread block walks a [npix][C] matrix
  int tx, ty;     // threads across channel units / across pixels
  int units;      // channel units per row (16-byte chunks, or single elements in scalar mode)
  unsigned magic; // div_magic(tx): thread -> (tx, ty) without an integer division per lane (these kernels are often ~5 us)
};

inline RowGeom row_geom(int units) {
  RowGeom r;
  r.units = units;
  r.tx = units < 64 ? units : 64;
  r.ty = 256 / r.tx;
  r.magic = r.tx > 1 ? (unsigned)(0x100000000ULL / (unsigned)r.tx) + 1u : 0u;
  return r;
}

// threadIdx.x -> ty = tid / tx (tid < 256: exact with the 2^32/tx + 1 magic), tx = tid - ty * tx
#define ROW_TXTY(rg, tx, ty) const int ty = (rg).tx > 1 ? (int)__umulhi((unsigned)threadIdx.x, (rg).magic) : (int)threadIdx.x; \
                             const int tx = (int)threadIdx.x - ty * (rg).tx

// y = act(x*scale + shift) (+ res);  grid: (pixel slabs, unit groups, stat groups)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                         const T* __restrict__ res, int ldr,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int C, long npix_g, int act, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int g = blockIdx.z, c0 = u * N;
  float sc[N], sf[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { sc[e] = scale ? scale[g * C + c0 + e] : 1.f; sf[e] = shift ? shift[g * C + c0 + e] : 0.f; }
  const long base = (long)g * npix_g;
  for (long pix = (long)blockIdx.x * rg.ty + ty; pix < npix_g; pix += (long)gridDim.x * rg.ty) {
    float f[N], r[N];
    Unit<T, VEC>::load(x + (base + pix) * ldx + c0, f);
    if (res) Unit<T, VEC>::load(res + (base + pix) * ldr + c0, r);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float v = fmaf(f[e], sc[e], sf[e]);
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = 1.f / (1.f + __expf(-v));
      if (res) v += r[e];
      f[e] = v;
    }
    Unit<T, VEC>::store(y + (base + pix) * ldy + c0, f);
  }
}

// gx = gy * act'(x*scale+shift) * scale;  dscale += sum gy*act'*x;  dshift += sum gy*act'
// mode 0: both; mode 1: only the reductions (gx not written)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const T* gy, int ldg, const T* __restrict__ x, int ldx,
                                                             T* gx, int ldgx,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ dscale, float* __restrict__ dshift, int nrep,
                                                             int C, long npix_g, int act, int accumulate, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  __shared__ float red[2][256 * (VEC ? Chunk<T>::N : 1)];
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  const bool live = ty < rg.ty && u < rg.units;
  const int g = blockIdx.z, c0 = u * N;
  float sc[N], sf[N], a1[N], a2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = (live && scale) ? scale[g * C + c0 + e] : 1.f;
    sf[e] = (live && shift) ? shift[g * C + c0 + e] : 0.f;
    a1[e] = 0.f; a2[e] = 0.f;
  }
  const long base = (long)g * npix_g;
  if (live) {
    // 4 pixels per trip, all 8 loads issued before the first use: with the grid kept small for the sake of the
    // atomics below (see there) each thread has to keep more bytes in flight itself
    const long stride = (long)gridDim.x * rg.ty;
    for (long pix0 = (long)blockIdx.x * rg.ty + ty; pix0 < npix_g; pix0 += 4 * stride) {
      float gv[4][N], xv[4][N];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = pix0 + k * stride;
        if (pix < npix_g) {
          Unit<T, VEC>::load(gy + (base + pix) * ldg + c0, gv[k]);
          Unit<T, VEC>::load(x + (base + pix) * ldx + c0, xv[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = pix0 + k * stride;
        if (pix < npix_g) {
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const float z = fmaf(xv[k][e], sc[e], sf[e]);
            float gm = gv[k][e];
            if (act == 1) gm = z > 0.f ? gm : 0.f;
            else if (act == 2) { const float sg = 1.f / (1.f + __expf(-z)); gm *= sg * (1.f - sg); }
            else if (act == 4) gm *= z * (1.f - z);  // x holds the sigmoid OUTPUT
            a1[e] = fmaf(gm, xv[k][e], a1[e]);
            a2[e] += gm;
            gv[k][e] = gm * sc[e];
          }
          if (gx) {
            if (accumulate) {
              float old[N];
              Unit<T, VEC>::load(gx + (base + pix) * ldgx + c0, old);
#pragma unroll
              for (int e = 0; e < N; ++e) gv[k][e] += old[e];
            }
            Unit<T, VEC>::store(gx + (base + pix) * ldgx + c0, gv[k]);
          }
        }
      }
    }
  }
  if (!dscale) return;  // uniform
  // Reduce over the ty rows of the block in LDS, then ONE atomic wave-instruction per sum with consecutive lanes on
  // consecutive channels.  Float atomics execute at the memory side, one 256-byte request at a time per line: thousands
  // of workgroups x 16 scattered 8-lane instructions into the same replica line serialised into most of this kernel's
  // run time (69 us on a 400 MB pass that streams in 25).
#pragma unroll
  for (int e = 0; e < N; ++e) { red[0][threadIdx.x * N + e] = a1[e]; red[1][threadIdx.x * N + e] = a2[e]; }
  __syncthreads();
  const int nch = min(rg.tx, rg.units - blockIdx.y * rg.tx) * N;   // channels of this block's unit group
  for (int cc = threadIdx.x; cc < 2 * nch; cc += 256) {
    const int which = cc >= nch, ch = cc - which * nch;
    const int txx = ch / N, e = ch - txx * N;
    float sum = 0.f;
    for (int r = 0; r < rg.ty; ++r) sum += red[which][(r * rg.tx + txx) * N + e];
    const long rep = (long)(blockIdx.x % nrep) * gridDim.z * C;   // replica r of [nrep][G][C]
    atomicAdd((which ? dshift : dscale) + rep + g * C + blockIdx.y * rg.tx * N + ch, sum);
  }
}

// g_out = g_in + dS1 + 2 * x * dS2   (dS is [G][2][C], f64 like the statistics it is the gradient of)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void stats_fix_kernel(const T* gin, int ldgi, const T* __restrict__ x, int ldx,
                                                        T* gout, int ldgo, const double* __restrict__ dS, int ldc,
                                                        int C, long npix_g, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int g = blockIdx.z, c0 = u * N;
  float a[N], b2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { a[e] = (float)dS[((long)g * 2 + 0) * ldc + c0 + e]; b2[e] = (float)(2.0 * dS[((long)g * 2 + 1) * ldc + c0 + e]); }
  const long base = (long)g * npix_g;
  for (long pix = (long)blockIdx.x * rg.ty + ty; pix < npix_g; pix += (long)gridDim.x * rg.ty) {
    float gv[N], xv[N];
    Unit<T, VEC>::load(gin + (base + pix) * ldgi + c0, gv);
    Unit<T, VEC>::load(x + (base + pix) * ldx + c0, xv);
#pragma unroll
    for (int e = 0; e < N; ++e) gv[e] = gv[e] + fmaf(xv[e], b2[e], a[e]);
    Unit<T, VEC>::store(gout + (base + pix) * ldgo + c0, gv);
  }
}

// Second phase of the two-phase BatchNorm backward: gx = gy * act'(x*scale+shift) * scale + dS1 + 2 * x * dS2 in ONE pass
// (the first phase is affine_act_bwd with gx == NULL: reductions only).  10 bytes per element instead of the 12 of
// "write gx, then read it back for the statistics path".
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ gy, int ldg, const T* __restrict__ x, int ldx,
                                                           T* __restrict__ gx, int ldgx, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const double* __restrict__ dS, int ldc,
                                                           int C, long npix_g, int act, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int g = blockIdx.z, c0 = u * N;
  float sc[N], sf[N], a[N], b2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = scale[g * C + c0 + e]; sf[e] = shift[g * C + c0 + e];
    a[e] = (float)dS[((long)g * 2 + 0) * ldc + c0 + e]; b2[e] = (float)(2.0 * dS[((long)g * 2 + 1) * ldc + c0 + e]);
  }
  const long base = (long)g * npix_g;
  const long stride = (long)gridDim.x * rg.ty;
  for (long pix0 = (long)blockIdx.x * rg.ty + ty; pix0 < npix_g; pix0 += 4 * stride) {
    float gv[4][N], xv[4][N];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long pix = pix0 + k * stride;
      if (pix < npix_g) {
        Unit<T, VEC>::load(gy + (base + pix) * ldg + c0, gv[k]);
        Unit<T, VEC>::load(x + (base + pix) * ldx + c0, xv[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long pix = pix0 + k * stride;
      if (pix < npix_g) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float z = fmaf(xv[k][e], sc[e], sf[e]);
          float gm = gv[k][e];
          if (act == 1) gm = z > 0.f ? gm : 0.f;
          else if (act == 2) { const float sg = 1.f / (1.f + __expf(-z)); gm *= sg * (1.f - sg); }
          // the first-phase result the one-pass form would have stored is rounded to T before the statistics path is added
          gv[k][e] = Elem<T>::rnd(gm * sc[e]) + fmaf(xv[k][e], b2[e], a[e]);
        }
        Unit<T, VEC>::store(gx + (base + pix) * ldgx + c0, gv[k]);
      }
    }
  }
}

// ---- consumer-side finalize: the per-channel kernels between a reduction and the pass that uses its result are pure
// launch latency (~5 us + a graph node each, ~400 per training step).  These variants let every workgroup of the
// elementwise pass derive the coefficients of ITS channel slice from the replica sums (a few KB from L2), and let the
// first workgroup of each slice publish them / update the parameters' side outputs.

// y = act(BatchNorm_train(x)) (+ res) straight from the statistics S (f64 [nrep][G][2][ldc]) of x.
// Also writes scale/shift/mean/invstd ([G][C], for the backward pass) and updates the running statistics (groups in order).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void affine_act_bn_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                            const T* __restrict__ res, int ldr,
                                                            const double* __restrict__ S, int ldc, int nrep,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ rmean, float* __restrict__ rvar,
                                                            float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                            float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                            int C, long npix_g, double count, float eps, float momentum, int act, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  __shared__ float s_sc[64 * 8], s_sf[64 * 8];
  __shared__ double s_red[2][256];
  const int G = gridDim.z, g = blockIdx.z;
  const int ch0 = blockIdx.y * rg.tx * N;
  const int nch = min(rg.tx, rg.units - (int)blockIdx.y * rg.tx) * N;
  // Replica fold of the workgroup's channels in ONE round trip: the 256 threads are (channel slot cl, replica lane rl) with as
  // many replica lanes as fit (64 channels: 4 lanes of 8 replicas each), one barrier pair per statistics group.  (The first
  // form walked the channels 32 at a time — four dependent load / barrier rounds for the 128-channel DenseNet bottlenecks,
  // most of this kernel's time on the small maps.)
  const int ncp = nch <= 32 ? 32 : nch <= 64 ? 64 : nch <= 128 ? 128 : 256;
  const int P = 256 / ncp;
  const int cl = threadIdx.x & (ncp - 1), rl = threadIdx.x / ncp;
  const bool writer = blockIdx.x == 0 && g == 0 && rmean;   // wave-uniform per workgroup: running statistics of the slice
  for (int cb = 0; cb < nch; cb += ncp) {
    const int c = ch0 + cb + cl;
    const bool okc = cb + cl < nch && c < C;
    float rm = 0.f, rv = 0.f;
    if (writer && rl == 0 && okc) { rm = rmean[c]; rv = rvar[c]; }
    for (int gg = 0; gg < G; ++gg) {
      if (gg != g && !writer) continue;                      // everybody needs its own group; the writer all, in order
      double a1 = 0., a2 = 0.;
      if (okc) {
        for (int r = rl; r < nrep; r += P) {
          const double* Sr = S + (long)r * G * 2 * ldc;
          a1 += Sr[((long)gg * 2 + 0) * ldc + c];
          a2 += Sr[((long)gg * 2 + 1) * ldc + c];
        }
      }
      __syncthreads();
      s_red[0][threadIdx.x] = a1; s_red[1][threadIdx.x] = a2;
      __syncthreads();
      if (rl == 0 && okc) {
        double s1 = 0., s2 = 0.;
        for (int r = 0; r < P; ++r) { s1 += s_red[0][r * ncp + cl]; s2 += s_red[1][r * ncp + cl]; }
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        if (var < 0.) var = 0.;
        if (gg == g) {
          const float inv = (float)(1.0 / sqrt(var + (double)eps));
          const float sc = (gamma ? gamma[c] : 1.f) * inv;
          const float sf = (float)((double)(beta ? beta[c] : 0.f) - mu * (double)sc);
          s_sc[cb + cl] = sc; s_sf[cb + cl] = sf;
          if (blockIdx.x == 0) {
            scale_out[g * C + c] = sc; shift_out[g * C + c] = sf; mean_out[g * C + c] = (float)mu; invstd_out[g * C + c] = inv;
          }
        }
        if (writer) {
          const double unb = count > 1. ? var * count / (count - 1.) : var;
          rm = (1.f - momentum) * rm + momentum * (float)mu;
          rv = (1.f - momentum) * rv + momentum * (float)unb;
        }
      }
    }
    if (writer && rl == 0 && okc) { rmean[c] = rm; rvar[c] = rv; }
  }
  __syncthreads();
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int c0 = u * N;
  float sc[N], sf[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { sc[e] = s_sc[tx * N + e]; sf[e] = s_sf[tx * N + e]; }
  const long base = (long)g * npix_g;
  for (long pix = (long)blockIdx.x * rg.ty + ty; pix < npix_g; pix += (long)gridDim.x * rg.ty) {
    float f[N], r[N];
    Unit<T, VEC>::load(x + (base + pix) * ldx + c0, f);
    if (res) Unit<T, VEC>::load(res + (base + pix) * ldr + c0, r);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float v = fmaf(f[e], sc[e], sf[e]);
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = 1.f / (1.f + __expf(-v));
      if (res) v += r[e];
      f[e] = v;
    }
    Unit<T, VEC>::store(y + (base + pix) * ldy + c0, f);
  }
}

// gx = gy * act'(x*scale+shift) * scale + dS1 + 2*x*dS2 with dS derived in the kernel from the (dscale, dshift) replica
// sums of the first phase; the first workgroup of a channel slice also writes dgamma / dbeta (summed over groups).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_fin_kernel(const T* gy, int ldg, const T* __restrict__ x, int ldx,
                                                               T* gx, int ldgx, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ dscale,
                                                               const float* __restrict__ dshift, int nrep,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int acc_par, float pscale, int C, long npix_g,
                                                               double inv_count, int act, const double* __restrict__ dsum, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  __shared__ float s_a[64 * 8], s_b2[64 * 8];
  __shared__ float s_red[2][256];
  const int G = gridDim.z, g = blockIdx.z;
  const int ch0 = blockIdx.y * rg.tx * N;
  const int nch = min(rg.tx, rg.units - (int)blockIdx.y * rg.tx) * N;
  // replica fold in one round trip (see affine_act_bn_kernel): threads = (channel slot cl, replica lane rl)
  const int ncp = nch <= 32 ? 32 : nch <= 64 ? 64 : nch <= 128 ? 128 : 256;
  const int P = 256 / ncp;
  const int cl = threadIdx.x & (ncp - 1), rl = threadIdx.x / ncp;
  const bool writer = blockIdx.x == 0 && g == 0 && dgamma;     // wave-uniform per workgroup
  for (int cb = 0; cb < nch; cb += ncp) {
    const int c = ch0 + cb + cl;
    const bool okc = cb + cl < nch && c < C;
    float dg = 0.f, db = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      if (gg != g && !writer) continue;                         // everybody needs its own group; the writer all of them
      float ds = 0.f, dh = 0.f;
      if (okc) {
        if (dsum) {   // f64 [nrep][G][2][C] as written by the epilogue of sdhip_conv2d_fwd_bnbwd
          double d0 = 0., d1 = 0.;
          for (int r = rl; r < nrep; r += P) { d0 += dsum[(((long)r * G + gg) * 2 + 0) * C + c]; d1 += dsum[(((long)r * G + gg) * 2 + 1) * C + c]; }
          ds = (float)d0; dh = (float)d1;
        } else {
          for (int r = rl; r < nrep; r += P) { ds += dscale[((long)r * G + gg) * C + c]; dh += dshift[((long)r * G + gg) * C + c]; }
        }
      }
      __syncthreads();
      s_red[0][threadIdx.x] = ds; s_red[1][threadIdx.x] = dh;
      __syncthreads();
      if (rl == 0 && okc) {
        ds = 0.f; dh = 0.f;
        for (int r = 0; r < P; ++r) { ds += s_red[0][r * ncp + cl]; dh += s_red[1][r * ncp + cl]; }
        const float gm = gamma ? gamma[c] : 1.f;
        const float mu = mean[gg * C + c], inv = invstd[gg * C + c];
        const float t = ds - mu * dh;
        dg += inv * t; db += dh;
        if (gg == g) {
          const double dinv = (double)gm * t;
          const double dvar = -0.5 * dinv * (double)inv * inv * inv;
          const double dmu = -(double)gm * inv * dh - 2.0 * mu * dvar;
          s_a[cb + cl] = (float)(dmu * inv_count);
          s_b2[cb + cl] = (float)(2.0 * dvar * inv_count);
        }
      }
    }
    if (writer && rl == 0 && okc) {      // pscale = 1 / world size when the sums are global (the gradient all-reduce adds the ranks' shares)
      dg *= pscale; db *= pscale;
      dgamma[c] = acc_par ? dgamma[c] + dg : dg;
      dbeta[c] = acc_par ? dbeta[c] + db : db;
    }
  }
  __syncthreads();
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int c0 = u * N;
  float sc[N], sf[N], a[N], b2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = scale[g * C + c0 + e]; sf[e] = shift[g * C + c0 + e];
    a[e] = s_a[tx * N + e]; b2[e] = s_b2[tx * N + e];
  }
  const long base = (long)g * npix_g;
  const long stride = (long)gridDim.x * rg.ty;
  for (long pix0 = (long)blockIdx.x * rg.ty + ty; pix0 < npix_g; pix0 += 4 * stride) {
    float gv[4][N], xv[4][N];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long pix = pix0 + k * stride;
      if (pix < npix_g) {
        Unit<T, VEC>::load(gy + (base + pix) * ldg + c0, gv[k]);
        Unit<T, VEC>::load(x + (base + pix) * ldx + c0, xv[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long pix = pix0 + k * stride;
      if (pix < npix_g) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float z = fmaf(xv[k][e], sc[e], sf[e]);
          float gm = gv[k][e];
          if (act == 1) gm = z > 0.f ? gm : 0.f;
          else if (act == 2) { const float sg = 1.f / (1.f + __expf(-z)); gm *= sg * (1.f - sg); }
          gv[k][e] = Elem<T>::rnd(gm * sc[e]) + fmaf(xv[k][e], b2[e], a[e]);
        }
        Unit<T, VEC>::store(gx + (base + pix) * ldgx + c0, gv[k]);
      }
    }
  }
}

// S[g][0][c] += sum x, S[g][1][c] += sum x^2  (f64 atomics)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, int ldx, double* __restrict__ S, int ldc, int nrep,
                                                            int C, long npix_g, RowGeom rg) {
  constexpr int N = Unit<T, VEC>::N;
  __shared__ float red[2][256 * (VEC ? Chunk<T>::N : 1)];
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  const bool live = ty < rg.ty && u < rg.units;
  const int g = blockIdx.z, c0 = u * N;
  float a1[N], a2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  const long base = (long)g * npix_g;
  if (live) {
    for (long pix = (long)blockIdx.x * rg.ty + ty; pix < npix_g; pix += (long)gridDim.x * rg.ty) {
      float xv[N];
      Unit<T, VEC>::load(x + (base + pix) * ldx + c0, xv);
#pragma unroll
      for (int e = 0; e < N; ++e) { a1[e] += xv[e]; a2[e] = fmaf(xv[e], xv[e], a2[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) { red[0][threadIdx.x * N + e] = a1[e]; red[1][threadIdx.x * N + e] = a2[e]; }
  __syncthreads();
  const int nch = min(rg.tx, rg.units - blockIdx.y * rg.tx) * N;   // channels of this block's unit group
  for (int cc = threadIdx.x; cc < 2 * nch; cc += 256) {             // consecutive lanes -> consecutive channels (see affine_act_bwd)
    const int which = cc >= nch, ch = cc - which * nch;
    const int txx = ch / N, e = ch - txx * N;
    double sum = 0.;
    for (int r = 0; r < rg.ty; ++r) sum += red[which][(r * rg.tx + txx) * N + e];
    double* Sr = S + (long)(blockIdx.x % nrep) * gridDim.z * 2 * ldc;   // replica r of [nrep][G][2][ldc]
    atomicAdd(Sr + ((long)g * 2 + which) * ldc + blockIdx.y * rg.tx * N + ch, sum);
  }
}

// ---- per-channel finalize (tiny) ------------------------------------------------
// train: S -> mean, invstd, scale, shift, running stats (groups applied in order)
__global__ void bn_finalize_kernel(const double* __restrict__ S, int ldc, int nrep, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, int C, int G, double count, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  if (S == nullptr) {  // eval mode: running statistics
    const float inv = 1.f / sqrtf(rvar[c] + eps);
    for (int g = 0; g < G; ++g) {
      scale[g * C + c] = gm * inv;
      shift[g * C + c] = bt - rmean[c] * gm * inv;
      if (mean_out) { mean_out[g * C + c] = rmean[c]; invstd_out[g * C + c] = inv; }
    }
    return;
  }
  float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    double s1 = 0., s2 = 0.;
#pragma unroll 8
    for (int r = 0; r < nrep; ++r) {
      const double* Sr = S + (long)r * G * 2 * ldc;
      s1 += Sr[((long)g * 2 + 0) * ldc + c];
      s2 += Sr[((long)g * 2 + 1) * ldc + c];
    }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.) var = 0.;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gm * inv;
    scale[g * C + c] = sc;
    shift[g * C + c] = (float)((double)bt - mu * (double)sc);
    mean_out[g * C + c] = (float)mu;
    invstd_out[g * C + c] = inv;
    const double unb = count > 1. ? var * count / (count - 1.) : var;
    rm = (1.f - momentum) * rm + momentum * (float)mu;
    rv = (1.f - momentum) * rv + momentum * (float)unb;
  }
  if (rmean) { rmean[c] = rm; rvar[c] = rv; }
}

// (dscale, dshift)[rep][G][C] -> dgamma[C], dbeta[C] (summed over groups), dS[G][2][C]
// One workgroup = 32 channels x 8 replica lanes: every thread sums nrep/8 replicas (independent loads, one round trip),
// the 8 partials meet in LDS.  (A single thread per channel walking 32 replicas made this per-layer kernel a 17 us
// latency chain on the backward critical path.)
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ dscale, const float* __restrict__ dshift, int nrep,
                                       const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, double* __restrict__ dS, int ldc, int acc_flags,
                                       int C, int G, double inv_count, int train) {
  __shared__ float red[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool okc = c < C;
  const int acc_ds = acc_flags & 1, acc_par = acc_flags & 2;   // bit 0: dstats += ; bit 1: dgamma/dbeta +=
  float dg = 0.f, db = 0.f;
  for (int g = 0; g < G; ++g) {
    float ds = 0.f, dh = 0.f;
    if (okc) {
#pragma unroll 4
      for (int r = rl; r < nrep; r += 8) { ds += dscale[((long)r * G + g) * C + c]; dh += dshift[((long)r * G + g) * C + c]; }
    }
    __syncthreads();   // previous group's partials are consumed
    red[0][rl][cl] = ds; red[1][rl][cl] = dh;
    __syncthreads();
    if (rl == 0 && okc) {
      ds = 0.f; dh = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) { ds += red[0][r][cl]; dh += red[1][r][cl]; }
      const float gm = gamma ? gamma[c] : 1.f;
      const float mu = mean[g * C + c], inv = invstd[g * C + c];
      const float t = ds - mu * dh;       // d/d(gamma*invstd) collected
      dg += inv * t;
      db += dh;
      if (dS) {
        if (train) {
          const double dinv = (double)gm * t;
          const double dvar = -0.5 * dinv * (double)inv * inv * inv;
          const double dmu = -(double)gm * inv * dh - 2.0 * mu * dvar;
          double* d0 = dS + ((long)g * 2 + 0) * ldc + c;
          double* d1 = dS + ((long)g * 2 + 1) * ldc + c;
          *d0 = (acc_ds ? *d0 : 0.) + dmu * inv_count;
          *d1 = (acc_ds ? *d1 : 0.) + dvar * inv_count;
        } else if (!acc_ds) {
          dS[((long)g * 2 + 0) * ldc + c] = 0.;
          dS[((long)g * 2 + 1) * ldc + c] = 0.;
        }
      }
    }
  }
  if (rl == 0 && okc) {
    if (dgamma) dgamma[c] = acc_par ? dgamma[c] + dg : dg;
    if (dbeta) dbeta[c] = acc_par ? dbeta[c] + db : db;
  }
}

template <typename T>
bool vec_rows(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  const int n = Chunk<T>::N;
  if (C % n) return false;
  for (int l : lds) if (l % n) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

struct Plan { RowGeom rg; dim3 grid; };
Plan plan(int units, long npix_g, int G, int max_blocks = 2048) {
  Plan p;
  p.rg = row_geom(units);
  const int gy = sdhip_cdiv(units, p.rg.tx);
  long gx = (npix_g + p.rg.ty - 1) / p.rg.ty;
  const long cap = max_blocks / ((long)gy * G) > 0 ? max_blocks / ((long)gy * G) : 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  p.grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)G);
  return p;
}

// workgroups of the consumer-side-finalize kernels: each one re-derives its coefficients from the replica sums (KBs from L2)
int tune_fused_blocks() { return sdhip_diag().tune_fused_blocks; }

int check_rows(const char* who, long npix, int C, int G, int dtype) {
  SDHIP_CHECK_ARG(npix > 0 && C > 0 && G >= 1 && npix % G == 0, "%s: bad shape npix=%ld C=%d groups=%d", who, npix, C, G);
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "%s: unknown dtype %d", who, dtype);
  return 0;
}

}  // namespace

// out[row*ldo + c] += sum_r ws[r*rep_stride + row*ldw + c]   (rows = 2*groups statistics rows)
__global__ void replica_sum_kernel(const double* __restrict__ ws, double* __restrict__ out, int nrep, long rep_stride,
                                   int rows, int C, int ldw, int ldo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const int row = i / C, c = i - row * C;
  double s = 0.;
#pragma unroll 8
  for (int r = 0; r < nrep; ++r) s += ws[(long)r * rep_stride + (long)row * ldw + c];
  out[(long)row * ldo + c] += s;
}

// fold + finalize in one launch (DenseNet: the statistics of a layer's 32 new slab channels are folded into the slab
// statistics and the NEXT layer's norm1 — which reads all channels up to and including the new ones — is finalized):
// channel c of [c_new0, c_new0 + Cn) first adds its replica sums into S, then every channel < C is finalized from S.
__global__ void bn_fold_finalize_kernel(const double* __restrict__ ws, int nrep, int ldw, int c_new0, int Cn, double* __restrict__ S, int ldc,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                        float* __restrict__ rvar, float* __restrict__ scale, float* __restrict__ shift,
                                        float* __restrict__ mean_out, float* __restrict__ invstd_out, int C, int G, double count, float eps,
                                        float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float gm = gamma[c], bt = beta[c];
  float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
  const bool fresh = c >= c_new0 && c < c_new0 + Cn;
  for (int g = 0; g < G; ++g) {
    double s1 = S[((long)g * 2 + 0) * ldc + c], s2 = S[((long)g * 2 + 1) * ldc + c];
    if (fresh) {
      const int cn = c - c_new0;
#pragma unroll 8
      for (int r = 0; r < nrep; ++r) {
        const double* Sr = ws + (long)r * G * 2 * ldw;
        s1 += Sr[((long)g * 2 + 0) * ldw + cn];
        s2 += Sr[((long)g * 2 + 1) * ldw + cn];
      }
      S[((long)g * 2 + 0) * ldc + c] = s1;
      S[((long)g * 2 + 1) * ldc + c] = s2;
    }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.) var = 0.;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gm * inv;
    scale[g * C + c] = sc;
    shift[g * C + c] = (float)((double)bt - mu * (double)sc);
    mean_out[g * C + c] = (float)mu;
    invstd_out[g * C + c] = inv;
    const double unb = count > 1. ? var * count / (count - 1.) : var;
    rm = (1.f - momentum) * rm + momentum * (float)mu;
    rv = (1.f - momentum) * rv + momentum * (float)unb;
  }
  if (rmean) { rmean[c] = rm; rvar[c] = rv; }
}

extern "C" int sdhip_bn_fold_finalize(const double* ws, int nrep, int ldw, int c_new0, int Cn, double* S, int ldc,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      float* scale, float* shift, float* mean_out, float* invstd_out, int C, int groups,
                                      double count, float eps, float momentum, void* stream) {
  SDHIP_CHECK_ARG(ws && S && gamma && beta && scale && shift && mean_out && invstd_out && nrep >= 1 && Cn > 0 && c_new0 >= 0 &&
                  c_new0 + Cn <= C && ldw >= Cn && ldc >= C && groups >= 1 && count > 0., "bn_fold_finalize: bad arguments");
  SDHIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_fold_finalize: running statistics must come together");
  hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3(sdhip_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, ws, nrep, ldw, c_new0, Cn, S, ldc,
                     gamma, beta, running_mean, running_var, scale, shift, mean_out, invstd_out, C, groups, count, eps, momentum);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_stats_replica_sum(const double* ws, double* out, int nrep, int groups, int C, int ldw, int ldo, void* stream) {
  SDHIP_CHECK_ARG(ws && out && nrep >= 1 && groups >= 1 && C > 0 && ldw >= C && ldo >= C, "stats_replica_sum: bad arguments");
  const int rows = 2 * groups;
  hipLaunchKernelGGL(replica_sum_kernel, dim3(sdhip_cdiv((long)rows * C, 256)), dim3(256), 0, (hipStream_t)stream, ws, out, nrep,
                     (long)rows * ldw, rows, C, ldw, ldo);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_affine_act(const void* x, int ldx, void* y, int ldy, const void* res, int ldr,
                                const float* scale, const float* shift, long npix, int C, int groups, int act,
                                int dtype, void* stream) {
  const int G = groups;
  if (int rc = check_rows("affine_act", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(x && y && ldx >= C && ldy >= C && (!res || ldr >= C), "affine_act: bad pointers/strides");
  hipStream_t s = (hipStream_t)stream;
#define ARGS(T) (const T*)x, ldx, (T*)y, ldy, (const T*)res, ldr, scale, shift, C, npix / G, act
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldx, ldy, res ? ldr : 0}, {x, y, res});
    Plan pl = plan(v ? C / 4 : C, npix / G, G);
    if (v) hipLaunchKernelGGL((affine_act_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((affine_act_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldx, ldy, res ? ldr : 0}, {x, y, res});
    Plan pl = plan(v ? C / 8 : C, npix / G, G);
    if (v) hipLaunchKernelGGL((affine_act_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((affine_act_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_affine_act_bwd(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                                    const float* scale, const float* shift, float* dscale, float* dshift, int nrep,
                                    long npix, int C, int groups, int act, int accumulate, int prezeroed, int dtype, void* stream) {
  if (nrep < 1) nrep = 1;
  const int G = groups;
  if (int rc = check_rows("affine_act_bwd", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && x && ldg >= C && ldx >= C && (!gx || ldgx >= C), "affine_act_bwd: bad pointers/strides");
  SDHIP_CHECK_ARG((dscale == nullptr) == (dshift == nullptr), "affine_act_bwd: dscale/dshift must come together");
  hipStream_t s = (hipStream_t)stream;
  if (dscale && !prezeroed) {
    if (sdhip_zero_async(dscale, sizeof(float) * (size_t)nrep * G * C, s) != hipSuccess ||
        sdhip_zero_async(dshift, sizeof(float) * (size_t)nrep * G * C, s) != hipSuccess)
      SDHIP_FAIL(SDHIP_ERR_LAUNCH, "affine_act_bwd: memset failed");
  }
#define ARGS(T) (const T*)gy, ldg, (const T*)x, ldx, (T*)gx, ldgx, scale, shift, dscale, dshift, nrep, C, npix / G, act, accumulate
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldg, ldx, gx ? ldgx : 0}, {gy, x, gx});
    Plan pl = plan(v ? C / 4 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((affine_act_bwd_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((affine_act_bwd_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldg, ldx, gx ? ldgx : 0}, {gy, x, gx});
    Plan pl = plan(v ? C / 8 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((affine_act_bwd_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((affine_act_bwd_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_stats_fix(const void* gin, int ldgi, const void* x, int ldx, void* gout, int ldgo,
                               const double* dS, int ldc, long npix, int C, int groups, int dtype, void* stream) {
  if (ldc <= 0) ldc = C;
  const int G = groups;
  if (int rc = check_rows("stats_fix", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(gin && x && gout && dS && ldgi >= C && ldx >= C && ldgo >= C, "stats_fix: bad pointers/strides");
  hipStream_t s = (hipStream_t)stream;
#define ARGS(T) (const T*)gin, ldgi, (const T*)x, ldx, (T*)gout, ldgo, dS, ldc, C, npix / G
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldgi, ldx, ldgo}, {gin, x, gout});
    Plan pl = plan(v ? C / 4 : C, npix / G, G);
    if (v) hipLaunchKernelGGL((stats_fix_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((stats_fix_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldgi, ldx, ldgo}, {gin, x, gout});
    Plan pl = plan(v ? C / 8 : C, npix / G, G);
    if (v) hipLaunchKernelGGL((stats_fix_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((stats_fix_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_bn_bwd_apply(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                                  const float* scale, const float* shift, const double* dS, int ldc,
                                  long npix, int C, int groups, int act, int dtype, void* stream) {
  if (ldc <= 0) ldc = C;
  const int G = groups;
  if (int rc = check_rows("bn_bwd_apply", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && x && gx && scale && shift && dS && ldg >= C && ldx >= C && ldgx >= C, "bn_bwd_apply: bad pointers/strides");
  SDHIP_CHECK_ARG(act == 0 || act == 1 || act == 2, "bn_bwd_apply: activation %d", act);
  hipStream_t s = (hipStream_t)stream;
#define ARGS(T) (const T*)gy, ldg, (const T*)x, ldx, (T*)gx, ldgx, scale, shift, dS, ldc, C, npix / G, act
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldg, ldx, ldgx}, {gy, x, gx});
    Plan pl = plan(v ? C / 4 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldg, ldx, ldgx}, {gy, x, gx});
    Plan pl = plan(v ? C / 8 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_affine_act_bn(const void* x, int ldx, void* y, int ldy, const void* res, int ldr,
                                   const double* stats, int ldc, int nrep, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float* scale, float* shift, float* mean_out,
                                   float* invstd_out, long npix, int C, int groups, double count, float eps, float momentum,
                                   int act, int dtype, void* stream) {
  if (ldc <= 0) ldc = C;
  if (nrep < 1) nrep = 1;
  const int G = groups;
  if (int rc = check_rows("affine_act_bn", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(x && y && stats && scale && shift && mean_out && invstd_out && ldx >= C && ldy >= C && (!res || ldr >= C),
                  "affine_act_bn: bad pointers/strides");
  SDHIP_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && count > 0., "affine_act_bn: bad statistics arguments");
  hipStream_t s = (hipStream_t)stream;
#define ARGS(T) (const T*)x, ldx, (T*)y, ldy, (const T*)res, ldr, stats, ldc, nrep, gamma, beta, running_mean, running_var, \
                scale, shift, mean_out, invstd_out, C, npix / G, count, eps, momentum, act
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldx, ldy, res ? ldr : 0}, {x, y, res});
    Plan pl = plan(v ? C / 4 : C, npix / G, G, tune_fused_blocks());
    if (v) hipLaunchKernelGGL((affine_act_bn_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((affine_act_bn_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldx, ldy, res ? ldr : 0}, {x, y, res});
    Plan pl = plan(v ? C / 8 : C, npix / G, G, tune_fused_blocks());
    if (v) hipLaunchKernelGGL((affine_act_bn_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((affine_act_bn_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

static int bn_bwd_apply_fin_impl(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                                 const float* scale, const float* shift, const float* dscale, const float* dshift, const double* dsum, int nrep,
                                      const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                      int accumulate_params, float param_scale, long npix, int C, int groups, double count, int act, int dtype,
                                      void* stream) {
  if (nrep < 1) nrep = 1;
  const int G = groups;
  if (int rc = check_rows("bn_bwd_apply_fin", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(gy && x && gx && scale && shift && ((dscale && dshift) || dsum) && mean && invstd && ldg >= C && ldx >= C && ldgx >= C,
                  "bn_bwd_apply_fin: bad pointers/strides");
  SDHIP_CHECK_ARG((act == 0 || act == 1 || act == 2) && count > 0. && (dgamma == nullptr) == (dbeta == nullptr),
                  "bn_bwd_apply_fin: bad arguments");
  hipStream_t s = (hipStream_t)stream;
#define ARGS(T) (const T*)gy, ldg, (const T*)x, ldx, (T*)gx, ldgx, scale, shift, dscale, dshift, nrep, gamma, mean, invstd, \
                dgamma, dbeta, accumulate_params, param_scale, C, npix / G, 1.0 / count, act, dsum
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldg, ldx, ldgx}, {gy, x, gx});
    Plan pl = plan(v ? C / 4 : C, npix / G, G, tune_fused_blocks());
    if (v) hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldg, ldx, ldgx}, {gy, x, gx});
    Plan pl = plan(v ? C / 8 : C, npix / G, G, tune_fused_blocks());
    if (v) hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_bn_bwd_apply_fin(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                                      const float* scale, const float* shift, const float* dscale, const float* dshift, int nrep,
                                      const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                      int accumulate_params, float param_scale, long npix, int C, int groups, double count, int act, int dtype,
                                      void* stream) {
  return bn_bwd_apply_fin_impl(gy, ldg, x, ldx, gx, ldgx, scale, shift, dscale, dshift, nullptr, nrep, gamma, mean, invstd, dgamma, dbeta,
                               accumulate_params, param_scale, npix, C, groups, count, act, dtype, stream);
}

// ... with the two reductions given as f64 [nrep][groups][2][C] (sum(gm*x), sum(gm)): the layout the epilogue of
// sdhip_conv2d_fwd_bnbwd adds them in.
extern "C" int sdhip_bn_bwd_apply_fin_d(const void* gy, int ldg, const void* x, int ldx, void* gx, int ldgx,
                                        const float* scale, const float* shift, const double* sums, int nrep,
                                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                        int accumulate_params, float param_scale, long npix, int C, int groups, double count, int act, int dtype,
                                        void* stream) {
  return bn_bwd_apply_fin_impl(gy, ldg, x, ldx, gx, ldgx, scale, shift, nullptr, nullptr, sums, nrep, gamma, mean, invstd, dgamma, dbeta,
                               accumulate_params, param_scale, npix, C, groups, count, act, dtype, stream);
}

extern "C" int sdhip_channel_stats(const void* x, int ldx, double* stats, int ldc, int nrep, long npix, int C, int groups,
                                   int zero_first, int dtype, void* stream) {
  if (ldc <= 0) ldc = C;
  if (nrep < 1) nrep = 1;
  const int G = groups;
  if (int rc = check_rows("channel_stats", npix, C, G, dtype)) return rc;
  SDHIP_CHECK_ARG(x && stats && ldx >= C, "channel_stats: bad pointers/strides");
  hipStream_t s = (hipStream_t)stream;
  if (zero_first) {
    // kernels, not memset nodes (sdhip_common.h): dense replicas in one launch, a strided slice row by row
    const size_t rows = 2 * (size_t)G * nrep;
    if (ldc == C) {
      if (sdhip_zero_async(stats, sizeof(double) * (size_t)C * rows, s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "channel_stats: clear failed");
    } else {
      for (size_t r = 0; r < rows; ++r)
        if (sdhip_zero_async(stats + r * (size_t)ldc, sizeof(double) * (size_t)C, s) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "channel_stats: clear failed");
    }
  }
#define ARGS(T) (const T*)x, ldx, stats, ldc, nrep, C, npix / G
  if (dtype == SDHIP_F32) {
    const bool v = vec_rows<float>(C, {ldx}, {x});
    Plan pl = plan(v ? C / 4 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((channel_stats_kernel<float, true>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
    else hipLaunchKernelGGL((channel_stats_kernel<float, false>), pl.grid, dim3(256), 0, s, ARGS(float), pl.rg);
  } else {
    const bool v = vec_rows<bf16_t>(C, {ldx}, {x});
    Plan pl = plan(v ? C / 8 : C, npix / G, G, 1024);
    if (v) hipLaunchKernelGGL((channel_stats_kernel<bf16_t, true>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
    else hipLaunchKernelGGL((channel_stats_kernel<bf16_t, false>), pl.grid, dim3(256), 0, s, ARGS(bf16_t), pl.rg);
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_bn_finalize(const double* stats, int ldc, int nrep, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var,
                                 float* scale, float* shift, float* mean_out, float* invstd_out,
                                 int C, int groups, double count, float eps, float momentum, void* stream) {
  SDHIP_CHECK_ARG(C > 0 && groups >= 1 && scale && shift, "bn_finalize: bad arguments");
  SDHIP_CHECK_ARG(stats || (running_mean && running_var), "bn_finalize: eval mode needs running statistics");
  SDHIP_CHECK_ARG(!stats || (mean_out && invstd_out && count >= 1.), "bn_finalize: train mode needs mean/invstd outputs and a count");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(sdhip_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, stats, ldc > 0 ? ldc : C, nrep > 0 ? nrep : 1, gamma, beta,
                     running_mean, running_var, scale, shift, mean_out, invstd_out, C, groups, count, eps, momentum);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// bn_finalize_bwd of one DenseNet layer's norm1 + stats_fix of the NEXT layer to be processed, in one launch.
// The backward walk of a dense block (models/densenet.py:41-45) alternates a per-channel kernel (the statistics gradient
// dS of the slab channels a layer read) with an elementwise one (dy = g + dS0 + 2 x dS1 on the 32 channels the previous
// layer wrote, which needs exactly the top 32 of the dS just updated).  Here every workgroup re-derives the 32 coefficients
// it needs from the replica sums (a few KB from L2) while the first ceil(Cf/32) workgroups also do the per-channel work for
// all Cf channels (dgamma, dbeta, dS of the channels below the slice — the slice's own dS is consumed here and never again).
template <typename T>
__global__ __launch_bounds__(256) void stats_fix_fin_kernel(const T* gin, int ldgi, const T* __restrict__ x, int ldx, T* gout, int ldgo,
                                                            long npix_g, RowGeom rg, double* __restrict__ dS, int ldc, int cs,
                                                            const float* __restrict__ dscale, const float* __restrict__ dshift, int nrep,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int acc_par, float pscale, int Cf, int G, double inv_count) {
  constexpr int N = Unit<T, true>::N;
  __shared__ float red[2][8][32];
  __shared__ float fa[32], fb[32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int nfb = (Cf + 31) >> 5;
  if (blockIdx.z == 0 && blockIdx.y == 0 && (int)blockIdx.x < nfb) {     // workgroup-uniform: per-channel work of block blockIdx.x
    const int c = blockIdx.x * 32 + cl;
    const bool okc = c < Cf;
    float dg = 0.f, db = 0.f;
    for (int g = 0; g < G; ++g) {
      float ds = 0.f, dh = 0.f;
      if (okc) {
#pragma unroll 4
        for (int r = rl; r < nrep; r += 8) { ds += dscale[((long)r * G + g) * Cf + c]; dh += dshift[((long)r * G + g) * Cf + c]; }
      }
      __syncthreads();
      red[0][rl][cl] = ds; red[1][rl][cl] = dh;
      __syncthreads();
      if (rl == 0 && okc) {
        ds = 0.f; dh = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { ds += red[0][r][cl]; dh += red[1][r][cl]; }
        const float gm = gamma[c];
        const float mu = mean[g * Cf + c], inv = invstd[g * Cf + c];
        const float t = ds - mu * dh;
        dg += inv * t;
        db += dh;
        if (c < cs || c >= cs + 32) {                 // the slice's dS is formed (and consumed) by the elementwise part below
          const double dinv = (double)gm * t;
          const double dvar = -0.5 * dinv * (double)inv * inv * inv;
          const double dmu = -(double)gm * inv * dh - 2.0 * mu * dvar;
          dS[((long)g * 2 + 0) * ldc + c] += dmu * inv_count;
          dS[((long)g * 2 + 1) * ldc + c] += dvar * inv_count;
        }
      }
    }
    if (rl == 0 && okc) {
      dg *= pscale; db *= pscale;                    // (see bn_bwd_apply_fin_kernel)
      dgamma[c] = acc_par ? dgamma[c] + dg : dg;
      dbeta[c] = acc_par ? dbeta[c] + db : db;
    }
  }
  // coefficients of the slice channels cs .. cs+31 for this workgroup's statistics group
  const int g = blockIdx.z;
  {
    const int c = cs + cl;
    float ds = 0.f, dh = 0.f;
#pragma unroll 4
    for (int r = rl; r < nrep; r += 8) { ds += dscale[((long)r * G + g) * Cf + c]; dh += dshift[((long)r * G + g) * Cf + c]; }
    __syncthreads();
    red[0][rl][cl] = ds; red[1][rl][cl] = dh;
    __syncthreads();
    if (rl == 0) {
      ds = 0.f; dh = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) { ds += red[0][r][cl]; dh += red[1][r][cl]; }
      const float gm = gamma[c];
      const float mu = mean[g * Cf + c], inv = invstd[g * Cf + c];
      const float t = ds - mu * dh;
      const double dinv = (double)gm * t;
      const double dvar = -0.5 * dinv * (double)inv * inv * inv;
      const double dmu = -(double)gm * inv * dh - 2.0 * mu * dvar;
      fa[cl] = (float)(dS[((long)g * 2 + 0) * ldc + c] + dmu * inv_count);
      fb[cl] = (float)(2.0 * (dS[((long)g * 2 + 1) * ldc + c] + dvar * inv_count));
    }
    __syncthreads();
  }
  ROW_TXTY(rg, tx, ty);
  const int u = blockIdx.y * rg.tx + tx;
  if (ty >= rg.ty || u >= rg.units) return;
  const int c0 = u * N;
  float a[N], b2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { a[e] = fa[c0 + e]; b2[e] = fb[c0 + e]; }
  const long base = (long)g * npix_g;
  for (long pix = (long)blockIdx.x * rg.ty + ty; pix < npix_g; pix += (long)gridDim.x * rg.ty) {
    float gv[N], xv[N];
    Unit<T, true>::load(gin + (base + pix) * ldgi + c0, gv);
    Unit<T, true>::load(x + (base + pix) * ldx + c0, xv);
#pragma unroll
    for (int e = 0; e < N; ++e) gv[e] = gv[e] + fmaf(xv[e], b2[e], a[e]);
    Unit<T, true>::store(gout + (base + pix) * ldgo + c0, gv);
  }
}

extern "C" int sdhip_stats_fix_fin(const void* gin, int ldgi, const void* x, int ldx, void* gout, int ldgo, long npix,
                                   double* dS, int ldc, int cs, const float* dscale, const float* dshift, int nrep,
                                   const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                   int accumulate_params, float param_scale, int Cf, int groups, double count, int dtype, void* stream) {
  const int G = groups;
  if (int rc = check_rows("stats_fix_fin", npix, 32, G, dtype)) return rc;
  SDHIP_CHECK_ARG(gin && x && gout && dS && dscale && dshift && gamma && mean && invstd && dgamma && dbeta && ldgi >= 32 && ldx >= 32 &&
                  ldgo >= 32 && cs >= 0 && cs + 32 <= Cf && ldc >= Cf && nrep >= 1 && count > 0., "stats_fix_fin: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int nfb = sdhip_cdiv(Cf, 32);
#define ARGS(T) (const T*)gin, ldgi, (const T*)x, ldx, (T*)gout, ldgo, npix / G, pl.rg, dS, ldc, cs, dscale, dshift, nrep, gamma, mean, invstd, \
                dgamma, dbeta, accumulate_params, param_scale, Cf, G, 1.0 / count
  if (dtype == SDHIP_F32) {
    SDHIP_CHECK_ARG((vec_rows<float>(32, {ldgi, ldx, ldgo}, {gin, x, gout})), "stats_fix_fin: rows must be 16-byte aligned");
    Plan pl = plan(32 / 4, npix / G, G);
    if ((int)pl.grid.x < nfb) pl.grid.x = nfb;            // enough workgroups for the per-channel part
    hipLaunchKernelGGL((stats_fix_fin_kernel<float>), pl.grid, dim3(256), 0, s, ARGS(float));
  } else {
    SDHIP_CHECK_ARG((vec_rows<bf16_t>(32, {ldgi, ldx, ldgo}, {gin, x, gout})), "stats_fix_fin: rows must be 16-byte aligned");
    Plan pl = plan(32 / 8, npix / G, G);
    if ((int)pl.grid.x < nfb) pl.grid.x = nfb;
    hipLaunchKernelGGL((stats_fix_fin_kernel<bf16_t>), pl.grid, dim3(256), 0, s, ARGS(bf16_t));
  }
#undef ARGS
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

extern "C" int sdhip_bn_finalize_bwd(const float* dscale, const float* dshift, int nrep, const float* gamma,
                                     const float* mean, const float* invstd,
                                     float* dgamma, float* dbeta, double* dstats, int ldc, int accumulate_dstats,
                                     int C, int groups, double count, int train, void* stream) {
  SDHIP_CHECK_ARG(C > 0 && groups >= 1 && dscale && dshift && mean && invstd, "bn_finalize_bwd: bad arguments");
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(sdhip_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                     nrep > 0 ? nrep : 1, gamma, mean, invstd, dgamma, dbeta, dstats, ldc > 0 ? ldc : C, accumulate_dstats, C, groups,
                     1.0 / count, train);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
