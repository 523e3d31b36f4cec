/*
 * ORACLE (test infrastructure only — never linked into or called by the product).
 *
 * Plain-C restatement of the spatial correlation sampler used by the reference at
 * models/dsnet_t2.py:1078-1087 (ctor) and :1188-1193, :1233-1234, :221-223 (calls).
 *
 * The arithmetic lives in a third-party dependency that is absent from
 * /root/reference: PyPI `spatial-correlation-sampler` (upstream
 * ClementPinard/Pytorch-Correlation-extension), installed UNPINNED by
 * scripts/scriptsDocker/Torch/Dockerfile:53.  This file restates that package's
 * published CPU algorithm (correlation.cpp: correlate_patch / correlate_patch_grad,
 * output (B, PH, PW, oH, oW), displacement applied to input2, zero outside the
 * image, no normalisation).  No reference test pins the op's output:
 * PARITY UNPINNED at this boundary — the pin is the reference's usage contract
 * (rank-5 output, squeeze(1) valid for patch (1,17), spatial size preserved with
 * padding 0, un-normalised in 1-D mode and divided by C by the caller in 2-D mode).
 *
 * Layout: NCHW contiguous float (the reference's layout).
 */
#include <stddef.h>
#include <string.h>

static int out_size(int in, int pad, int k, int dil, int stride) {
  return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}

#define IN(p, n, c, i, j) p[(((size_t)(n) * C + (c)) * H + (i)) * W + (j)]

void corr_ref_forward(const float* in1, const float* in2, float* out,
                      int B, int C, int H, int W,
                      int kH, int kW, int PH, int PW,
                      int padH, int padW, int dilH, int dilW,
                      int dilPH, int dilPW, int dH, int dW) {
  const int oH = out_size(H, padH, kH, dilH, dH), oW = out_size(W, padW, kW, dilW, dW);
  const int radH = (PH - 1) / 2 * dilPH, radW = (PW - 1) / 2 * dilPW;
  for (int n = 0; n < B; ++n)
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int h = 0; h < oH; ++h)
          for (int w = 0; w < oW; ++w) {
            const int u = -padH + h * dH, v = -padW + w * dW;
            const int su = ph * dilPH - radH, sv = pw * dilPW - radW;
            float acc = 0.f;
            for (int c = 0; c < C; ++c)
              for (int i = 0; i < kH; ++i) {
                const int i1 = u + i * dilH, i2 = i1 + su;
                if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
                for (int j = 0; j < kW; ++j) {
                  const int j1 = v + j * dilW, j2 = j1 + sv;
                  if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
                  acc += IN(in1, n, c, i1, j1) * IN(in2, n, c, i2, j2);
                }
              }
            out[((((size_t)n * PH + ph) * PW + pw) * oH + h) * oW + w] = acc;
          }
}

void corr_ref_backward(const float* in1, const float* in2, const float* gout,
                       float* gin1, float* gin2,
                       int B, int C, int H, int W,
                       int kH, int kW, int PH, int PW,
                       int padH, int padW, int dilH, int dilW,
                       int dilPH, int dilPW, int dH, int dW) {
  const int oH = out_size(H, padH, kH, dilH, dH), oW = out_size(W, padW, kW, dilW, dW);
  const int radH = (PH - 1) / 2 * dilPH, radW = (PW - 1) / 2 * dilPW;
  memset(gin1, 0, sizeof(float) * (size_t)B * C * H * W);
  memset(gin2, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int n = 0; n < B; ++n)
    for (int ph = 0; ph < PH; ++ph)
      for (int pw = 0; pw < PW; ++pw)
        for (int h = 0; h < oH; ++h)
          for (int w = 0; w < oW; ++w) {
            const float g = gout[((((size_t)n * PH + ph) * PW + pw) * oH + h) * oW + w];
            const int u = -padH + h * dH, v = -padW + w * dW;
            const int su = ph * dilPH - radH, sv = pw * dilPW - radW;
            for (int c = 0; c < C; ++c)
              for (int i = 0; i < kH; ++i) {
                const int i1 = u + i * dilH, i2 = i1 + su;
                if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
                for (int j = 0; j < kW; ++j) {
                  const int j1 = v + j * dilW, j2 = j1 + sv;
                  if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
                  IN(gin1, n, c, i1, j1) += g * IN(in2, n, c, i2, j2);
                  IN(gin2, n, c, i2, j2) += g * IN(in1, n, c, i1, j1);
                }
              }
          }
}
