// On-device sample preparation for gfx950.
//
// The reference's CustomDataset.__getitem__ (util/utilTorchDataLoader.py:133-274) runs, per sample and on ONE CPU worker
// (torch_implementation.py:752-756): PFM decode + flipud (util/utilIOPfm.py:66-101), depth -> disparity
// (`np.where(d > 0, f*b/d, 0)`, :173-181), the output-activation variants (:188-197), a per-class python loop that
// builds the one-hot target (:199-211) or the cityscapes id -> trainId table (util/utilCityscape.py:173-186), the crop
// (RandomCrop, :435-463), `x/255 - mean)/std` in float64 (:247-248) and the HWC -> CHW transposes of ToTensor (:608-630).
// Here the host only decodes PNG containers and picks the crop offsets; everything above is ONE kernel that reads the
// raw uint8 / PFM bytes once and writes the batch slot of the four training tensors directly in the layout the
// network consumes (NHWC, pixel stride ld) — pure HBM-bound byte work, one thread per output pixel.
#include "sdhip_common.h"

namespace {

struct PrepArgs {
  const unsigned char* left; const unsigned char* right; long img_pitch; int img_cs;
  const unsigned char* seg; long seg_pitch; int seg_cs; int seg_channel; int seg_mode; int seg_threshold;
  const unsigned char* lut;
  const unsigned char* depth; long depth_pitch; int depth_mode; int depth_flip;
  int H, W, top, left0, oh, ow, roll;
  float fb, max_d; int activation;
  double mean[3], stdv[3];
  void* out_left; void* out_right; int ld_img;
  float* out_seg; int ld_seg; int n_seg;
  float* out_disp;
};

template <typename T>
__global__ __launch_bounds__(256) void prepare_sample_kernel(const PrepArgs a) {
  const long n = (long)a.oh * a.ow;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    const int oy = (int)(p / a.ow), ox = (int)(p - (long)oy * a.ow);
    // sliceandSwitch (RandomCrop, util/utilTorchDataLoader.py:455-467): the cropped maps are cut at row `roll` and the two
    // pieces exchanged — output row oy is crop row (oy + roll) mod oh
    int cy = oy + a.roll;
    if (cy >= a.oh) cy -= a.oh;
    const int y = a.top + cy, x = a.left0 + ox;
    if (a.left) {
      // ((x / 255.0 - mean) / std).astype(float32): uint8 / python float is float64 in numpy, so is the rest
      const unsigned char* lp = a.left + y * a.img_pitch + (long)x * a.img_cs;
      const unsigned char* rp = a.right + y * a.img_pitch + (long)x * a.img_cs;
      T* ol = (T*)a.out_left + p * a.ld_img;
      T* orr = (T*)a.out_right + p * a.ld_img;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        Elem<T>::st(ol + c, (float)(((double)lp[c] / 255.0 - a.mean[c]) / a.stdv[c]));
        Elem<T>::st(orr + c, (float)(((double)rp[c] / 255.0 - a.mean[c]) / a.stdv[c]));
      }
    }
    if (a.seg) {
      const unsigned char v = a.seg[y * a.seg_pitch + (long)x * a.seg_cs + a.seg_channel];
      int cls;
      if (a.seg_mode == SDHIP_SEG_THRESHOLD) cls = v > a.seg_threshold ? 1 : 0;   // roses: `seg > 128` on channel 2, class j <=> (binary == j)
      else if (a.seg_mode == SDHIP_SEG_ID_PLUS_ONE) cls = (int)v - 1;            // garden: channel j <=> (seg == j + 1)
      else cls = a.lut[v];                                                        // cityscapes / kitti: trainId table, 255 -> channel n_labels
      float* os = a.out_seg + p * a.ld_seg;
      for (int j = 0; j < a.n_seg; ++j) os[j] = (j == cls) ? 1.f : 0.f;
    }
    if (a.depth) {
      float d;
      if (a.depth_mode == SDHIP_DEPTH_PFM) {
        // PFM rows are stored bottom-up (np.flipud after the reshape); the header's sign gives the byte order
        const int ry = a.H - 1 - y;
        unsigned int u = *reinterpret_cast<const unsigned int*>(a.depth + ry * a.depth_pitch + (long)x * 4);
        if (a.depth_flip) u = __builtin_bswap32(u);
        const float z = __builtin_bit_cast(float, u);
        d = z > 0.f ? a.fb / z : 0.f;                // np.where(depth > 0, f*b*1/depth, 0) in float32
      } else {
        // 16-bit disparity PNG: `disp.astype(np.float32) / 256.0`
        const unsigned short u = *reinterpret_cast<const unsigned short*>(a.depth + y * a.depth_pitch + (long)x * 2);
        d = (float)u / 256.f;
      }
      if (a.activation != SDHIP_ACT_LINEAR && d > a.max_d) d = a.max_d;             // `disp_image[disp_image > max_d] = max_d`
      if (a.activation == SDHIP_ACT_SIGMOID) d = d / a.max_d;
      else if (a.activation == SDHIP_ACT_TANH) d = d != 0.f ? 2.f * d / a.max_d - 1.f : -1.f;
      a.out_disp[p] = d;
    }
  }
}

}  // namespace

extern "C" int sdhip_prepare_sample(const unsigned char* left, const unsigned char* right, long img_pitch, int img_cs,
                                    const unsigned char* seg, long seg_pitch, int seg_cs, int seg_channel, int seg_mode,
                                    int seg_threshold, const unsigned char* lut, const void* depth, long depth_pitch,
                                    int depth_mode, int depth_big_endian, int H, int W, int crop_top, int crop_left,
                                    int out_h, int out_w, int row_roll, float fb, float max_d, int activation, const float* mean,
                                    const float* stdv, void* out_left, void* out_right, int ld_img, float* out_seg,
                                    int ld_seg, int n_seg, float* out_disp, int dtype, void* stream) {
  SDHIP_CHECK_ARG(H > 0 && W > 0 && out_h > 0 && out_w > 0, "prepare_sample: empty image");
  SDHIP_CHECK_ARG(crop_top >= 0 && crop_left >= 0 && crop_top + out_h <= H && crop_left + out_w <= W,
                  "prepare_sample: crop %dx%d at (%d,%d) leaves the %dx%d image", out_h, out_w, crop_top, crop_left, H, W);
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "prepare_sample: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG(row_roll >= 0 && row_roll < out_h, "prepare_sample: row_roll %d outside [0, %d)", row_roll, out_h);
  SDHIP_CHECK_ARG(left || seg || depth, "prepare_sample: nothing to do");
  if (left) SDHIP_CHECK_ARG(right && out_left && out_right && mean && stdv && img_cs >= 3 && img_pitch >= (long)W * img_cs && ld_img >= 3,
                            "prepare_sample: image arguments (cs=%d pitch=%ld ld=%d)", img_cs, img_pitch, ld_img);
  if (seg) {
    SDHIP_CHECK_ARG(out_seg && seg_cs >= 1 && seg_channel >= 0 && seg_channel < seg_cs && seg_pitch >= (long)W * seg_cs && n_seg >= 1 && ld_seg >= n_seg,
                    "prepare_sample: segmentation arguments (cs=%d channel=%d pitch=%ld n=%d ld=%d)", seg_cs, seg_channel, seg_pitch, n_seg, ld_seg);
    SDHIP_CHECK_ARG(seg_mode == SDHIP_SEG_THRESHOLD || seg_mode == SDHIP_SEG_ID_PLUS_ONE || (seg_mode == SDHIP_SEG_LUT && lut),
                    "prepare_sample: unknown segmentation mode %d (or missing table)", seg_mode);
  }
  if (depth) {
    SDHIP_CHECK_ARG(out_disp && (depth_mode == SDHIP_DEPTH_PFM || depth_mode == SDHIP_DEPTH_U16) &&
                    depth_pitch >= (long)W * (depth_mode == SDHIP_DEPTH_PFM ? 4 : 2) && (depth_pitch % (depth_mode == SDHIP_DEPTH_PFM ? 4 : 2)) == 0 &&
                    ((uintptr_t)depth % (depth_mode == SDHIP_DEPTH_PFM ? 4 : 2)) == 0,
                    "prepare_sample: depth arguments (mode=%d pitch=%ld; rows must be element-aligned)", depth_mode, depth_pitch);
    SDHIP_CHECK_ARG(activation == SDHIP_ACT_LINEAR || ((activation == SDHIP_ACT_SIGMOID || activation == SDHIP_ACT_TANH) && max_d > 0.f),
                    "prepare_sample: unknown output activation %d (or max_d <= 0)", activation);
  }
  PrepArgs a;
  a.left = left; a.right = right; a.img_pitch = img_pitch; a.img_cs = img_cs;
  a.seg = seg; a.seg_pitch = seg_pitch; a.seg_cs = seg_cs; a.seg_channel = seg_channel; a.seg_mode = seg_mode; a.seg_threshold = seg_threshold;
  a.lut = lut;
  a.depth = (const unsigned char*)depth; a.depth_pitch = depth_pitch; a.depth_mode = depth_mode; a.depth_flip = depth_big_endian;   // the GPU is little-endian
  a.H = H; a.W = W; a.top = crop_top; a.left0 = crop_left; a.oh = out_h; a.ow = out_w; a.roll = row_roll;
  a.fb = fb; a.max_d = max_d; a.activation = activation;
  for (int c = 0; c < 3; ++c) { a.mean[c] = left ? (double)mean[c] : 0.0; a.stdv[c] = left ? (double)stdv[c] : 1.0; }
  a.out_left = out_left; a.out_right = out_right; a.ld_img = ld_img;
  a.out_seg = out_seg; a.ld_seg = ld_seg; a.n_seg = n_seg; a.out_disp = out_disp;
  const long n = (long)out_h * out_w;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(prepare_sample_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(prepare_sample_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}

// ---- horizontal flip of a prepared stereo sample (RandomCrop(flipHorizontal=True), util/utilTorchDataLoader.py:476-499) ----
// The reference, on the cropped HWC arrays (cityscapes only):
//   left' = fliplr(right), right' = fliplr(left);
//   every pixel (r, c) of disp and seg is moved to column max(int(c - disp[r,c]), 0) of the SAME arrays, in row-major
//   order (a fancy-index assignment: where several pixels land on one column the LAST one, i.e. the largest c, wins;
//   columns nobody lands on keep their old content);
//   disp[:, -10:] = 0, seg[:, -20:] = 0; mask = (disp == 0); the void channel (last) = mask, all others *= 1 - mask;
//   disp and seg are flipped left-right.
// Three kernels per sample: the winner per (row, column) by atomicMax, the shifted / masked / flipped maps into scratch,
// and the copy back together with the swap + flip of the two images.
namespace {

__global__ __launch_bounds__(256) void flip_init_kernel(int* __restrict__ winner, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) winner[i] = -1;
}

__global__ __launch_bounds__(256) void flip_winner_kernel(const float* __restrict__ disp, int* __restrict__ winner, int H, int W) {
  const long n = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % W);
    const long r = i / W;
    int t = (int)((double)c - (double)disp[i]);      // numpy: int64 column minus float32 disparity = float64, .astype(int) truncates
    t = t < 0 ? 0 : t;
    if (t < W) atomicMax(winner + r * W + t, c);
  }
}

__global__ __launch_bounds__(256) void flip_maps_kernel(const float* __restrict__ disp, const float* __restrict__ seg, int ld_seg, int n_seg,
                                                        const int* __restrict__ winner, float* __restrict__ dtmp, float* __restrict__ stmp,
                                                        int H, int W) {
  const long n = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long r = i / W;
    const int wsrc = winner[i];
    const long src = wsrc >= 0 ? r * W + wsrc : i;
    float d = disp[src];
    if (x >= W - 10) d = 0.f;
    const float mask = d == 0.f ? 1.f : 0.f;
    const long dst = r * W + (W - 1 - x);            // the final left-right flip
    dtmp[dst] = d;
    for (int k = 0; k < n_seg; ++k) {
      float v = x >= W - 20 ? 0.f : seg[src * ld_seg + k];
      v = k == n_seg - 1 ? mask : v * (1.f - mask);
      stmp[dst * n_seg + k] = v;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void flip_apply_kernel(T* __restrict__ left, T* __restrict__ right, int ld_img, float* __restrict__ disp,
                                                         float* __restrict__ seg, int ld_seg, int n_seg, const float* __restrict__ dtmp,
                                                         const float* __restrict__ stmp, int H, int W) {
  const long n = (long)H * W;
  const int half = (W + 1) / 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long r = i / W;
    disp[i] = dtmp[i];
    for (int k = 0; k < n_seg; ++k) seg[i * ld_seg + k] = stmp[i * n_seg + k];
    if (x < half) {                                  // this thread owns the column pair (x, W-1-x) of both images
      const long a = (r * W + x) * ld_img, b = (r * W + (W - 1 - x)) * ld_img;
      for (int ch = 0; ch < 3; ++ch) {
        const T la = left[a + ch], lb = left[b + ch], ra = right[a + ch], rb = right[b + ch];
        left[a + ch] = rb; left[b + ch] = ra;
        right[a + ch] = lb; right[b + ch] = la;
      }
    }
  }
}

}  // namespace

extern "C" long sdhip_flip_sample_workspace_bytes(int H, int W, int n_seg) {
  if (H <= 0 || W <= 0 || n_seg <= 0) return 0;
  return (long)H * W * 4 * (2 + n_seg);
}

extern "C" int sdhip_flip_sample(void* left, void* right, int ld_img, float* seg, int ld_seg, int n_seg, float* disp, int H, int W,
                                 void* workspace, long workspace_bytes, int dtype, void* stream) {
  SDHIP_CHECK_ARG(left && right && seg && disp && workspace && H > 0 && W > 20 && n_seg >= 2 && ld_img >= 3 && ld_seg >= n_seg,
                  "flip_sample: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "flip_sample: unknown dtype %d", dtype);
  SDHIP_CHECK_ARG(workspace_bytes >= sdhip_flip_sample_workspace_bytes(H, W, n_seg), "flip_sample: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const long n = (long)H * W;
  int* winner = (int*)workspace;
  float* dtmp = (float*)workspace + n;
  float* stmp = dtmp + n;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(flip_init_kernel, dim3((unsigned)blocks), dim3(256), 0, s, winner, n);
  hipLaunchKernelGGL(flip_winner_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)disp, winner, H, W);
  hipLaunchKernelGGL(flip_maps_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)disp, (const float*)seg, ld_seg, n_seg,
                     (const int*)winner, dtmp, stmp, H, W);
  if (dtype == SDHIP_F32)
    hipLaunchKernelGGL(flip_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)left, (float*)right, ld_img, disp, seg, ld_seg,
                       n_seg, (const float*)dtmp, (const float*)stmp, H, W);
  else
    hipLaunchKernelGGL(flip_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (bf16_t*)left, (bf16_t*)right, ld_img, disp, seg, ld_seg,
                       n_seg, (const float*)dtmp, (const float*)stmp, H, W);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}


// ---- augment_DoubleLeftImg (RandomCrop, util/utilTorchDataLoader.py:469-474): with probability 0.1 a sample becomes a
// zero-disparity pair — left mirrored, right = the mirrored left, disparity 0.0001 everywhere, one-hot map mirrored.
// In place on one prepared batch slot; a thread owns the pixel pair (x, W-1-x) of a row, so the mirror needs no scratch.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void double_left_kernel(T* __restrict__ left, T* __restrict__ right, int ld_img, float* __restrict__ seg,
                                                          int ld_seg, int n_seg, float* __restrict__ disp, int H, int W) {
  const int half = (W + 1) / 2;
  const long n = (long)H * half;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % half);
    const long r = i / half;
    const long pa = r * W + x, pb = r * W + (W - 1 - x);
    if (left) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const T a = left[pa * ld_img + c], b = left[pb * ld_img + c];
        left[pa * ld_img + c] = b; left[pb * ld_img + c] = a;
        right[pa * ld_img + c] = b; right[pb * ld_img + c] = a;
      }
    }
    if (seg) {
      for (int c = 0; c < n_seg; ++c) {
        const float a = seg[pa * ld_seg + c], b = seg[pb * ld_seg + c];
        seg[pa * ld_seg + c] = b; seg[pb * ld_seg + c] = a;
      }
    }
    if (disp) { disp[pa] = 0.0001f; disp[pb] = 0.0001f; }      // np.zeros_like(disp) + 0.0001 in float32
  }
}
}  // namespace

extern "C" int sdhip_double_left_sample(void* left, void* right, int ld_img, float* seg, int ld_seg, int n_seg, float* disp,
                                        int H, int W, int dtype, void* stream) {
  SDHIP_CHECK_ARG(H > 0 && W > 0 && (left || seg || disp), "double_left_sample: bad arguments");
  SDHIP_CHECK_ARG(dtype == SDHIP_F32 || dtype == SDHIP_BF16, "double_left_sample: unknown dtype %d", dtype);
  if (left) SDHIP_CHECK_ARG(right && ld_img >= 3, "double_left_sample: image arguments");
  if (seg) SDHIP_CHECK_ARG(n_seg >= 1 && ld_seg >= n_seg, "double_left_sample: segmentation arguments");
  const long n = (long)H * ((W + 1) / 2);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == SDHIP_F32) hipLaunchKernelGGL(double_left_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)left, (float*)right, ld_img, seg, ld_seg, n_seg, disp, H, W);
  else hipLaunchKernelGGL(double_left_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)left, (bf16_t*)right, ld_img, seg, ld_seg, n_seg, disp, H, W);
  SDHIP_LAUNCH_CHECK();
  return SDHIP_OK;
}
