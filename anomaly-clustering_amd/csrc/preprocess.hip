// Input pipeline on gfx950: decoded uint8 RGB images -> the fp32 NCHW tensor ac_vit_prepare_tokens reads.
//
// Reference arithmetic: models/patchcore/datasets/mvtec.py:74-81 -
//   transforms.Resize(256) -> CenterCrop(224) -> ToTensor() -> Normalize(ImageNet mean, std)
// applied to one PIL image at a time on the host (DataLoader batch size 1, examples/main.py:213), twice per
// image (examples/main.py:254-262 and :266).  For PIL inputs torchvision's Resize is Pillow's
// Image.resize(BILINEAR): a two-pass (horizontal, then vertical) triangle filter whose support is scaled by
// the reduction factor, float64 weights normalised and rounded to 22-bit fixed point, int32 accumulation
// from a 2^21 rounding bias, shift, clamp to uint8 after EACH pass (Pillow src/libImaging/Resample.c).  That is
// integer arithmetic, so the kernels below reproduce it bit for bit (tests/test_preprocess.py checks them
// against Pillow itself and against the numpy restatement in oracle/preprocess.py); ToTensor / Normalize are
// one fp32 division, one subtraction and one division per value, IEEE-rounded (-ffp-contract=off, correctly
// rounded fp32 division is hipcc's default).  Only the 224 x 224 window CenterCrop keeps is computed.
//
// HBM-bound and tiny next to the ViT (2.4 MB read + 0.6 MB written per 900 x 900 image); what it removes is
// the host-side per-image Pillow resize and the second pass over the dataset.
#include "common.h"

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct AxisPlan {          // one axis of the resize, restricted to the cropped window
  int in_size, out_size, first_out, count, ksize;
  double scale, filterscale, support;
};

static AxisPlan plan_axis(int in_size, int out_size, int first_out, int count) {
  AxisPlan p;
  p.in_size = in_size; p.out_size = out_size; p.first_out = first_out; p.count = count;
  p.scale = (double)in_size / (double)out_size;
  p.filterscale = p.scale < 1.0 ? 1.0 : p.scale;
  p.support = 1.0 * p.filterscale;                       // bilinear: support 1
  p.ksize = (int)ceil(p.support) * 2 + 1;
  return p;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, one thread per output index of the window.
__global__ void resize_coeffs_kernel(AxisPlan p, int* __restrict__ bounds, int* __restrict__ kk) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= p.count) return;
  const int xx = p.first_out + o;
  const double center = ((double)xx + 0.5) * p.scale;
  const double ss = 1.0 / p.filterscale;
  int xmin = (int)(center - p.support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + p.support + 0.5);
  if (xmax > p.in_size) xmax = p.in_size;
  xmax -= xmin;
  int* k = kk + (size_t)o * p.ksize;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double t = ((double)(x + xmin) - center + 0.5) * ss;
    if (t < 0.0) t = -t;
    ww += t < 1.0 ? 1.0 - t : 0.0;
  }
  for (int x = 0; x < p.ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double t = ((double)(x + xmin) - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      w = t < 1.0 ? 1.0 - t : 0.0;
      if (ww != 0.0) w /= ww;
    }
    k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
  }
  bounds[o * 2] = xmin;
  bounds[o * 2 + 1] = xmax;
}

__device__ inline unsigned clip8(int v) {
  v >>= PRECISION_BITS;
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass: block = (input row y0 + blockIdx.x, image blockIdx.y); thread -> (ox, c).
// tmp[b][row][ox][c] uint8.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ img, int H, int W, int y0, int crop,
                                                       int ksize, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, uint8_t* __restrict__ tmp, int rows) {
  const int r = blockIdx.x, b = blockIdx.y;
  const uint8_t* src = img + ((size_t)b * H + (y0 + r)) * W * 3;
  uint8_t* dst = tmp + ((size_t)b * rows + r) * crop * 3;
  for (int e = threadIdx.x; e < crop * 3; e += 256) {
    const int ox = e / 3, c = e - ox * 3;
    const int xmin = bounds[ox * 2], n = bounds[ox * 2 + 1];
    const int* k = kk + (size_t)ox * ksize;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < n; ++x) ss += (int)src[(xmin + x) * 3 + c] * k[x];
    dst[e] = (uint8_t)clip8(ss);
  }
}

// Vertical pass + ToTensor + Normalize: block = (output row oy, image); thread -> (c, ox), NCHW store.
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int rows, int y0, int crop,
                                                            int ksize, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, float m0, float m1, float m2,
                                                            float s0, float s1, float s2, float* __restrict__ out) {
  const int oy = blockIdx.x, b = blockIdx.y;
  const int ymin = bounds[oy * 2] - y0, n = bounds[oy * 2 + 1];
  const int* k = kk + (size_t)oy * ksize;
  const uint8_t* src = tmp + (size_t)b * rows * crop * 3;
  for (int e = threadIdx.x; e < crop * 3; e += 256) {
    const int c = e / crop, ox = e - c * crop;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < n; ++y) ss += (int)src[((size_t)(ymin + y) * crop + ox) * 3 + c] * k[y];
    const float x = (float)clip8(ss) / 255.0f;                          // ToTensor
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[(((size_t)b * 3 + c) * crop + oy) * crop + ox] = (x - mean) / sd;   // Normalize
  }
}

struct Geometry {
  int nw, nh, left, top, y0, rows;
  AxisPlan ph, pv;
};

static int geometry(int H, int W, int resize, int crop, Geometry& g) {
  if (W <= H) { g.nw = resize; g.nh = (int)((double)resize * H / W); }        // torchvision Resize(int): short edge
  else { g.nh = resize; g.nw = (int)((double)resize * W / H); }
  if (g.nw < crop || g.nh < crop) return ac_fail(AC_EINVAL, "ac_preprocess_images: resized image %dx%d smaller than the crop %d", g.nw, g.nh, crop);
  g.top = (int)nearbyint((g.nh - crop) / 2.0);                                  // CenterCrop: int(round(./2.0)), ties to even
  g.left = (int)nearbyint((g.nw - crop) / 2.0);
  g.ph = plan_axis(W, g.nw, g.left, crop);
  g.pv = plan_axis(H, g.nh, g.top, crop);
  // input rows the vertical pass of the window touches (same formula as the kernel; one row of margin)
  const double c0 = ((double)g.top + 0.5) * g.pv.scale, c1 = ((double)(g.top + crop - 1) + 0.5) * g.pv.scale;
  int lo = (int)(c0 - g.pv.support + 0.5) - 1, hi = (int)(c1 + g.pv.support + 0.5) + 1;
  if (lo < 0) lo = 0;
  if (hi > H) hi = H;
  g.y0 = lo;
  g.rows = hi - lo;
  return AC_OK;
}

extern "C" size_t ac_preprocess_workspace_bytes(int B, int H, int W, int resize, int crop) {
  Geometry g;
  if (B <= 0 || H <= 0 || W <= 0 || resize <= 0 || crop <= 0 || geometry(H, W, resize, crop, g)) return 256;
  size_t tables = align_up((size_t)crop * 2 * 4, 256) * 2 + align_up((size_t)crop * g.ph.ksize * 4, 256) +
                  align_up((size_t)crop * g.pv.ksize * 4, 256);
  return tables + align_up((size_t)B * g.rows * crop * 3, 256) + 256;
}

extern "C" int ac_preprocess_images(const uint8_t* images, int B, int H, int W, int resize, int crop, const float* mean,
                                    const float* stddev, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = as_stream(stream_);
  AC_REQUIRE(images && mean && stddev && out && ws, "ac_preprocess_images: null pointer");
  AC_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && resize > 0 && crop > 0 && crop <= 65535,
             "ac_preprocess_images: bad shape B=%d H=%d W=%d resize=%d crop=%d", B, H, W, resize, crop);
  Geometry g;
  if (int rc = geometry(H, W, resize, crop, g)) return rc;
  if (ws_bytes < ac_preprocess_workspace_bytes(B, H, W, resize, crop))
    return ac_fail(AC_EWORKSPACE, "ac_preprocess_images: workspace too small");
  Arena ar(ws, ws_bytes);
  int* bh = ar.take<int>((size_t)crop * 2);
  int* bv = ar.take<int>((size_t)crop * 2);
  int* kh = ar.take<int>((size_t)crop * g.ph.ksize);
  int* kv = ar.take<int>((size_t)crop * g.pv.ksize);
  uint8_t* tmp = ar.take<uint8_t>((size_t)B * g.rows * crop * 3);
  if (!ar.ok()) return ac_fail(AC_EWORKSPACE, "ac_preprocess_images: workspace carve failed");
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3((crop + 63) / 64), dim3(64), 0, stream, g.ph, bh, kh);
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3((crop + 63) / 64), dim3(64), 0, stream, g.pv, bv, kv);
  hipLaunchKernelGGL(resize_h_kernel, dim3(g.rows, B), dim3(256), 0, stream, images, H, W, g.y0, crop, g.ph.ksize, bh, kh,
                     tmp, g.rows);
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3(crop, B), dim3(256), 0, stream, tmp, g.rows, g.y0, crop, g.pv.ksize, bv, kv,
                     mean[0], mean[1], mean[2], stddev[0], stddev[1], stddev[2], out);
  return ac_check_launch("ac_preprocess_images");
}
