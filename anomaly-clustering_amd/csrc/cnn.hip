// Convolutional backbone pieces on gfx950: the layers of torchvision's wide_resnet50_2 that the clustering path
// runs (reference: models/patchcore/backbones.py:27, tapped at layer2 / layer3 by common.py:211-292).
//
// A convolution is a GEMM on the split-fp16 MFMA engine (gemm_hl5.h / gemm_hl2.h): activations live in HBM as
// NHWC fp32 (one row per pixel, channels contiguous); an im2col kernel writes the hl32 operand
// [B*Ho*Wo, kh*kw*C] (k = (ky*kw + kx)*C + c, so a thread copies 8 consecutive CHANNELS of one tap with two
// 16-byte loads; for a 1x1 stride-1 convolution that is just the hl32 split of the activation), the weight is the
// [Cout, kh*kw*C] hl32 matrix in the same k order, and the epilogue applies eval-mode BatchNorm as torch's CPU
// kernel does (y = x * alpha + beta, alpha = weight / sqrt(var + eps), beta = bias - mean * alpha), the residual
// and the ReLU, writing NHWC fp32 rows with 16-byte stores (swapped accumulator orientation, gemm_hl.h).
#include "gemm_hl5.h"

// out[row][col] = relu?( acc * alpha[col] + beta[col] (+ res[row][col]) )
// nn.ReLU / nn.MaxPool2d propagate a NaN (torch: `x < 0 ? 0 : x`, max with NaN = NaN); fmaxf() returns the non-NaN
// operand and would turn a poisoned activation into a finite, meaningless feature.  A NaN fails the comparison and is kept.
__device__ inline float relu_nan(float v) { return v < 0.f ? 0.f : v; }

struct EpiConvBN {
  const float* alpha; const float* beta; const float* res; float* out; int M, N, relu;
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      if (col >= N) continue;
      const float a = alpha[col], b = beta[col];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          const int row = m_wave + F::row(mi, i, lane);
          if (row < M) {
            float v = f.acc[mi][ni][i] * a + b;
            if (res) v += res[(size_t)row * N + col];
            out[(size_t)row * N + col] = relu ? relu_nan(v) : v;
          }
        }
    }
  }
  // swapped orientation: 8 consecutive columns of one row per (mi, p); N % 8 == 0
  static constexpr bool kSwap = true;
  template <class F>
  __device__ void swapped(const F& f, int m_wave, int n_wave, int lane) const {
    using S = gemm::SwapFrag;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int col = n_wave + S::col8(p, lane);
      if (col >= N) continue;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(alpha + col), a1 = *reinterpret_cast<const f32x4*>(alpha + col + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + col), b1 = *reinterpret_cast<const f32x4*>(beta + col + 4);
      f32x4 r0[4], r1[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {                 // all residual loads first, unconditionally (row clamped)
        r0[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        r1[mi] = r0[mi];
        if (res) {
          const float* s = res + (size_t)min(m_wave + S::row(mi, lane), M - 1) * N + col;
          r0[mi] = *reinterpret_cast<const f32x4*>(s);
          r1[mi] = *reinterpret_cast<const f32x4*>(s + 4);
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = m_wave + S::row(mi, lane);
        f32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o0[j] = f.acc[mi][2 * p][j] * a0[j] + b0[j];
          o1[j] = f.acc[mi][2 * p + 1][j] * a1[j] + b1[j];
          if (res) { o0[j] += r0[mi][j]; o1[j] += r1[mi][j]; }
          if (relu) { o0[j] = relu_nan(o0[j]); o1[j] = relu_nan(o1[j]); }
        }
        if (row < M) {
          float* d = out + (size_t)row * N + col;
          *reinterpret_cast<f32x4*>(d) = o0;
          *reinterpret_cast<f32x4*>(d + 4) = o1;
        }
      }
    }
  }
};

extern "C" int ac_conv_gemm_bn(const void* A_hl32, const void* W_hl32, int64_t M, int N, int K, const float* alpha,
                               const float* beta, const float* res, int relu, float* out, void* stream) {
  AC_REQUIRE(A_hl32 && W_hl32 && alpha && beta && out, "ac_conv_gemm_bn: null pointer");
  AC_REQUIRE(M > 0 && M <= 0x7fffff00LL && N > 0 && N % 8 == 0 && K > 0 && K % 32 == 0,
             "ac_conv_gemm_bn: bad shape M=%lld N=%d K=%d (N %% 8 == 0, K %% 32 == 0)", (long long)M, N, K);
  EpiConvBN epi{alpha, beta, res, out, (int)M, N, relu};
  const int policy = ac_gemm_policy();
  // the tile shape that finishes first in whole rounds of the chip (the rule of vit.hip's run_gemm)
  const long long cus = ac_device_cus();
  const long long tiles5 = ((M + 255) / 256) * ((N + 255) / 256), tiles2 = ((M + 255) / 256) * ((N + 127) / 128);
  const double cost5 = (double)((tiles5 + cus - 1) / cus), cost2 = 0.54 * (double)((tiles2 + cus - 1) / cus);
  const bool big = policy == AC_GEMM_256x256 || (policy != AC_GEMM_256x128 && cost5 <= cost2);
  if (big) return gemm5::launch<EpiConvBN>(A_hl32, W_hl32, (int)M, N, K, epi, as_stream(stream));
  return gemm2::launch<EpiConvBN, false>(A_hl32, W_hl32, (int)M, N, K, epi, as_stream(stream), 8, 0, 1);
}

// im2col into hl32.  x: fp32, element strides (sb, sc, sy, sx) - NHWC activations (sc == 1) or the NCHW input image.
// out row r = (b, oy, ox), column k = (ky * kw + kx) * C + c; columns [kh*kw*C, Kpad) are zero.  One thread per 8 k.
__global__ __launch_bounds__(256) void im2col_hl32_kernel(const float* __restrict__ x, int B, int C, int H, int W, int64_t sb,
                                                          int64_t sc, int64_t sy, int64_t sx, int kh, int kw, int stride, int pad,
                                                          int Ho, int Wo, int Kpad, uint8_t* __restrict__ out) {
  const int cpr = Kpad / 8;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)B * Ho * Wo * cpr;
  if (idx >= total) return;
  const int ch = (int)(idx % cpr);
  const size_t row = idx / cpr;
  const int k0 = ch * 8, K = kh * kw * C;
  const int b = (int)(row / ((size_t)Ho * Wo)), pp = (int)(row - (size_t)b * Ho * Wo), oy = pp / Wo, ox = pp - oy * Wo;
  float v[8];
  if (sc == 1 && (C & 7) == 0 && k0 < K) {              // 8 consecutive channels of one tap
    const int tap = k0 / C, c = k0 - tap * C, ky = tap / kw, kx = tap - ky * kw;
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const float* s = x + (size_t)b * sb + (size_t)iy * sy + (size_t)ix * sx + c;
      *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(s);
      *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(s + 4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float t = 0.f;
      if (k < K) {
        const int tap = k / C, c = k - tap * C, ky = tap / kw, kx = tap - ky * kw;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) t = x[(size_t)b * sb + (size_t)c * sc + (size_t)iy * sy + (size_t)ix * sx];
      }
      v[j] = t;
    }
  }
  store_hl32_8(out + row * (size_t)Kpad * 4, k0, v);
}

extern "C" int ac_im2col_hl32(const float* x, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int kh,
                              int kw, int stride, int pad, int Kpad, void* out_hl32, void* stream) {
  AC_REQUIRE(x && out_hl32, "ac_im2col_hl32: null pointer");
  AC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "ac_im2col_hl32: bad shape");
  AC_REQUIRE(Kpad % 32 == 0 && Kpad >= kh * kw * C, "ac_im2col_hl32: Kpad=%d must be a multiple of 32 and >= kh*kw*C=%d", Kpad,
             kh * kw * C);
  const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
  AC_REQUIRE(Ho > 0 && Wo > 0, "ac_im2col_hl32: empty output");
  AC_REQUIRE(sc != 1 || (C & 7) != 0 || ((((uintptr_t)x) & 15) == 0 && sb % 4 == 0 && sy % 4 == 0 && sx % 4 == 0),
             "ac_im2col_hl32: NHWC input must be 16-byte aligned");
  const size_t total = (size_t)B * Ho * Wo * (Kpad / 8);
  AC_REQUIRE((total + 255) / 256 <= 0x7fffffffULL, "ac_im2col_hl32: too large");
  hipLaunchKernelGGL(im2col_hl32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, B, C, H, W, sb,
                     sc, sy, sx, kh, kw, stride, pad, Ho, Wo, Kpad, static_cast<uint8_t*>(out_hl32));
  return ac_check_launch("ac_im2col_hl32");
}

// nn.MaxPool2d(k, stride, pad) on NHWC fp32 (padding never wins: windows are clipped to the image); C % 4 == 0.
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float* __restrict__ x, int B, int H, int W, int C, int k, int stride,
                                                           int pad, int Ho, int Wo, float* __restrict__ out) {
  const int c4 = C / 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * Ho * Wo * c4) return;
  const int c = (int)(idx % c4) * 4;
  const size_t pix = idx / c4;
  const int b = (int)(pix / ((size_t)Ho * Wo)), pp = (int)(pix - (size_t)b * Ho * Wo), oy = pp / Wo, ox = pp - oy * Wo;
  f32x4 m = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * stride - pad + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * stride - pad + kx;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = (v[j] > m[j] || v[j] != v[j]) ? v[j] : m[j];   // a NaN pixel wins, as in nn.MaxPool2d
    }
  }
  *reinterpret_cast<f32x4*>(out + pix * C + c) = m;
}

extern "C" int ac_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, float* out, void* stream) {
  AC_REQUIRE(x && out, "ac_maxpool_nhwc: null pointer");
  AC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k,
             "ac_maxpool_nhwc: bad shape (C %% 4 == 0, pad <= k/2)");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  AC_REQUIRE(Ho > 0 && Wo > 0, "ac_maxpool_nhwc: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x, B, H, W, C, k,
                     stride, pad, Ho, Wo, out);
  return ac_check_launch("ac_maxpool_nhwc");
}
