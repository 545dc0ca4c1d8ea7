// DINO ViT forward on gfx950: patch embedding, LayerNorm, fused-QKV attention and MLP.
// Reference arithmetic: models/patchcore/vision_transformer.py (PatchEmbed :128-131,
// prepare_tokens :196-207, Attention :80-92, Mlp :59-65, Block :107-113; eps 1e-6 from :253).
//
// Every dense contraction (patch projection, qkv, QK^T, PV, proj, fc1, fc2) runs on
// v_mfma_f32_16x16x32_f16 in the 3-pass split-fp16 scheme (gemm_hl.h), so the result is
// fp32-grade like the reference's fp32 CPU path.  The residual stream x stays fp32 in HBM;
// everything that feeds an MFMA (LN output, q/k/v, attention output, GELU output) is written
// directly in hl32 form by the producing kernel's epilogue - same bytes as fp32, no
// conversion pass.
#include "gemm_hl5.h"
#ifdef AC_EXPERIMENTAL
#include "gemm_hl1.h"
#include "gemm_hl3.h"
#include "gemm_hl4.h"
#endif

constexpr int HD = 64;          // head dim (all DINO ViTs)

// Linear layers: the 256x256 kernel (gemm_hl5.h), or - for small problems, e.g. a rank's shard of a category
// on 8 GPUs - the 256x128 kernel (gemm_hl2.h), whichever needs less time in whole rounds of the chip (below).
// The two kernels give bit-identical results (tests/test_gemm_generations_gpu.py); ac_set_gemm_policy() pins
// one of them for that test and for same-box A/B timing.
template <class Epi>
static int run_gemm(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream) {
  const int policy = ac_gemm_policy();
#ifdef AC_EXPERIMENTAL
  if (policy == 101) return gemm::launch(A, B, M, N, K, epi, stream);
  if (policy == 103) return gemm3::launch<Epi>(A, B, M, N, K, epi, stream);
  if (policy == 104) return gemm4::launch<Epi, 0>(A, B, M, N, K, epi, stream);
  if (policy == 105) return gemm4::launch<Epi, 1>(A, B, M, N, K, epi, stream);
  // 256x192 on TWELVE waves (3 per SIMD, 158 registers): the "third MFMA-issuing wave per SIMD" experiment of
  // round 2 - bit-identical, 2.5 % SLOWER than the 8-wave 256x256 kernel on a ViT block at 83 images
  if (policy == 107) return gemm5::launch<Epi, 4, false, 3>(A, B, M, N, K, epi, stream);
#endif
  // AUTO: the tile shape that finishes first on this chip.  One workgroup per CU at a time, so a launch takes
  // ceil(tiles / CUs) rounds; a 256x128 tile costs 0.54 of a 256x256 tile (half the work at ~8 % less efficiency:
  // DESIGN.md section 4).  At 83 images both rules pick 256x256 everywhere; at a rank's 11 or 21 images of an 8- or
  // 4-GPU run the half-size tiles fill the last round (qkv at 11 images: 306 big tiles = 2 rounds vs 612 small = 3
  // rounds x 0.54 = 1.6).
  const long long cus = ac_device_cus();
  const long long tiles5 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const long long tiles2 = (long long)((M + 255) / 256) * ((N + 127) / 128);
  const double cost5 = (double)((tiles5 + cus - 1) / cus), cost2 = 0.54 * (double)((tiles2 + cus - 1) / cus);
#ifdef AC_OLD_TILE_RULE          // (same-box A/B knob: round 2's rule - half-size tiles only when they fit one round)
  const bool big = policy == AC_GEMM_256x256 || (policy != AC_GEMM_256x128 && 2 * tiles5 > cus);
#else
  const bool big = policy == AC_GEMM_256x256 || (policy != AC_GEMM_256x128 && cost5 <= cost2);
#endif
  if (big) return gemm5::launch<Epi>(A, B, M, N, K, epi, stream);
  return gemm2::launch<Epi, false>(A, B, M, N, K, epi, stream, 8, 0, 1);
}

__host__ __device__ inline int tpad_of(int T) { return (T + 31) / 32 * 32; }

// ------------------------------------------------------------------ im2col
// images [B, Cin, S, S] -> hl32 [B*G*G, Cin*p*p], k = c*p*p + ky*p + kx (conv weight flattening).
// One thread per 8 consecutive k (8 | p, so the 8 pixels are contiguous in the image row).
__global__ __launch_bounds__(256) void im2col_hl_kernel(const float* __restrict__ img, int B, int Cin, int S, int p,
                                                        uint8_t* __restrict__ out) {
  const int G = S / p, Kd = Cin * p * p, cpr = Kd / 8;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)B * G * G * cpr;
  if (idx >= total) return;
  int ch = (int)(idx % cpr);
  size_t row = idx / cpr;
  int k = ch * 8;
  int c = k / (p * p), rem = k - c * p * p, ky = rem / p, kx = rem - ky * p;
  int b = (int)(row / (G * G)), pp = (int)(row - (size_t)b * G * G), py = pp / G, px = pp - py * G;
  const float* src = img + (((size_t)b * Cin + c) * S + (py * p + ky)) * S + px * p + kx;
  float v[8];
  *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
  *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
  store_hl32_8(out + row * Kd * 4, k, v);
}

__global__ void set_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, int B, int T, int C,
                               float* __restrict__ x) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  int b = i / C, c = i - b * C;
  x[(size_t)b * T * C + c] = cls[c] + pos[c];
}

// rows = patches (b*G*G + p), cols = embedding channel: x[b][1+p][col] = acc + bias + pos[1+p]
struct EpiPatchEmbed {
  const float* bias; const float* pos; float* x; int M, N, PP, T;
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      if (col >= N) continue;
      const float bc = bias[col];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          int row = m_wave + F::row(mi, i, lane);
          int rc = min(row, M - 1);
          int b = rc / PP, p = rc - b * PP;
          float pv = pos[(size_t)(1 + p) * N + col];          // unconditional load (see EpiResidual)
          if (row < M) x[((size_t)b * T + 1 + p) * N + col] = (f.acc[mi][ni][i] + bc) + pv;
        }
    }
  }
};

// --------------------------------------------------------------- LayerNorm
// One wave per row; the row lives in registers (C <= 2048), two-pass mean / variance in fp32.
template <bool HL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, uint8_t* __restrict__ out_hl, float* __restrict__ out_f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  const int nch = C / 8;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int ch = lane + 64 * t;
    if (ch < nch) {
      *reinterpret_cast<f32x4*>(v[t]) = *reinterpret_cast<const f32x4*>(xr + ch * 8);
      *reinterpret_cast<f32x4*>(v[t] + 4) = *reinterpret_cast<const f32x4*>(xr + ch * 8 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[t][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int ch = lane + 64 * t;
    if (ch < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = v[t][j] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int ch = lane + 64 * t;
    if (ch < nch) {
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (v[t][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j];
      if (HL) store_hl32_8(out_hl + (size_t)row * C * 4, ch * 8, y);
      else {
        *reinterpret_cast<f32x4*>(out_f + row * C + ch * 8) = *reinterpret_cast<f32x4*>(y);
        *reinterpret_cast<f32x4*>(out_f + row * C + ch * 8 + 4) = *reinterpret_cast<f32x4*>(y + 4);
      }
    }
  }
}

// ------------------------------------------------------------ GEMM epilogues
// helpers of the LDS-staged epilogues: 32 consecutive result columns of one row (+ bias) ...
__device__ inline void load_staged_row(const float* t, const float* bias32, float* v) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f32x4 a = *reinterpret_cast<const f32x4*>(t + 4 * c);
    f32x4 b = *reinterpret_cast<const f32x4*>(bias32 + 4 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * c + j] = a[j] + b[j];
  }
}
// ... written as one 128-byte hl32 group: 32 hi halves then 32 lo halves
__device__ inline void store_hl32_line(uint8_t* line, const float* v) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    half8 hi, lo;
    split8(v + 8 * c, hi, lo);
    *reinterpret_cast<half8*>(line + 16 * c) = hi;
    *reinterpret_cast<half8*>(line + 64 + 16 * c) = lo;
  }
}

// q, k, v: rows = tokens (b*T + t), cols in [0, 3C): dest hl32 [which][B*H][Tpad][64] (v in the same
// token-major layout as k; the attention kernel reads it through transposing LDS reads)
struct EpiQK {
  const float* bias; uint8_t* q; uint8_t* k; uint8_t* v; int M, C, T, Tpad, H;
  __device__ uint8_t* dest(int which) const { return which == 0 ? q : which == 1 ? k : v; }
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      if (col >= 3 * C) continue;
      const int which = col / C, fcol = col - which * C, head = fcol >> 6, d = fcol & 63;
      uint8_t* base = dest(which);
      const float bc = bias[col];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          int row = m_wave + F::row(mi, i, lane);
          if (row < M) {
            int b = row / T, t = row - b * T;
            float v = f.acc[mi][ni][i] + bc;
            _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            uint8_t* p = base + (((size_t)b * H + head) * Tpad + t) * (HD * 4) + (d >> 5) * 128 + (d & 31) * 2;
            *reinterpret_cast<_Float16*>(p) = hi;
            *reinterpret_cast<_Float16*>(p + 64) = lo;
          }
        }
    }
  }
  // LDS-staged form: a wave's 64 columns are exactly one head of q or k; lane -> (token row, half of
  // the head), one full 128-byte hl32 group per item
  __device__ void staged(const float* tl, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int item = pass * 64 + lane, rl = item >> 1, half = item & 1;
      const int row = m_wave + rl, col0 = n_wave + half * 32;
      if (row >= M || col0 >= 3 * C) continue;
      float v[32];
      load_staged_row(tl + rl * gemm::EPI_PITCH + half * 32, bias + col0, v);
      const int which = col0 / C, fcol = col0 - which * C, head = fcol >> 6, dh = (fcol & 63) >> 5;
      const int b = row / T, t = row - b * T;
      store_hl32_line(dest(which) + (((size_t)b * H + head) * Tpad + t) * (HD * 4) + dh * 128, v);
    }
  }
  // swapped orientation (gemm_hl2.h): 8 consecutive columns of one token row per (mi, p)
  static constexpr bool kSwap = true;
  template <class F>
  __device__ void swapped(const F& f, int m_wave, int n_wave, int lane) const {
    using S = gemm::SwapFrag;
    int tok_b[4], tok_t[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = min(m_wave + S::row(mi, lane), M - 1);
      tok_b[mi] = row / T;
      tok_t[mi] = row - tok_b[mi] * T;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int col = n_wave + S::col8(p, lane);
      if (col >= 3 * C) continue;
      const int which = col / C, fcol = col - which * C, head = fcol >> 6, d = fcol & 63;
      uint8_t* base = dest(which);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col), b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = f.acc[mi][2 * p][j] + b0[j]; v[4 + j] = f.acc[mi][2 * p + 1][j] + b1[j]; }
        if (m_wave + S::row(mi, lane) < M)
          store_hl32_8(base + (((size_t)tok_b[mi] * H + head) * Tpad + tok_t[mi]) * (HD * 4), d, v);
      }
    }
  }
};
template <> struct gemm::EpiTraits<EpiQK> { static constexpr bool kSymmetric = false; static constexpr bool kStaged = true; };

// out[m][col] = acc + bias[col] + res[m][col]   (fp32; out may alias res)
struct EpiResidual {
  const float* bias; const float* res; float* out; int M, N;
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      if (col >= N) continue;
      const float bc = bias[col];
      // all residual loads of this column first, unconditionally (row clamped): a load under
      // `if (row < M)` makes hipcc branch and wait vmcnt(0) per element - serial round trips
      float rv[F::NT][F::NR];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i)
          rv[mi][i] = res[(size_t)min(m_wave + F::row(mi, i, lane), M - 1) * N + col];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          int row = m_wave + F::row(mi, i, lane);
          if (row < M) out[(size_t)row * N + col] = rv[mi][i] + (f.acc[mi][ni][i] + bc);
        }
    }
  }
  // swapped orientation (gemm_hl2.h): 8 consecutive columns of one row per (mi, p), N % 8 == 0.
  // All residual loads first, unconditionally (row clamped), then the sums and the stores.
  static constexpr bool kSwap = true;
  template <class F>
  __device__ void swapped(const F& f, int m_wave, int n_wave, int lane) const {
    using S = gemm::SwapFrag;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int col = n_wave + S::col8(p, lane);
      if (col >= N) continue;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col), b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
      f32x4 r0[4], r1[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const float* s = res + (size_t)min(m_wave + S::row(mi, lane), M - 1) * N + col;
        r0[mi] = *reinterpret_cast<const f32x4*>(s);
        r1[mi] = *reinterpret_cast<const f32x4*>(s + 4);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = m_wave + S::row(mi, lane);
        f32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o0[j] = r0[mi][j] + (f.acc[mi][2 * p][j] + b0[j]);
          o1[j] = r1[mi][j] + (f.acc[mi][2 * p + 1][j] + b1[j]);
        }
        if (row < M) {
          float* d = out + (size_t)row * N + col;
          *reinterpret_cast<f32x4*>(d) = o0;
          *reinterpret_cast<f32x4*>(d + 4) = o1;
        }
      }
    }
  }
  // The wave's WHOLE accumulator (64 rows x NT*16 columns; gemm_hl5.h) in one call: the residual comes in groups of
  // 32 columns (8 x 16-byte loads per lane), and group q + 2 is requested before group q is summed and stored, so two
  // groups are always in flight.  Through swapped() the same tile takes four dependent round trips (loads -> wait ->
  // stores, four times: `out` may alias `res`, so the compiler may not move a load above an earlier store), each a
  // full memory latency while all 256 CUs burst their epilogues at once - ~13 us per tile more than an epilogue
  // without loads (round 3).  Each element is still loaded before it is stored, by the same
  // lane, so the in-place update x += f(x) stays correct.  Loads use clamped rows / columns; only stores are masked.
#ifndef AC_RES_NO_PIPELINE     // (same-box A/B knob of scratch/ab_vit.py; never defined in the product build)
  static constexpr bool kWholeTile = true;
#endif
  template <int NT>
  __device__ void swapped_tile(const f32x4 (&acc)[4][NT], int m_wave, int n_wave, int lane) const {
    using S = gemm::SwapFrag;
    constexpr int Q = NT / 2;
    if (m_wave >= M) return;                          // wave-uniform: nothing of this wave's 64 rows exists
    f32x4 r[Q][4][2], bv[Q][2];     // the bias vectors travel WITH their group: a load issued right before its use
    const float* rowp[4];           // would drain the whole queue (vmcnt counts in issue order)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) rowp[mi] = res + (size_t)min(m_wave + S::row(mi, lane), M - 1) * N;
    const int c0 = n_wave + 8 * (lane >> 4);
    auto load = [&](int q) {
      const int col = min(c0 + 32 * q, N - 8);
      bv[q][0] = *reinterpret_cast<const f32x4*>(bias + col);
      bv[q][1] = *reinterpret_cast<const f32x4*>(bias + col + 4);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        r[q][mi][0] = *reinterpret_cast<const f32x4*>(rowp[mi] + col);
        r[q][mi][1] = *reinterpret_cast<const f32x4*>(rowp[mi] + col + 4);
      }
    };
    // Stores go through a buffer resource that spans exactly this wave's valid rows ((M - m_wave) rows, at most 64):
    // a row beyond M is out of range and the hardware drops the store - no exec-mask branch, so the compiler counts
    // its vmcnt waits exactly instead of draining to 0 at every branch.  A column beyond N gets an out-of-range
    // offset the same way.
    const int rows_here = min(64, M - m_wave);        // >= 1 (checked above): the resource never spans a missing row
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m_wave * N, 0, rows_here * N * 4, 0x00020000);
    auto finish = [&](int q) {
      const int col = c0 + 32 * q;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        f32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o0[j] = r[q][mi][0][j] + (acc[mi][2 * q][j] + bv[q][0][j]);
          o1[j] = r[q][mi][1][j] + (acc[mi][2 * q + 1][j] + bv[q][1][j]);
        }
        const unsigned off = col < N ? (unsigned)(S::row(mi, lane) * N + col) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), orsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), orsrc, off + 16u, 0, 0);
      }
    };
    load(0);
    if constexpr (Q > 1) load(1);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q + 2 < Q) load(q + 2);
      finish(q);
    }
  }
};

// exact-erf GELU (nn.GELU default): 0.5 v (1 + erf(v / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. fp32 rounding level) - one v_rcp, one v_exp and a degree-5 Horner chain
// instead of libm's branchy erff, which made the fc1 epilogue cost as much as its K loop.
__device__ inline float gelu_erf(float v) {
  const float z = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  const float pe = poly * e;                       // = erfc(z): no cancellation on the negative side
  return 0.5f * v * (v >= 0.f ? 2.0f - pe : pe);
}

// Two values at a time: the fabs/Horner/product steps map to packed fp32 instructions (v_pk_fma_f32,
// v_pk_mul_f32 - two results per lane per issue), which is what the fc1 epilogue is bound by; same
// operations in the same order per component as gelu_erf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 gelu_erf2(f32x2 v) {
  const f32x2 z = __builtin_elementwise_abs(v) * 0.70710678118654752440f;
  const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f32x2 poly = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(poly, t, f32x2{0.254829592f, 0.254829592f});
  poly *= t;
  const f32x2 a = -z * z * 1.44269504088896340736f;
  const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32x2 pe = poly * e;
  const f32x2 hv = 0.5f * v;
  return f32x2{hv.x * (v.x >= 0.f ? 2.0f - pe.x : pe.x), hv.y * (v.y >= 0.f ? 2.0f - pe.y : pe.y)};
}

// g = gelu(acc + bias) (exact erf form, nn.GELU default) -> hl32 [M, N]
struct EpiGeluHl {
  const float* bias; uint8_t* out; int M, N;
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      if (col >= N) continue;
      const float bc = bias[col];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          int row = m_wave + F::row(mi, i, lane);
          if (row < M) {
            float v = gelu_erf(f.acc[mi][ni][i] + bc);
            _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            uint8_t* p = out + (size_t)row * N * 4 + (col >> 5) * 128 + (col & 31) * 2;
            *reinterpret_cast<_Float16*>(p) = hi;
            *reinterpret_cast<_Float16*>(p + 64) = lo;
          }
        }
    }
  }
  // LDS-staged form: lane -> (row, 32-column half); one full hl32 line (8 x 16-byte stores) per item
  __device__ void staged(const float* t, int m_wave, int n_wave, int lane) const {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int item = pass * 64 + lane, rl = item >> 1, half = item & 1;
      const int row = m_wave + rl, col0 = n_wave + half * 32;
      if (row >= M || col0 >= N) continue;
      float v[32];
      load_staged_row(t + rl * gemm::EPI_PITCH + half * 32, bias + col0, v);
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = gelu_erf(v[j]);
      store_hl32_line(out + (size_t)row * N * 4 + (col0 >> 5) * 128, v);
    }
  }
  // swapped orientation (gemm_hl2.h): 8 consecutive columns of one row per (mi, p), N % 8 == 0
  static constexpr bool kSwap = true;
  template <class F>
  __device__ void swapped(const F& f, int m_wave, int n_wave, int lane) const {
    using S = gemm::SwapFrag;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int col = n_wave + S::col8(p, lane);
      if (col >= N) continue;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col), b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row = m_wave + S::row(mi, lane);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const f32x2 g0 = gelu_erf2(f32x2{f.acc[mi][2 * p][j] + b0[j], f.acc[mi][2 * p][j + 1] + b0[j + 1]});
          const f32x2 g1 = gelu_erf2(f32x2{f.acc[mi][2 * p + 1][j] + b1[j], f.acc[mi][2 * p + 1][j + 1] + b1[j + 1]});
          v[j] = g0.x; v[j + 1] = g0.y;
          v[4 + j] = g1.x; v[5 + j] = g1.y;
        }
        if (row < M) store_hl32_8(out + (size_t)row * N * 4, col, v);
      }
    }
  }
};
template <> struct gemm::EpiTraits<EpiGeluHl> { static constexpr bool kSymmetric = false; static constexpr bool kStaged = true; };

// ----------------------------------------------------------------- attention
// softmax(q k^T * scale) v for one (image, head), flash style, split-fp16 MFMA for both products.
// Swapped orientation: S^T = K Q^T puts the KEYS on the accumulator registers and the QUERY on
// the lane, so the softmax max/sum over keys are in-lane reductions plus one lane-half exchange,
// and the exponentiated tile is already laid out as the B operand of O^T += V^T P^T
// (accumulator-as-operand, k order 16s + 8(j>>2) + 4h + (j&3); V^T fragments are read in that order).
// Block = 4 waves x 32 queries.  K and V tiles of 32 keys go global -> LDS by global_load_lds into a
// double buffer, one tile ahead of the MFMAs (no staging registers: a register double buffer cost a wave
// of occupancy and ran slower; without any prefetch the ~1.5 us load latency of every tile was exposed
// and the MFMA pipe sat at 31 %).
// Key order: MFMA row rho of the S^T tile holds key pi(rho) = rho with bits 2 and 3 swapped.  The
// accumulator-as-operand trick feeds P^T to the second product with k index j <-> accumulator row
// 16s + 8(j>>2) + 4h + (j&3); with pi that row is key 16s + 8h + j, i.e. eight CONSECUTIVE keys per lane.
// V stays token-major ([key][64 d] hl32 rows, written by the fused q/k/v GEMM like K - no transposed
// GEMM, no V^T buffer): the V^T fragment of O^T += V^T P^T comes from ds_read_b64_tr_b16, which hands
// lane i of a 16-lane group column i of a 4-key x 16-feature block, i.e. four consecutive keys of one
// feature; two of them give the lane its eight keys.
#ifndef AC_ATT_WAVES
#define AC_ATT_WAVES 4                     // waves (x 32 queries) per workgroup: 2 / 4 / 8 (same-box A/B builds; 4 ships)
#endif
constexpr int ATT_W = AC_ATT_WAVES, ATT_Q = 32 * ATT_W, ATT_KT = 32;
constexpr int ATT_TILE = 16384;            // K tile [32 keys][256 B] + V tile [32 keys][256 B]
constexpr int ATT_LDS = ATT_W * 32 * 65 * 4 > 2 * ATT_TILE ? ATT_W * 32 * 65 * 4 : 2 * ATT_TILE;   // epilogue transpose buffer / the two tile buffers
static_assert(ATT_W == 2 || ATT_W == 4 || ATT_W == 8, "8 DMA groups per tile are dealt to the waves");

__global__ __launch_bounds__(64 * ATT_W, 12 / ATT_W) void attention_kernel(const uint8_t* __restrict__ Q, const uint8_t* __restrict__ K,
                                                        const uint8_t* __restrict__ V, uint8_t* __restrict__ O,
                                                        int T, int Tpad, int H, float scale) {
  __shared__ __attribute__((aligned(256))) uint8_t lds[ATT_LDS];
  // per buffer: K tile, 16-byte slots XOR (row & 15); then V tile, 16-byte slots XOR ((row&3)<<2 | (row>>2)&3)
  // (conflict-free for the transposed reads)
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y, b = bh / H, head = bh - b * H;
  const int q0 = blockIdx.x * ATT_Q + wave * 32;
  const bool active = q0 < T;                          // wave-uniform
  const int C = H * HD;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // Q fragments (B operand): lane = query column, element j <-> feature 16s + 8h + j
  half8 qh[4], ql[4];
  {
    const uint8_t* qrow = Q + ((size_t)bh * Tpad + min(q0 + r, T - 1)) * (HD * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint8_t* p = qrow + (s >> 1) * 128 + (s & 1) * 32 + h * 16;
      qh[s] = *reinterpret_cast<const half8*>(p);
      ql[s] = *reinterpret_cast<const half8*>(p + 64);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -__builtin_inff(), l_run = 0.f;
  const float c_log2 = scale * 1.44269504088896340736f;

  const uint8_t* Kbase = K + (size_t)bh * Tpad * (HD * 4);
  const uint8_t* Vbase = V + (size_t)bh * Tpad * (HD * 4);
  const int ntiles = (T + ATT_KT - 1) / ATT_KT;
  // DMA map (1 KB per wave-instruction, lane l lands in 16-byte slot l of it; the swizzle goes on the
  // SOURCE address).  K and V tiles: instruction g = key rows 4g..4g+3.
  // Wave w issues g = w and w + 4 of each.
  constexpr int ATT_G = 8 / ATT_W;       // DMA groups (of K, and of V) per wave and tile
  const uint8_t* ksrc[ATT_G];
  const uint8_t* vsrc[ATT_G];
#pragma unroll
  for (int j = 0; j < ATT_G; ++j) {
    const int g = wave + ATT_W * j;
    const int krow = 4 * g + (lane >> 4), kslot = (lane & 15) ^ (krow & 15);
    ksrc[j] = Kbase + (size_t)krow * (HD * 4) + kslot * 16;
    const int vslot = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
    vsrc[j] = Vbase + (size_t)krow * (HD * 4) + vslot * 16;
  }
  // Keys >= T (the tail of the last tile): their scores are masked to -inf below, so p = 0 - but 0 * V must stay
  // 0, and the pad rows of the V buffer are never written.  Those lanes fetch row T-1 instead (any finite row
  // does), which replaces the pad-zeroing kernel of round 1.
  auto dma_tile = [&](int kt) {
    uint8_t* buf = lds + (kt & 1) * ATT_TILE;
    const bool last = kt == ntiles - 1;               // block-uniform
#pragma unroll
    for (int j = 0; j < ATT_G; ++j) {
      const int g = wave + ATT_W * j;
      const uint8_t* vs = vsrc[j] + (size_t)kt * (ATT_KT * HD * 4);
      if (last) {
        const int over = kt * ATT_KT + 4 * g + (lane >> 4) - (T - 1);
        if (over > 0) vs -= (size_t)over * (HD * 4);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)(ksrc[j] + (size_t)kt * (ATT_KT * HD * 4)), (lptr_t)(buf + g * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)vs, (lptr_t)(buf + 8192 + g * 1024), 16, 0, 0);
    }
  };
  const int pr = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);   // pi(r): key held by MFMA row r
  // Transposed V reads.  16-lane group G = lane >> 4 serves features 16 (G&1) .. +15 of the 32-feature
  // half mt and key half hk = G >> 1; lane 4q + p of the group addresses key row q of the 4-key block,
  // features 4p .. 4p+3 (8 bytes).  For key block j2 (keys 16s + 8hk + 4 j2 + q) of feature half mt, plane
  // hi/lo: byte = row*256 + ((ch ^ x) << 4) + 8 (p&1), ch = 8 mt + 4 plane + 2 (G&1) + (p>>1),
  // x = (q << 2) | ((2 hk + j2) & 3).  mt and plane only flip address bits 7 and 6, s adds 4096.
  const int trq = (lane & 15) >> 2, trp = lane & 3, trG = lane >> 4, trh = trG >> 1;
  unsigned vaddr[2];
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2) {
    const int row = 8 * trh + 4 * j2 + trq, ch = 2 * (trG & 1) + (trp >> 1), x = (trq << 2) | ((2 * trh + j2) & 3);
    vaddr[j2] = (unsigned)(8192 + row * 256 + ((ch ^ x) << 4) + 8 * (trp & 1));
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  dma_tile(0);
  // One key tile.  MASKED (a compile-time tag): only the last tile can hold keys >= T.  As a run-time condition inside
  // one loop body hipcc turned the block-uniform branch into selects - 16 v_cmp + 16 v_cndmask + the key indices,
  // ~58 of ~330 instructions of EVERY tile, in a loop whose VALU issue is as busy as its MFMA pipe - so the last tile
  // is peeled into its own instantiation.
  auto key_tile = [&](const int kt, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int key0 = kt * ATT_KT;
    const uint8_t* Ks = lds + (kt & 1) * ATT_TILE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own share of tile kt has landed
    __syncthreads();                                   // ... everybody's has; tile kt-1 is fully consumed
#ifndef AC_ATT_ABLATE            // timing-only diagnostic builds (results wrong): 1 no K fragment reads after the first tile,
#define AC_ATT_ABLATE 0         // 2 no transposed V reads after the first tile, 4 no K / V DMA after the first tile.  0 ships.
#endif
    if (kt + 1 < ntiles && !((AC_ATT_ABLATE & 4) && kt > 0)) dma_tile(kt + 1);             // flies under this tile's MFMAs
    // A wave whose 32 queries all lie beyond the sequence (T = 785: the last block of 128 queries holds 17) only helps
    // with the DMA and the barriers: its SIMD slots go to the other workgroups of the CU (3 of the 28 wave slots per
    // head did redundant work on clamped rows: -10 % attention time at T = 785).
#ifndef AC_ATT_NO_SKIP        // (same-box A/B knob of scratch/ab_vit.py; never defined in the product build)
    if (!active) return;
#endif

    // V^T fragments: two transposed 4-key blocks per (feature half, key group, plane), issued now so that
    // their latency hides under the S^T MFMAs and the softmax
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    static_assert(AC_ATT_ABLATE == 0 || true, "");
    u32x2_t vr[2][2][2][2];              // [mt][s][plane][j2]
    if ((AC_ATT_ABLATE & 2) && kt > 0) {
#pragma unroll
      for (int a = 0; a < 16; ++a) vr[a >> 3][(a >> 2) & 1][(a >> 1) & 1][a & 1] = u32x2_t{(unsigned)(0x3c003c00u + a + kt), 0x3c003c00u};
    } else {
      const unsigned vb0 = lds_base + (unsigned)((kt & 1) * ATT_TILE) + vaddr[0];
      const unsigned vb1 = lds_base + (unsigned)((kt & 1) * ATT_TILE) + vaddr[1];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pn = 0; pn < 2; ++pn) {
          const unsigned a0 = vb0 ^ (unsigned)(mt * 128 + pn * 64), a1 = vb1 ^ (unsigned)(mt * 128 + pn * 64);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vr[mt][0][pn][0]) : "v"(a0) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vr[mt][0][pn][1]) : "v"(a1) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vr[mt][1][pn][0]) : "v"(a0) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(vr[mt][1][pn][1]) : "v"(a1) : "memory");
        }
    }
    // S^T tile [32 keys x 32 queries]
    f32x16 s_acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) s_acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int slot = (s >> 1) * 8 + (s & 1) * 2 + h;
      half8 kh, kl;
      if ((AC_ATT_ABLATE & 1) && kt > 0) {
        kh = qh[s]; kl = ql[s];
      } else {
        kh = *reinterpret_cast<const half8*>(Ks + pr * 256 + ((slot ^ (pr & 15)) << 4));
        kl = *reinterpret_cast<const half8*>(Ks + pr * 256 + (((slot + 4) ^ (pr & 15)) << 4));
      }
      s_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], s_acc, 0, 0, 0);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], s_acc, 0, 0, 0);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], s_acc, 0, 0, 0);
    }
    // online softmax over the key axis (registers + the other lane half), in the log2 domain:
    // p = 2^(s*c - m) with c = scale*log2(e) - one FMA and one v_exp_f32 per score.  Only the last
    // tile has padded keys (block-uniform branch); the O / l rescale is skipped (wave-uniform
    // branch) whenever no lane's running maximum moved.  Accumulator element i of lane half h is
    // key key0 + 16(i>>3) + 8h + (i&7) (see pi above).
    if constexpr (MASKED) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int key = key0 + 16 * (i >> 3) + 8 * h + (i & 7);
        s_acc[i] = key < T ? s_acc[i] : -__builtin_inff();
      }
    }
    float mloc = s_acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s_acc[i]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc * c_log2);
    if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {
      const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= corr;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= corr; o[1][i] *= corr; }
      m_run = m_new;
    }
    float psum = 0.f;
    half8 ph[2], pl[2];
    {
      u32x4 phw[2], plw[2];                // the hi / lo split of P two scores at a time (common.h split2: 1.5 VALU per score)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s_acc[i], c_log2, -m_new));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s_acc[i + 1], c_log2, -m_new));
        psum += p0;
        psum += p1;
        uint32_t a, b;
        split2(p0, p1, a, b);
        phw[i >> 3][(i & 7) >> 1] = a;
        plw[i >> 3][(i & 7) >> 1] = b;
      }
      ph[0] = __builtin_bit_cast(half8, phw[0]); ph[1] = __builtin_bit_cast(half8, phw[1]);
      pl[0] = __builtin_bit_cast(half8, plw[0]); pl[1] = __builtin_bit_cast(half8, plw[1]);
    }
    l_run += psum;
    // O^T += V^T P^T ; A fragment element j <-> key 16s + 8h + j (the transposed reads were issued above)
    {
      // The reads are asynchronous and the compiler does not know it: the wait must be a DEPENDENCY of
      // every use of their results, or copies of the destination registers may be scheduled above it.
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vr[0][0][0][0]), "+v"(vr[0][0][0][1]), "+v"(vr[0][0][1][0]), "+v"(vr[0][0][1][1]),
                     "+v"(vr[0][1][0][0]), "+v"(vr[0][1][0][1]), "+v"(vr[0][1][1][0]), "+v"(vr[0][1][1][1]),
                     "+v"(vr[1][0][0][0]), "+v"(vr[1][0][0][1]), "+v"(vr[1][0][1][0]), "+v"(vr[1][0][1][1]),
                     "+v"(vr[1][1][0][0]), "+v"(vr[1][1][0][1]), "+v"(vr[1][1][1][0]), "+v"(vr[1][1][1][1])
                   :: "memory");
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          u32x4 th = {vr[mt][s][0][0][0], vr[mt][s][0][0][1], vr[mt][s][0][1][0], vr[mt][s][0][1][1]};
          u32x4 tl = {vr[mt][s][1][0][0], vr[mt][s][1][0][1], vr[mt][s][1][1][0], vr[mt][s][1][1][1]};
          const half8 vh = __builtin_bit_cast(half8, th), vl = __builtin_bit_cast(half8, tl);
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o[mt], 0, 0, 0);
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o[mt], 0, 0, 0);
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o[mt], 0, 0, 0);
        }
    }
  };
  for (int kt = 0; kt + 1 < ntiles; ++kt) key_tile(kt, std::false_type{});
  if (ntiles * ATT_KT > T) key_tile(ntiles - 1, std::true_type{});
  else key_tile(ntiles - 1, std::false_type{});
  // normalise, transpose through LDS ([32 q][65] floats per wave), write hl32 lines
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  __syncthreads();
  float* tb = reinterpret_cast<float*>(lds) + wave * 32 * 65;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) tb[r * 65 + mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = o[mt][i] * inv;
  __builtin_amdgcn_s_waitcnt(0);   // LDS writes of this wave are visible to itself after lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  const int q = lane >> 1, half = lane & 1, tq = q0 + q;
  if (tq < T) {
    const float* src = tb + q * 65 + half * 32;
    uint8_t* dst = O + ((size_t)b * T + tq) * ((size_t)C * 4) + (head * 2 + half) * 128;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[c8 * 8 + j];
      half8 hi, lo;
      split8(v, hi, lo);
      *reinterpret_cast<half8*>(dst + c8 * 16) = hi;
      *reinterpret_cast<half8*>(dst + 64 + c8 * 16) = lo;
    }
  }
}

struct VitWs {
  uint8_t *h, *q, *k, *v, *o, *g, *cols;
};

static size_t vit_ws_carve(Arena& ar, int B, int T, int C, int hidden, int heads, int patch_dim, VitWs* w) {
  const size_t M = (size_t)B * T, Tp = tpad_of(T);
  VitWs t;
  t.h = ar.take<uint8_t>(M * C * 4);
  t.q = ar.take<uint8_t>((size_t)B * heads * Tp * HD * 4);
  t.k = ar.take<uint8_t>((size_t)B * heads * Tp * HD * 4);
  t.v = ar.take<uint8_t>((size_t)B * heads * Tp * HD * 4);
  t.o = ar.take<uint8_t>(M * C * 4);
  t.g = ar.take<uint8_t>(M * (size_t)hidden * 4);
  t.cols = ar.take<uint8_t>(M * (size_t)patch_dim * 4);
  if (w) *w = t;
  return ar.off;
}

extern "C" size_t ac_vit_workspace_bytes(int B, int T, int C, int hidden, int heads, int patch_dim) {
  if (B <= 0 || T <= 0 || C <= 0 || hidden <= 0 || heads <= 0 || patch_dim <= 0) return 256;
  Arena ar(nullptr, ~(size_t)0 >> 1);
  return vit_ws_carve(ar, B, T, C, hidden, heads, patch_dim, nullptr) + 256;
}

extern "C" int ac_vit_prepare_tokens(const float* images, int B, int in_ch, int img, int patch, int C,
                                     const void* proj_w, const float* proj_b, const float* cls, const float* pos,
                                     float* x, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = as_stream(stream_);
  AC_REQUIRE(images && proj_w && proj_b && cls && pos && x && ws, "ac_vit_prepare_tokens: null pointer");
  AC_REQUIRE(B > 0 && in_ch > 0 && patch > 0 && img % patch == 0, "ac_vit_prepare_tokens: bad geometry");
  AC_REQUIRE(patch % 8 == 0, "ac_vit_prepare_tokens: patch=%d must be a multiple of 8", patch);
  const int G = img / patch, PP = G * G, T = PP + 1, Kd = in_ch * patch * patch;
  AC_REQUIRE(Kd % 32 == 0, "ac_vit_prepare_tokens: in_ch*patch^2=%d must be a multiple of 32", Kd);
  size_t need = align_up((size_t)B * PP * Kd * 4, 256);
  if (ws_bytes < need) return ac_fail(AC_EWORKSPACE, "ac_vit_prepare_tokens: workspace %zu < %zu", ws_bytes, need);
  uint8_t* cols = static_cast<uint8_t*>(ws);
  size_t chunks = (size_t)B * PP * (Kd / 8);
  hipLaunchKernelGGL(im2col_hl_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, images, B, in_ch,
                     img, patch, cols);
  hipLaunchKernelGGL(set_cls_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, cls, pos, B, T, C, x);
  EpiPatchEmbed epi{proj_b, pos, x, B * PP, C, PP, T};
  return run_gemm(cols, proj_w, B * PP, C, Kd, epi, stream);
}

extern "C" int ac_layernorm_rows(const float* x, int64_t rows, int C, const float* w, const float* b, float eps,
                                 float* y, void* stream) {
  AC_REQUIRE(x && w && b && y, "ac_layernorm_rows: null pointer");
  AC_REQUIRE(C > 0 && C % 8 == 0 && C <= 2048, "ac_layernorm_rows: C=%d must be a multiple of 8, <= 2048", C);
  if (rows <= 0) return AC_OK;
  hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, rows,
                     C, w, b, eps, (uint8_t*)nullptr, y);
  return ac_check_launch("layernorm_rows");
}

extern "C" int ac_vit_block(const ac_block_weights* w, const float* x_in, float* x_out, int B, int T, int C, int heads,
                            int hidden, float eps, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = as_stream(stream_);
  AC_REQUIRE(w && x_in && x_out && ws, "ac_vit_block: null pointer");
  AC_REQUIRE(B > 0 && T > 0, "ac_vit_block: empty batch");
  AC_REQUIRE(heads > 0 && C == heads * HD, "ac_vit_block: C=%d must equal heads*64 (heads=%d)", C, heads);
  AC_REQUIRE(C % 32 == 0 && hidden % 32 == 0 && C <= 2048, "ac_vit_block: C, hidden must be multiples of 32; C <= 2048");
  AC_REQUIRE((long long)B * T < (1LL << 30), "ac_vit_block: too many tokens");
  {
    // the epilogues read biases / residual rows and write results with 16-byte accesses
    const void* ptrs[] = {x_in, x_out, w->ln1_w, w->ln1_b, w->qkv_w, w->qkv_b, w->proj_w, w->proj_b,
                          w->ln2_w, w->ln2_b, w->fc1_w, w->fc1_b, w->fc2_w, w->fc2_b, ws};
    for (const void* p : ptrs)
      AC_REQUIRE(p && ((uintptr_t)p & 15) == 0, "ac_vit_block: every pointer must be non-null and 16-byte aligned");
  }
  if (ws_bytes < ac_vit_workspace_bytes(B, T, C, hidden, heads, 32))
    return ac_fail(AC_EWORKSPACE, "ac_vit_block: workspace %zu < %zu", ws_bytes,
                   ac_vit_workspace_bytes(B, T, C, hidden, heads, 32));
  Arena ar(ws, ws_bytes);
  VitWs b;
  vit_ws_carve(ar, B, T, C, hidden, heads, 32, &b);
  if (!ar.ok()) return ac_fail(AC_EWORKSPACE, "ac_vit_block: workspace carve failed");
  const int M = B * T, Tp = tpad_of(T);
  const unsigned ln_grid = (unsigned)((M + 3) / 4);
  int rc;

  // x_mid = x_in + proj(attn(ln1(x_in)))
  hipLaunchKernelGGL(layernorm_kernel<true>, dim3(ln_grid), dim3(256), 0, stream, x_in, (int64_t)M, C, w->ln1_w, w->ln1_b,
                     eps, b.h, (float*)nullptr);
  const uint8_t* wqkv = static_cast<const uint8_t*>(w->qkv_w);
  EpiQK eqkv{w->qkv_b, b.q, b.k, b.v, M, C, T, Tp, heads};
  if ((rc = run_gemm(b.h, wqkv, M, 3 * C, C, eqkv, stream))) return rc;
  hipLaunchKernelGGL(attention_kernel, dim3((T + ATT_Q - 1) / ATT_Q, B * heads), dim3(64 * ATT_W), 0, stream, b.q, b.k, b.v,
                     b.o, T, Tp, heads, 1.0f / sqrtf((float)HD));
  EpiResidual ep{w->proj_b, x_in, x_out, M, C};
  if ((rc = run_gemm(b.o, w->proj_w, M, C, C, ep, stream))) return rc;

  // x_out = x_mid + fc2(gelu(fc1(ln2(x_mid))))
  hipLaunchKernelGGL(layernorm_kernel<true>, dim3(ln_grid), dim3(256), 0, stream, x_out, (int64_t)M, C, w->ln2_w,
                     w->ln2_b, eps, b.h, (float*)nullptr);
  EpiGeluHl eg{w->fc1_b, b.g, M, hidden};
  if ((rc = run_gemm(b.h, w->fc1_w, M, hidden, C, eg, stream))) return rc;
  EpiResidual e2{w->fc2_b, x_out, x_out, M, C};
  if ((rc = run_gemm(b.g, w->fc2_w, M, C, hidden, e2, stream))) return rc;
  return ac_check_launch("ac_vit_block");
}

#ifdef AC_TILE_TIMING
// Diagnostic builds only (see gemm_hl2.h): reset != 0 restarts the launch-slot counter; otherwise copies
// the [8 slots][8192 blocks][4 stamps] table to `out` and returns the number of GEMM launches since reset.
extern "C" int ac_debug_tile_times(unsigned long long* out, int reset) {
  if (reset) { gemm2::tt_counter() = 0; return 0; }
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm2::ac_tile_times), sizeof(gemm2::ac_tile_times)) != hipSuccess) return -1;
  return gemm2::tt_counter();
}
#endif
