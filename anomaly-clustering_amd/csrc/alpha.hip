// alpha-weight path on gfx950: all-pairs patch distance with fused min-reduce (MFMA),
// fp64 softmax, weighted-average embedding, condensed pdist.
// Reference arithmetic: models/patchcore/utils.py:222-277, examples/main.py:294-296,
// scipy pdist under examples/test.py:193-195.
#include "gemm_hl5.h"
#include "gemm_hl7.h"
#ifdef AC_EXPERIMENTAL
#include "gemm_hl1.h"
#include "gemm_hl3.h"
#include "gemm_hl4.h"
#include "gemm_hl6.h"
#endif

// ------------------------------------------------------------------ packing
// fp32 [rows, K] -> hl32 + optional squared row norms.  One wave per row; each lane
// converts 8 consecutive floats per trip (two 16-byte loads in, hi/lo 16-byte stores out).
__global__ __launch_bounds__(256) void pack_hl32_kernel(const float* __restrict__ src, int64_t rows, int K,
                                                        int64_t ld, uint8_t* __restrict__ dst,
                                                        float* __restrict__ sqnorm) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = src + row * ld;
  uint8_t* d = dst + (size_t)row * K * 4;
  double acc = 0.0;   // fp64 sum of squares, rounded once: the norm enters d2 at ~5x its magnitude
  for (int k = lane * 8; k < K; k += 512) {
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(s + k);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(s + k + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (double)v[j] * (double)v[j];
    store_hl32_8(d, k, v);
  }
  if (sqnorm) {
    acc = wave_sum(acc);
    if (lane == 0) sqnorm[row] = (float)acc;
  }
}

extern "C" int ac_pack_hl32(const float* src, int64_t rows, int K, int64_t src_ld, void* dst, float* row_sqnorm,
                            void* stream) {
  AC_REQUIRE(src && dst, "ac_pack_hl32: null pointer");
  AC_REQUIRE(K > 0 && K % 32 == 0, "ac_pack_hl32: K=%d must be a positive multiple of 32", K);
  AC_REQUIRE(src_ld >= K && src_ld % 4 == 0, "ac_pack_hl32: src_ld=%lld must be >= K and a multiple of 4",
             (long long)src_ld);
  AC_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "ac_pack_hl32: pointers must be 16-byte aligned");
  if (rows <= 0) return AC_OK;
  int64_t blocks = (rows + 3) / 4;
  AC_REQUIRE(blocks <= 0x7fffffff, "ac_pack_hl32: too many rows");
  hipLaunchKernelGGL(pack_hl32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, rows, K,
                     src_ld, static_cast<uint8_t*>(dst), row_sqnorm);
  return ac_check_launch("pack_hl32");
}

// ------------------------------------------------- pair-min-distance epilogue
// GEMM orientation: A (rows -> accumulator registers) = KEY patches, flat over all key
// images; B (columns -> lanes) = QUERY patches.  d2(key, q) = |key|^2 + |q|^2 - 2 key.q,
// clamped at 0 (torch.cdist's clamp_min, utils.py:226 via torch).  The min over the key
// patches of ONE key image is an in-lane min over registers, one lane-half exchange, and one
// atomicMin per (key image, query column) into minbuf[key_image][query] (d2 >= 0, so the
// IEEE bit pattern orders like an unsigned integer).
struct EpiPairMin {
  using Frag16 = gemm::FragT<16>;
  const float* key_sq;   // [M]  squared norms of the A rows
  const float* qry_sq;   // [N]  squared norms of the B rows
  unsigned* minbuf;      // [images of A][N]: min over the A patches of one image, per B patch
  unsigned* minbuf_t;    // [images of B][M]: the other direction (symmetric: == minbuf; two-way: its own matrix)
  int M, N, P;
  // upper_only (diagonal blocks of a symmetric problem): elements with row > column are left out - the
  // mirror element, formed with the lower row as the A operand, stands for both.
  // n_lim (gemm_hl7.h: a 64 x 32 piece of a 64 x 96 wave block padded to a fragment): columns >= n_lim are not this
  // fragment's and are left alone.
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane, bool upper_only = false, int n_lim = 0x7fffffff) const {
    if (m_wave >= M) return;                      // wave-uniform
    const int m_end = min(m_wave + 64, M), n_end = min(N, n_lim);
    float ksq[F::NT][F::NR];
#pragma unroll
    for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
      for (int i = 0; i < F::NR; ++i) ksq[mi][i] = key_sq[min(m_wave + F::row(mi, i, lane), M - 1)];
    const int img_first = m_wave / P, img_last = (m_end - 1) / P;
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      const float qs = qry_sq[min(col, N - 1)];
      float d2[F::NT][F::NR];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {  // (|k|^2 - k.q) + (|q|^2 - k.q): both differences are small before they meet
          const float v = fmaxf((ksq[mi][i] - f.acc[mi][ni][i]) + (qs - f.acc[mi][ni][i]), 0.f);
          d2[mi][i] = (upper_only && m_wave + F::row(mi, i, lane) > col) ? __builtin_inff() : v;
        }
      for (int img = img_first; img <= img_last; ++img) {   // wave-uniform trip count (<= 2 when P >= 64)
        const int lo = img * P, hi = min(lo + P, m_end);
        float mn = __builtin_inff();
#pragma unroll
        for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
          for (int i = 0; i < F::NR; ++i) {
            int row = m_wave + F::row(mi, i, lane);
            mn = (row >= lo && row < hi) ? fminf(mn, d2[mi][i]) : mn;
          }
        mn = F::min_over_row_groups(mn);
        if (F::first_row_group(lane) && col < n_end) atomicMin(minbuf + (size_t)img * N + col, __float_as_uint(mn));
      }
    }
  }

  // The same tile read the other way - the block's ROWS are the query patches, its COLUMNS the key
  // patches (symmetric case: keys == queries, M == N; two-way case: a rectangle between two row sets).  One TW-column slab of the wave's d2 at a time
  // goes through LDS transposed ([column][row], 65-float pitch), then lane l owns row l and takes the
  // min over the columns of each key image.
  template <class F>
  __device__ void transposed(const F& f, int m_wave, int n_wave, int lane, float* tb, bool upper_only = false,
                             int n_lim = 0x7fffffff) const {
    const int n_end = min(N, n_lim);
    if (m_wave >= M || n_wave >= n_end) return;   // wave-uniform
    float ksq[F::NT][F::NR];
#pragma unroll
    for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
      for (int i = 0; i < F::NR; ++i) ksq[mi][i] = key_sq[min(m_wave + F::row(mi, i, lane), M - 1)];
    const int my_row = m_wave + lane;
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int c_base = n_wave + ni * F::TW;
      if (c_base >= n_end) break;                  // wave-uniform
      const int cl = F::col(ni, lane) - ni * F::TW;
      const float qs = qry_sq[min(c_base + cl, N - 1)];
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          const float v = fmaxf((ksq[mi][i] - f.acc[mi][ni][i]) + (qs - f.acc[mi][ni][i]), 0.f);
          tb[cl * 65 + F::row(mi, i, lane)] =
              (upper_only && m_wave + F::row(mi, i, lane) > c_base + cl) ? __builtin_inff() : v;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int c_end = min(c_base + F::TW, n_end);
      for (int img = c_base / P; img <= (c_end - 1) / P; ++img) {   // wave-uniform, <= 2 trips when P >= 32
        const int lo = max(img * P, c_base) - c_base, hi = min((img + 1) * P, c_end) - c_base;
        float mn = __builtin_inff();
        for (int c = lo; c < hi; ++c) mn = fminf(mn, tb[c * 65 + lane]);
        if (my_row < M) atomicMin(minbuf_t + (size_t)img * M + my_row, __float_as_uint(mn));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- fast path of both reductions (round 4).  The general code above spends ~5 000 instructions per wave on
  // per-element row / column range tests, a second evaluation of d2 and an LDS transpose - 14.5 us per tile during
  // which the CU's MFMA pipe idles (one workgroup per CU: nothing else is resident), 9 % of the launch
  // (profiles/r04_allpairs_tile_timing.txt).  A wave block that lies wholly inside the problem, off the diagonal,
  // with at most one image boundary per side (P >= 64) needs none of that:
  //   * the norms of the wave's 64 rows arrive as four 16-byte loads (rows 16 mi + 4 g + 0..3 are consecutive),
  //   * d2 is formed ONCE, in place of the accumulator (same expression, same bits),
  //   * direction 1 (min over the key rows of an image, per query column): in-lane min over the lane's 16 rows,
  //     two lane exchanges across the four row groups, then lane (g, c) keeps the column 16 g + c: ONE atomicMin
  //     wave-instruction on 64 consecutive words per image,
  //   * direction 2 (min over the key COLUMNS of an image, per row): in-lane min over the lane's 4 column tiles, an
  //     all-reduce over the 16 lanes of a row group by four DPP row rotations (no LDS), then lane (g, c) keeps the
  //     row 16 (c >> 2) + 4 g + (c & 3): ONE atomicMin wave-instruction on the wave's 64 consecutive rows per image.
  // The minima are exact (min is order-free), so the result equals the general path's bit for bit.
  // Returns false (wave-uniform) when the block does not qualify; the caller then runs the general code.
  template <int N_ROR>
  static __device__ inline float row_ror(float v) {      // DPP row_ror:N_ROR inside each group of 16 lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N_ROR, 0xf, 0xf, false));
  }
  template <class F>
  __device__ bool fast(F& f, int m_wave, int n_wave, int lane, bool both_ways) const {
    static_assert(F::TW == 16 && F::NT == 4 && F::NR == 4, "written for the 16x16x32 accumulator layout");
    return fast_wide<4>(f.acc, m_wave, n_wave, lane, both_ways);
  }
  // Block-uniform: every 64 x W wave block of the bm x bn tile at (m0, n0) takes fast_wide with both_ways as the kernel
  // passes it - i.e. no wave of that tile needs LDS in its epilogue (gemm_hl7.h issues the next tile's loads early then).
  __device__ bool tile_is_fast(int m0, int n0, int bm, int bn, int W) const {
#ifdef AC_NO_FAST_EPI
    return false;
#endif
    return m0 + bm <= M && n0 + bn <= N && P >= W && !((reinterpret_cast<uintptr_t>(key_sq) | reinterpret_cast<uintptr_t>(qry_sq)) & 15) &&
           (!symmetric || m0 + bm - 1 <= n0);
  }
  // NTN column tiles of 16: the 64 x 64 wave block of gemm_hl2.h (NTN = 4) and the 64 x 96 block of gemm_hl7.h (6).
  template <int NTN>
  __device__ bool fast_wide(f32x4 (&acc)[4][NTN], int m_wave, int n_wave, int lane, bool both_ways) const {
    if (!fast_block<NTN>(m_wave, n_wave)) return false;
    f32x4 ks[4];
    float qs[NTN];
    fast_load<NTN>(m_wave, n_wave, lane, ks, qs);
    fast_reduce<NTN>(acc, ks, qs, m_wave, n_wave, lane, both_ways);
    return true;
  }
  // The three parts of fast_wide, for a caller that wants the norm loads in flight before other memory traffic of its
  // own (gemm_hl7.h issues the next tile's first stages between fast_load and fast_reduce).
  template <int NTN>
  __device__ bool fast_block(int m_wave, int n_wave) const {            // wave-uniform: does this block qualify?
    static_assert(NTN >= 4 && NTN <= 8, "direction 1 keeps its columns in two atomicMin wave-instructions");
#ifdef AC_NO_FAST_EPI                        // diagnostic builds: the general epilogue everywhere (same-box A/B)
    return false;
#endif
    return !(m_wave + 64 > M || n_wave + 16 * NTN > N || P < 16 * NTN || (reinterpret_cast<uintptr_t>(key_sq) & 15));
  }
  template <int NTN>
  __device__ void fast_load(int m_wave, int n_wave, int lane, f32x4 (&ks)[4], float (&qs)[NTN]) const {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) ks[mi] = *reinterpret_cast<const f32x4*>(key_sq + m_wave + 16 * mi + 4 * g);
#pragma unroll
    for (int ni = 0; ni < NTN; ++ni) qs[ni] = qry_sq[n_wave + 16 * ni + c];
  }
  // returns the number of atomicMin wave-instructions it issued (wave-uniform; 2 ... 6)
  template <int NTN>
  __device__ int fast_reduce(f32x4 (&acc)[4][NTN], const f32x4 (&ks)[4], const float (&qs)[NTN], int m_wave, int n_wave,
                             int lane, bool both_ways) const {
    constexpr int W = 16 * NTN;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTN; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i)      // (|k|^2 - k.q) + (|q|^2 - k.q): both differences are small before they meet
          acc[mi][ni][i] = fmaxf((ks[mi][i] - acc[mi][ni][i]) + (qs[ni] - acc[mi][ni][i]), 0.f);
    const float inf = __builtin_inff();

    // ---- direction 1: rows are key patches of image r_img (and r_img + 1 from local row br on)
    const int r_img = m_wave / P;
    const int br = (r_img + 1) * P - m_wave;            // first local row of the next image; >= 64: no boundary inside
    {
      float mn0[NTN], mn1[NTN];
      if (br >= 64) {                                   // wave-uniform
#pragma unroll
        for (int ni = 0; ni < NTN; ++ni) {
          float m = inf;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int i = 0; i < 4; ++i) m = fminf(m, acc[mi][ni][i]);
          mn0[ni] = m;
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NTN; ++ni) { mn0[ni] = inf; mn1[ni] = inf; }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool first = 16 * mi + 4 * g + i < br;
#pragma unroll
            for (int ni = 0; ni < NTN; ++ni) {
              const float v = acc[mi][ni][i];
              mn0[ni] = fminf(mn0[ni], first ? v : inf);
              mn1[ni] = fminf(mn1[ni], first ? inf : v);
            }
          }
      }
      // lane (g, c) owns column 16 g + c = lane of the first 64 columns, and column 64 + lane of the rest (lanes < W - 64)
      float v0 = inf, v1 = inf, u0 = inf, u1 = inf;
#pragma unroll
      for (int ni = 0; ni < NTN; ++ni) {
        float m = Frag16::min_over_row_groups(mn0[ni]);      // valid in every lane of the column
        if (ni < 4) v0 = (g == ni) ? m : v0; else u0 = (g == ni - 4) ? m : u0;
        if (br < 64) {
          m = Frag16::min_over_row_groups(mn1[ni]);
          if (ni < 4) v1 = (g == ni) ? m : v1; else u1 = (g == ni - 4) ? m : u1;
        }
      }
      unsigned* dst = minbuf + (size_t)r_img * N + n_wave + lane;
      atomicMin(dst, __float_as_uint(v0));
      if (br < 64) atomicMin(dst + N, __float_as_uint(v1));
      if (NTN > 4 && lane < W - 64) {
        atomicMin(dst + 64, __float_as_uint(u0));
        if (br < 64) atomicMin(dst + N + 64, __float_as_uint(u1));
      }
    }
    const int n_dir1 = (br < 64 ? 2 : 1) * (NTN > 4 ? 2 : 1);
    if (!both_ways) return n_dir1;

    // ---- direction 2: columns are key patches of image c_img (and c_img + 1 from local column bc on)
    const int c_img = n_wave / P;
    const int bc = (c_img + 1) * P - n_wave;
    {
      float t0[4][4], t1[4][4];
      if (bc >= W) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float m = fminf(fminf(acc[mi][0][i], acc[mi][1][i]), fminf(acc[mi][2][i], acc[mi][3][i]));
#pragma unroll
            for (int ni = 4; ni < NTN; ++ni) m = fminf(m, acc[mi][ni][i]);
            t0[mi][i] = m;
          }
      } else {
        bool first[NTN];
#pragma unroll
        for (int ni = 0; ni < NTN; ++ni) first[ni] = 16 * ni + c < bc;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float a = inf, b = inf;
#pragma unroll
            for (int ni = 0; ni < NTN; ++ni) {
              const float v = acc[mi][ni][i];
              a = fminf(a, first[ni] ? v : inf);
              b = fminf(b, first[ni] ? inf : v);
            }
            t0[mi][i] = a;
            t1[mi][i] = b;
          }
      }
      float v0 = inf, v1 = inf;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float m = t0[mi][i];
          m = fminf(m, row_ror<8>(m)); m = fminf(m, row_ror<4>(m)); m = fminf(m, row_ror<2>(m)); m = fminf(m, row_ror<1>(m));
          v0 = (c == 4 * mi + i) ? m : v0;
          if (bc < W) {
            m = t1[mi][i];
            m = fminf(m, row_ror<8>(m)); m = fminf(m, row_ror<4>(m)); m = fminf(m, row_ror<2>(m)); m = fminf(m, row_ror<1>(m));
            v1 = (c == 4 * mi + i) ? m : v1;
          }
        }
      // lane (g, c) owns local row 16 (c >> 2) + 4 g + (c & 3): the wave's 64 rows, each once
      unsigned* dst = minbuf_t + (size_t)c_img * M + m_wave + 16 * (c >> 2) + 4 * g + (c & 3);
      atomicMin(dst, __float_as_uint(v0));
      if (bc < W) atomicMin(dst + M, __float_as_uint(v1));
    }
    return n_dir1 + (bc < W ? 2 : 1);
  }
  bool symmetric;        // A == B: only the upper triangle of tiles runs, each reduced both ways into minbuf
  bool two_way;          // A != B, both directions wanted: every tile of the rectangle is reduced both ways
};
template <> struct gemm::EpiTraits<EpiPairMin> { static constexpr bool kSymmetric = true; static constexpr bool kStaged = false; };

// The all-pairs kernel: gemm_hl7.h's 256x192 tile on eight waves with two-level accumulation (round 5: the largest tile
// whose per-wave state - two accumulator sets, which fp32-grade distances need - fits 256 registers; DESIGN.md sections
// 2 and 4; the 256x256 stream without the second accumulator is 10 % faster and 6x further from the fp64 truth).
// gemm_hl2.h's 256x128 ring kernel with the same accumulation (FLUSH), the kernel of rounds 1-4, stays as the pinned
// fallback (ac_set_gemm_policy(AC_GEMM_256x128)); the two agree to the bit.
// L2 window of the rasterisation: 6 tile rows per group (the 32 workgroups an XCD runs at a time then share 6 row
// panels and ~6 column panels); same-box A/B over 2..16 rows and two tile orders: profiles/r02_raster_ab.txt.
static int launch_pair_min(const void* A, const void* B, int M, int N, int K, const EpiPairMin& epi, hipStream_t stream,
                           int part = 0, int nparts = 1) {
#ifdef AC_PAIR_GROUP_M
  const int group_m = AC_PAIR_GROUP_M;       // diagnostic builds (A/B of the L2 window)
#else
  const int group_m = 6;
#endif
#ifdef AC_EXPERIMENTAL
  const int v = epi.two_way ? 0 : ac_gemm_policy();
  if (v == 101 && nparts == 1) return gemm::launch(A, B, M, N, K, epi, stream, 16);
  if (v == 103 && nparts == 1 && !epi.symmetric) return gemm3::launch(A, B, M, N, K, epi, stream, 4);
  if (v == 104) return gemm4::launch<EpiPairMin, 0>(A, B, M, N, K, epi, stream, group_m, part, nparts);
  if (v == 105) return gemm4::launch<EpiPairMin, 1>(A, B, M, N, K, epi, stream, group_m, part, nparts);
  if (v == 106) return gemm6::launch<EpiPairMin, true>(A, B, M, N, K, epi, stream, group_m, part, nparts);
  if (v == 109) return gemm5::launch<EpiPairMin, 4, true>(A, B, M, N, K, epi, stream, group_m, part, nparts);
  if (v == 108) return gemm5::launch<EpiPairMin>(A, B, M, N, K, epi, stream, 4, part, nparts);   // single accumulation
#endif
#ifndef AC_PAIR_DYNAMIC
#define AC_PAIR_DYNAMIC 1           // dynamic tile queues for launches of >= 4 tiles per CU (gemm_hl2.h); 0: A/B builds
#endif
  // Round 5: the 256x192 tile (gemm_hl7.h) unless the 256x128 kernel is pinned (ac_set_gemm_policy(AC_GEMM_256x128):
  // the two agree to the bit, tests/test_gemm_generations_gpu.py; -DAC_PAIR_TILE=128 builds pin it for A/B runs).
#ifndef AC_PAIR_TILE
#define AC_PAIR_TILE 192
#endif
#ifndef AC_G7_MT
#define AC_G7_MT 4                  // 8: the four-wave, 512-register form of the 256x192 kernel (A/B builds; gemm_hl7.h)
#endif
  if (AC_PAIR_TILE == 192 && ac_gemm_policy() != AC_GEMM_256x128)
    return gemm7::launch<EpiPairMin, AC_G7_MT>(A, B, M, N, K, epi, stream, group_m, part, nparts, AC_PAIR_DYNAMIC != 0);
  return gemm2::launch<EpiPairMin, true>(A, B, M, N, K, epi, stream, group_m, part, nparts, false, AC_PAIR_DYNAMIC != 0);
}

__global__ void fill_u32_kernel(unsigned* p, size_t n, unsigned v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// Preset of a matrix of minima / an accumulator (+inf bits, or zeros for the int64 sums as 2 words each): the engine's
// buffers are initialised by the library itself, on the launch stream, between its own kernels.
extern "C" int ac_fill_u32(uint32_t* dst, int64_t count, uint32_t value, void* stream) {
  AC_REQUIRE(dst || count <= 0, "ac_fill_u32: null pointer");
  AC_REQUIRE(((uintptr_t)dst & 3) == 0, "ac_fill_u32: dst must be 4-byte aligned");
  if (count <= 0) return AC_OK;
  const int64_t blocks = (count + 1023) / 1024;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream), dst,
                     (size_t)count, value);
  return ac_check_launch("ac_fill_u32");
}

__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

extern "C" int ac_fill_u64(uint64_t* dst, int64_t count, uint64_t value, void* stream) {
  AC_REQUIRE(dst || count <= 0, "ac_fill_u64: null pointer");
  AC_REQUIRE(((uintptr_t)dst & 7) == 0, "ac_fill_u64: dst must be 8-byte aligned");
  if (count <= 0) return AC_OK;
  const int64_t blocks = (count + 1023) / 1024;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<unsigned long long*>(dst), (size_t)count, (unsigned long long)value);
  return ac_check_launch("ac_fill_u64");
}

// ---------------------------------------------------- from min d2 to the weights w
// minbuf[j][col] (min d2 between query patch `col` and key image j) -> w.  The unsupervised mean runs
// over ALL key images of the category, which a large or sharded problem sees block by block, on different
// ranks, in any order - so the sum is kept in 64-bit FIXED POINT: each sqrt (an fp32 value) is scaled by
// 2^shift and rounded to an integer, and integer addition is associative.  Any blocking, any rank count and
// atomics all give the same bits; one rounding to fp32 at the end.  shift comes from the data: the largest
// power of two (capped at 40, i.e. a resolution of 9e-13) for which n_keys * 2 sqrt(max |z|^2) stays
// below 2^62.  The supervised min needs none of this: min over uint32 bit patterns is exact.
__global__ void max_f32_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like their bits
}

extern "C" int ac_max_f32(const float* x, int64_t n, float* out, void* stream) {
  AC_REQUIRE(out && (x || n <= 0), "ac_max_f32: null pointer");
  if (n <= 0) return AC_OK;
  unsigned grid = (unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
  hipLaunchKernelGGL(max_f32_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, n, reinterpret_cast<unsigned*>(out));
  return ac_check_launch("ac_max_f32");
}

__global__ void fixed_shift_kernel(const float* __restrict__ max_sq, int64_t n_keys, int* __restrict__ shift) {
  // every distance is <= |a| + |b| <= 2 sqrt(max_sq); bound = n_keys of them (+1: never log2(0))
  const double bound = (double)(n_keys > 0 ? n_keys : 1) * 2.0 * sqrt((double)max_sq[0]) + 1.0;
  int e = 0;
  (void)frexp(bound, &e);                       // bound < 2^e
  int s = 62 - e;
  shift[0] = s > 40 ? 40 : (s < -60 ? -60 : s);
}

extern "C" int ac_pair_fixed_shift(const float* max_sq, int64_t n_keys, int32_t* shift, void* stream) {
  AC_REQUIRE(max_sq && shift, "ac_pair_fixed_shift: null pointer");
  hipLaunchKernelGGL(fixed_shift_kernel, dim3(1), dim3(1), 0, as_stream(stream), max_sq, n_keys, shift);
  return ac_check_launch("ac_pair_fixed_shift");
}

__global__ void pair_accumulate_kernel(const unsigned* __restrict__ minbuf, int64_t nk, int64_t ld, int64_t ncols, int P,
                                       int reduce, int64_t self_first, const int* __restrict__ shift_p,
                                       void* __restrict__ acc) {
  int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  if (reduce == AC_REDUCE_MEAN_OTHERS) {
    const int64_t self = self_first + col / P;
    const double scale = ldexp(1.0, shift_p[0]);
    long long s = 0;
    for (int64_t j = 0; j < nk; ++j)
      if (j != self) s += __double2ll_rn((double)sqrtf(__uint_as_float(minbuf[j * ld + col])) * scale);
    static_cast<long long*>(acc)[col] += s;      // one thread per column, launches on one stream are ordered
  } else {
    unsigned m = static_cast<unsigned*>(acc)[col];
    for (int64_t j = 0; j < nk; ++j) m = min(m, minbuf[j * ld + col]);
    static_cast<unsigned*>(acc)[col] = m;
  }
}

extern "C" int ac_pair_accumulate(const uint32_t* minbuf, int64_t nk, int64_t ld, int64_t ncols, int P, int reduce,
                                  int64_t self_first, const int32_t* shift, void* acc, void* stream) {
  AC_REQUIRE(minbuf && acc, "ac_pair_accumulate: null pointer");
  AC_REQUIRE(nk >= 0 && ncols >= 0 && ld >= ncols && P > 0, "ac_pair_accumulate: bad shape");
  AC_REQUIRE(reduce == AC_REDUCE_MEAN_OTHERS || reduce == AC_REDUCE_MIN_ALL, "ac_pair_accumulate: bad reduce=%d", reduce);
  AC_REQUIRE(reduce == AC_REDUCE_MIN_ALL || shift, "ac_pair_accumulate: the mean needs the fixed-point shift");
  if (nk == 0 || ncols == 0) return AC_OK;
  hipLaunchKernelGGL(pair_accumulate_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, as_stream(stream), minbuf,
                     nk, ld, ncols, P, reduce, self_first, shift, acc);
  return ac_check_launch("ac_pair_accumulate");
}

__global__ void pair_finalize_kernel(const void* __restrict__ acc, int64_t ncols, int reduce, int64_t count,
                                     const int* __restrict__ shift_p, float* __restrict__ w) {
  int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  if (reduce == AC_REDUCE_MEAN_OTHERS) {
    const double sum = ldexp((double)static_cast<const long long*>(acc)[col], -shift_p[0]);
    w[col] = (float)(sum / (double)count);        // count == 0 -> NaN, as torch.mean of an empty tensor
  } else {
    w[col] = sqrtf(__uint_as_float(static_cast<const unsigned*>(acc)[col]));
  }
}

extern "C" int ac_pair_finalize(const void* acc, int64_t ncols, int reduce, int64_t count, const int32_t* shift, float* w,
                                void* stream) {
  AC_REQUIRE(acc && w, "ac_pair_finalize: null pointer");
  AC_REQUIRE(reduce == AC_REDUCE_MEAN_OTHERS || reduce == AC_REDUCE_MIN_ALL, "ac_pair_finalize: bad reduce=%d", reduce);
  AC_REQUIRE(reduce == AC_REDUCE_MIN_ALL || shift, "ac_pair_finalize: the mean needs the fixed-point shift");
  if (ncols <= 0) return AC_OK;
  hipLaunchKernelGGL(pair_finalize_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, as_stream(stream), acc,
                     ncols, reduce, count, shift, w);
  return ac_check_launch("ac_pair_finalize");
}

static const size_t kMinbufBudget = (size_t)1 << 30;   // bytes of minbuf per query chunk

static int64_t query_chunk(int64_t nq, int64_t nk, int P) {
  size_t per_img = (size_t)nk * P * 4;
  int64_t c = (int64_t)(kMinbufBudget / (per_img ? per_img : 1));
  if (c < 1) c = 1;
  return c < nq ? c : nq;
}

extern "C" size_t ac_pair_min_dist_workspace_bytes(int64_t nq, int64_t nk, int P, int D) {
  if (nq <= 0 || nk <= 0 || P <= 0 || D <= 0) return 256;
  size_t hl_k = align_up((size_t)nk * P * D * 4, 256), hl_q = align_up((size_t)nq * P * D * 4, 256);
  size_t n_k = align_up((size_t)nk * P * 4, 256), n_q = align_up((size_t)nq * P * 4, 256);
  size_t mb = align_up((size_t)query_chunk(nq, nk, P) * P * nk * 4, 256);
  size_t acc = align_up((size_t)nq * P * 8, 256);
  return hl_k + hl_q + n_k + n_q + mb + acc + 512 + 1024;
}

extern "C" int ac_pair_min_dist(const float* Zq, int64_t nq, const float* Zk, int64_t nk, int P, int D, int reduce,
                                int64_t self_offset, float* w, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = as_stream(stream_);
  AC_REQUIRE(Zq && Zk && w && ws, "ac_pair_min_dist: null pointer");
  AC_REQUIRE(nq > 0 && nk > 0 && P > 0, "ac_pair_min_dist: empty input (nq=%lld nk=%lld P=%d)", (long long)nq,
             (long long)nk, P);
  AC_REQUIRE(D > 0 && D % 32 == 0, "ac_pair_min_dist: D=%d must be a positive multiple of 32", D);
  AC_REQUIRE(reduce == AC_REDUCE_MEAN_OTHERS || reduce == AC_REDUCE_MIN_ALL, "ac_pair_min_dist: bad reduce=%d", reduce);
  AC_REQUIRE((nk * P) <= 0x7fffff00LL && (nq * P) <= 0x7fffff00LL, "ac_pair_min_dist: too many patches for int32 tiles");
  if (ws_bytes < ac_pair_min_dist_workspace_bytes(nq, nk, P, D))
    return ac_fail(AC_EWORKSPACE, "ac_pair_min_dist: workspace %zu < %zu", ws_bytes,
                   ac_pair_min_dist_workspace_bytes(nq, nk, P, D));
  Arena ar(ws, ws_bytes);
  const bool same = (Zq == Zk && nq == nk);
  uint8_t* hk = ar.take<uint8_t>((size_t)nk * P * D * 4);
  float* sk = ar.take<float>((size_t)nk * P);
  uint8_t* hq = same ? hk : ar.take<uint8_t>((size_t)nq * P * D * 4);
  float* sq = same ? sk : ar.take<float>((size_t)nq * P);
  const int64_t chunk = query_chunk(nq, nk, P);
  unsigned* minbuf = ar.take<unsigned>((size_t)chunk * P * nk);
  long long* acc = ar.take<long long>((size_t)nq * P);
  float* max_sq = ar.take<float>(64);
  int32_t* shift = reinterpret_cast<int32_t*>(max_sq + 32);
  if (!ar.ok()) return ac_fail(AC_EWORKSPACE, "ac_pair_min_dist: workspace carve failed");

  int rc = ac_pack_hl32(Zk, nk * P, D, D, hk, sk, stream_);
  if (rc) return rc;
  if (!same) {
    rc = ac_pack_hl32(Zq, nq * P, D, D, hq, sq, stream_);
    if (rc) return rc;
  }
  const bool mean = reduce == AC_REDUCE_MEAN_OTHERS;
  const size_t ncols_all = (size_t)nq * P;
  if (mean) {
    if (hipMemsetAsync(max_sq, 0, 256, stream) != hipSuccess || hipMemsetAsync(acc, 0, ncols_all * 8, stream) != hipSuccess)
      return ac_fail(AC_ELAUNCH, "ac_pair_min_dist: hipMemsetAsync failed");
    if ((rc = ac_max_f32(sk, nk * P, max_sq, stream_))) return rc;
    if (!same && (rc = ac_max_f32(sq, nq * P, max_sq, stream_))) return rc;
    if ((rc = ac_pair_fixed_shift(max_sq, nk, shift, stream_))) return rc;
  } else {
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((ncols_all + 255) / 256 < 4096 ? (ncols_all + 255) / 256 : 4096)),
                       dim3(256), 0, stream, reinterpret_cast<unsigned*>(acc), ncols_all, 0x7f800000u);
  }
  const int M = (int)(nk * P);
  for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
    const int64_t qn = (q0 + chunk <= nq) ? chunk : nq - q0;
    const int N = (int)(qn * P);
    size_t cells = (size_t)nk * N;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096)),
                       dim3(256), 0, stream, minbuf, cells, 0x7f800000u);
    // whole square in one launch -> C = C^T
    EpiPairMin epi{sk, sq + q0 * P, minbuf, minbuf, M, N, P, same && chunk == nq, false};
    rc = launch_pair_min(hk, hq + (size_t)q0 * P * D * 4, M, N, D, epi, stream);
    if (rc) return rc;
    void* acc_q = mean ? static_cast<void*>(acc + q0 * P) : static_cast<void*>(reinterpret_cast<unsigned*>(acc) + q0 * P);
    if ((rc = ac_pair_accumulate(minbuf, nk, N, N, P, reduce, self_offset + q0, shift, acc_q, stream_))) return rc;
  }
  int64_t count = nk;
  if (mean) {
    // every query image must see the same number of "other" key images: self_offset places the queries inside
    // the key list (or outside it: then nothing is skipped)
    const bool inside = self_offset >= 0 && self_offset + nq <= nk;
    const bool outside = self_offset + nq <= 0 || self_offset >= nk;
    AC_REQUIRE(inside || outside, "ac_pair_min_dist: the query block must lie wholly inside or outside the key list");
    count = inside ? nk - 1 : nk;
  }
  return ac_pair_finalize(acc, (int64_t)ncols_all, reduce, count, shift, w, stream_);
}

// Staged form of the same computation (pack -> min-d2 GEMM -> accumulate -> finalize) for callers that keep
// the packed operands resident, shard the category over GPUs, or time the MFMA kernel on its own.
extern "C" int ac_pair_min_sq_packed(const void* Zk_hl, const float* k_sq, int64_t nk, const void* Zq_hl,
                                     const float* q_sq, int64_t nq, int P, int D, uint32_t* minbuf, uint32_t* minbuf_t,
                                     int part, int nparts, void* stream) {
  AC_REQUIRE(Zk_hl && k_sq && Zq_hl && q_sq && minbuf, "ac_pair_min_sq_packed: null pointer");
  AC_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, "ac_pair_min_sq_packed: bad part %d/%d", part, nparts);
  AC_REQUIRE(nq > 0 && nk > 0 && P > 0 && D > 0 && D % 32 == 0, "ac_pair_min_sq_packed: bad shape");
  AC_REQUIRE((nk * P) <= 0x7fffff00LL && (nq * P) <= 0x7fffff00LL, "ac_pair_min_sq_packed: too many patches");
  const bool same = Zk_hl == Zq_hl && nk == nq;
  AC_REQUIRE(!(same && minbuf_t), "ac_pair_min_sq_packed: keys == queries is the symmetric case; minbuf_t must be NULL");
  AC_REQUIRE(!(minbuf_t && nparts != 1), "ac_pair_min_sq_packed: two-way blocks are not sliced");
  EpiPairMin epi{k_sq, q_sq, minbuf, same ? minbuf : minbuf_t, (int)(nk * P), (int)(nq * P), P, same, minbuf_t != nullptr};
  return launch_pair_min(Zk_hl, Zq_hl, (int)(nk * P), (int)(nq * P), D, epi, as_stream(stream), part, nparts);
}

// ------------------------------------------- supervised weights: pruned exact nearest neighbour
// w_i[p] = min over EVERY patch of the normal-image bank (utils.py:230-237) is a 1-nearest-neighbour search, and an
// exact one does not need the exact distance of every pair:
//   pass 1  d~2(a, b) = |a|^2 + |b|^2 - 2 ah.bh with the high halves only - ONE third of the MFMA work and half the
//           operand bytes, on the 256x256 kernel (accuracy no longer depends on it).  |d~2 - d2| <= delta with
//           delta = 2 (2^-10 + ...) |a| |b| (Cauchy-Schwarz on the dropped terms ah.bl + al.bh + al.bl, |lo| <= 2^-11 |x|)
//           plus the fp32 accumulation error, which is two orders smaller.  For the true nearest neighbour b* and any
//           other row b:  d~2(b*) <= d2(b*) + delta <= d2(b) + delta <= d~2(b) + 2 delta,  so b* survives when every
//           row with  d~2 <= (running minimum of d~2) + slack,  slack >= 2 delta,  is kept as a candidate - whatever
//           order the tiles run in, because the running minimum only falls.
//   pass 2  the candidates (typically one to three per query patch) are re-measured exactly: sum_k (a_k - b_k)^2 in
//           fp64 on the 22-bit split values (hi + lo) - no cancellation at all - and the minimum is taken.
// A query whose candidate list overflows raises a flag and the caller falls back to the exhaustive kernel.
struct EpiNNCand {
  const float* key_sq;    // [M] bank rows
  const float* qry_sq;    // [N] query rows
  const float* slack;     // device [3]: relative (x |query|) and absolute part (+ the out-of-range flag, host side)
  unsigned* approx_min;   // [N] running min of d~2 (float bits), across launches
  int* cand;              // [N][cap] bank-row indices (+ row_offset)
  int* count;             // [N]
  int M, N, cap, row_offset;
  template <class F>
  __device__ void operator()(const F& f, int m_wave, int n_wave, int lane, bool = false) const {
    if (m_wave >= M) return;                      // wave-uniform
    const float sl_rel = slack[0], sl_abs = slack[1];      // slack of query a: sl_rel * |a| + sl_abs
    float ksq[F::NT][F::NR];
#pragma unroll
    for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
      for (int i = 0; i < F::NR; ++i) ksq[mi][i] = key_sq[min(m_wave + F::row(mi, i, lane), M - 1)];
#pragma unroll
    for (int ni = 0; ni < F::NT; ++ni) {
      const int col = n_wave + F::col(ni, lane);
      const int colc = min(col, N - 1);
      const float qs = qry_sq[colc];
      float d2[F::NT][F::NR];
      float mn = __builtin_inff();
#pragma unroll
      for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
        for (int i = 0; i < F::NR; ++i) {
          const bool live = m_wave + F::row(mi, i, lane) < M;
          const float v = fmaxf((ksq[mi][i] - f.acc[mi][ni][i]) + (qs - f.acc[mi][ni][i]), 0.f);
          d2[mi][i] = live ? v : __builtin_inff();
          mn = fminf(mn, d2[mi][i]);
        }
      mn = F::min_over_row_groups(mn);            // min over the wave's 64 rows, in every lane of the column
      unsigned old = 0x7f800000u;
      if (F::first_row_group(lane) && col < N) old = atomicMin(approx_min + col, __float_as_uint(mn));
      old = __shfl(old, lane & (F::TW - 1), 64);  // the lane of the first row group that owns this column
      const float thr = fminf(__uint_as_float(old), mn) + (sl_rel * sqrtf(qs) * 1.000001f + sl_abs);
      if (col < N) {
#pragma unroll
        for (int mi = 0; mi < F::NT; ++mi)
#pragma unroll
          for (int i = 0; i < F::NR; ++i)
            if (d2[mi][i] <= thr) {
              const int slot = atomicAdd(count + col, 1);
              if (slot < cap) cand[(size_t)col * cap + slot] = row_offset + m_wave + F::row(mi, i, lane);
            }
      }
    }
  }
};

// hi plane of an hl32 image -> plain fp16 rows [rows][K] (the operand of pass 1)
__global__ __launch_bounds__(256) void h16_from_hl32_kernel(const uint8_t* __restrict__ src, int64_t rows, int K,
                                                            uint8_t* __restrict__ dst) {
  const int64_t groups = (int64_t)K / 32;                         // 128-byte hl32 groups per row
  const int64_t total = rows * groups * 4;                        // 16-byte pieces of hi halves
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t g = e >> 2;
    const int piece = (int)(e & 3);
    *reinterpret_cast<u32x4*>(dst + g * 64 + piece * 16) = *reinterpret_cast<const u32x4*>(src + g * 128 + piece * 16);
  }
}

extern "C" int ac_h16_from_hl32(const void* hl32, int64_t rows, int K, void* dst_h16, void* stream) {
  AC_REQUIRE(hl32 && dst_h16, "ac_h16_from_hl32: null pointer");
  AC_REQUIRE(K > 0 && K % 64 == 0 && rows >= 0, "ac_h16_from_hl32: K=%d must be a positive multiple of 64", K);
  if (rows == 0) return AC_OK;
  const int64_t total = rows * (K / 32) * 4;
  hipLaunchKernelGGL(h16_from_hl32_kernel, dim3((unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535)), dim3(256),
                     0, as_stream(stream), static_cast<const uint8_t*>(hl32), rows, K, static_cast<uint8_t*>(dst_h16));
  return ac_check_launch("ac_h16_from_hl32");
}

__global__ void nn_slack_kernel(const float* __restrict__ max_sq_k, const float* __restrict__ max_sq_q, float* __restrict__ slack) {
  // slack of a query a = 2 delta(a), delta(a) = 2 * (2^-10 + 2^-22) |a| |b|max (dropped cross terms) + 2 * 160 * 2^-24 |a| |b|max
  // (fp32 accumulation of 128 MFMA steps, rigorous gamma bound), rounded up; both pairs of the proof share the query, so
  // the bound may use ITS norm.  slack[0] multiplies |a|; slack[1] is an absolute term for values in fp16's subnormal
  // range, whose low halves are bounded by 2^-25 instead of 2^-11 |x|.
  const double nb = sqrt((double)max_sq_k[0]), na = sqrt((double)max_sq_q[0]);
  slack[0] = (float)(2.0 * 2.0 * (9.77e-4 + 2.0e-5) * nb * 1.01);
  slack[1] = (float)(1.0e-5 * (na + nb)) + 1e-30f;
  // The bound assumes finite fp16 high halves: |x| < 65504 for every element, which |row| < 65504 guarantees.  Outside
  // that range (or with a non-finite norm) the first pass may see inf - inf; slack[2] != 0 tells the caller to run the
  // exhaustive split-fp16 blocks instead, which surface inf / NaN as the reference's cdist does.
  slack[2] = (nb < 65504.0 && na < 65504.0) ? 0.f : 1.f;      // false for NaN norms as well
}

extern "C" int ac_nn_slack(const float* max_sq_keys, const float* max_sq_queries, float* slack, void* stream) {
  AC_REQUIRE(max_sq_keys && max_sq_queries && slack, "ac_nn_slack: null pointer");
  hipLaunchKernelGGL(nn_slack_kernel, dim3(1), dim3(1), 0, as_stream(stream), max_sq_keys, max_sq_queries, slack);
  return ac_check_launch("ac_nn_slack");
}

extern "C" int ac_nn_candidates(const void* Zk_h16, const float* k_sq, int64_t nk_rows, const void* Zq_h16, const float* q_sq,
                                int64_t nq_rows, int D, const float* slack, uint32_t* approx_min, int32_t* cand,
                                int32_t* cand_count, int cap, int64_t key_row_offset, void* stream) {
  AC_REQUIRE(Zk_h16 && k_sq && Zq_h16 && q_sq && slack && approx_min && cand && cand_count, "ac_nn_candidates: null pointer");
  AC_REQUIRE(nk_rows > 0 && nq_rows > 0 && D > 0 && D % 64 == 0 && cap > 0, "ac_nn_candidates: bad shape");
  AC_REQUIRE(nk_rows <= 0x7fffff00LL && nq_rows <= 0x7fffff00LL && key_row_offset >= 0 &&
             key_row_offset + nk_rows <= 0x7fffffffLL, "ac_nn_candidates: too many rows");
  EpiNNCand epi{k_sq, q_sq, slack, approx_min, cand, cand_count, (int)nk_rows, (int)nq_rows, cap, (int)key_row_offset};
  // plain fp16 rows: D / 2 "hl32 columns" give the right row stride (D * 2 bytes) and stage count (D / 64)
  return gemm5::launch<EpiNNCand, 8, false, 2, true>(Zk_h16, Zq_h16, (int)nk_rows, (int)nq_rows, D / 2, epi, as_stream(stream), 6);
}

// pass 2: one wave per query row.  Lane l owns the hl32 groups l, l + 64, ... (32 columns each) of the 16 KB row;
// the query's values stay in registers (hi + lo is exact in fp32: 22 bits) while its candidates stream by, every
// candidate row as GPL * 8 independent 16-byte loads per lane.
template <int GPL>
__global__ __launch_bounds__(256) void nn_refine_kernel(const uint8_t* __restrict__ Zk, int64_t nk_rows, const uint8_t* __restrict__ Zq,
                                                        int64_t nq_rows, int D, const int* __restrict__ cand,
                                                        const int* __restrict__ count, int cap, int64_t key_row_offset,
                                                        double* __restrict__ best, int* __restrict__ overflow,
                                                        int* __restrict__ overflow_rows, const float* __restrict__ slack) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq_rows) return;                                        // wave-uniform
  int n = count[q];
  if (n > cap || (slack && slack[2] != 0.f)) {                     // the caller re-measures this query exhaustively
    // (slack[2]: norms outside the fp16 range of the first pass - nothing it produced can be trusted)
    if (lane == 0) { *overflow = 1; if (overflow_rows) overflow_rows[q] = 1; }
    return;
  }
  if (n == 0) return;
  const int groups = D / 32;
  const size_t ld = (size_t)D * 4;
  float av[GPL][32];
  const uint8_t* a = Zq + (size_t)q * ld;
#pragma unroll
  for (int u = 0; u < GPL; ++u) {
    const int g = min(lane + 64 * u, groups - 1);
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) {
      const half8 h = *reinterpret_cast<const half8*>(a + (size_t)g * 128 + piece * 16);
      const half8 l = *reinterpret_cast<const half8*>(a + (size_t)g * 128 + 64 + piece * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) av[u][piece * 8 + j] = (float)h[j] + (float)l[j];
    }
  }
  double bestd = best[q];
  for (int c = 0; c < n; ++c) {
    const int64_t row = (int64_t)cand[q * cap + c] - key_row_offset;
    if (row < 0 || row >= nk_rows) continue;                       // (cannot happen; guards the address)
    const uint8_t* b = Zk + (size_t)row * ld;
    half8 bh[GPL][4], bl[GPL][4];
#pragma unroll
    for (int u = 0; u < GPL; ++u) {
      const int g = min(lane + 64 * u, groups - 1);
#pragma unroll
      for (int piece = 0; piece < 4; ++piece) {
        bh[u][piece] = *reinterpret_cast<const half8*>(b + (size_t)g * 128 + piece * 16);
        bl[u][piece] = *reinterpret_cast<const half8*>(b + (size_t)g * 128 + 64 + piece * 16);
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < GPL; ++u) {
      if (lane + 64 * u < groups) {
#pragma unroll
        for (int piece = 0; piece < 4; ++piece)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double d = (double)av[u][piece * 8 + j] - ((double)(float)bh[u][piece][j] + (double)(float)bl[u][piece][j]);
            acc = fma(d, d, acc);
          }
      }
    }
    acc = wave_sum(acc);
    bestd = fmin(bestd, acc);
  }
  if (lane == 0) best[q] = bestd;
}

__global__ void nn_finalize_kernel(const double* __restrict__ best, int64_t n, float* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = (float)sqrt(best[i]);
}

extern "C" int ac_nn_finalize(const double* best_d2, int64_t n, float* w, void* stream) {
  AC_REQUIRE(best_d2 && w, "ac_nn_finalize: null pointer");
  if (n <= 0) return AC_OK;
  hipLaunchKernelGGL(nn_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), best_d2, n, w);
  return ac_check_launch("ac_nn_finalize");
}

extern "C" int ac_nn_refine(const void* Zk_hl32, int64_t nk_rows, const void* Zq_hl32, int64_t nq_rows, int D, const int32_t* cand,
                            const int32_t* cand_count, int cap, int64_t key_row_offset, double* best_d2, int32_t* overflow_flag,
                            int32_t* overflow_rows, const float* slack, void* stream) {
  AC_REQUIRE(Zk_hl32 && Zq_hl32 && cand && cand_count && best_d2 && overflow_flag, "ac_nn_refine: null pointer");
  AC_REQUIRE(nk_rows > 0 && nq_rows > 0 && D > 0 && D % 32 == 0 && cap > 0, "ac_nn_refine: bad shape");
  AC_REQUIRE(D <= 32 * 64 * 4, "ac_nn_refine: D=%d above 8192", D);
  const int gpl = (D / 32 + 63) / 64;               // hl32 groups per lane
  const dim3 grid((unsigned)((nq_rows + 3) / 4));
#define AC_REFINE(G)                                                                                                   \
  hipLaunchKernelGGL(nn_refine_kernel<G>, grid, dim3(256), 0, as_stream(stream), static_cast<const uint8_t*>(Zk_hl32),   \
                     nk_rows, static_cast<const uint8_t*>(Zq_hl32), nq_rows, D, cand, cand_count, cap, key_row_offset,     \
                     best_d2, overflow_flag, overflow_rows, slack)
  if (gpl <= 1) AC_REFINE(1);
  else if (gpl == 2) AC_REFINE(2);
  else AC_REFINE(4);
#undef AC_REFINE
  return ac_check_launch("ac_nn_refine");
}

// The queries whose candidate list overflowed on some key set (overflow_rows[q] != 0), compacted on the device in
// ascending order: one workgroup, every thread counts a contiguous chunk, one scan of the 1024 counts in LDS.
__global__ __launch_bounds__(1024) void nn_overflow_compact_kernel(const int* __restrict__ rows, int64_t n, int* __restrict__ idx,
                                                                   int* __restrict__ n_out) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024, lo = t * per, hi = (lo + per < n) ? lo + per : n;
  int c = 0;
  for (int64_t i = lo; i < hi; ++i) c += rows[i] != 0;
  cnt[t] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                       // inclusive scan (Hillis-Steele)
    const int v = (t >= o) ? cnt[t - o] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  int pos = cnt[t] - c;
  for (int64_t i = lo; i < hi; ++i)
    if (rows[i] != 0) idx[pos++] = (int)i;
  if (t == 1023) n_out[0] = cnt[1023];
}

extern "C" int ac_nn_overflow_compact(const int32_t* overflow_rows, int64_t n, int32_t* idx, int32_t* n_out, void* stream) {
  AC_REQUIRE(overflow_rows && idx && n_out, "ac_nn_overflow_compact: null pointer");
  AC_REQUIRE(n >= 0 && n <= 0x7fffff00LL, "ac_nn_overflow_compact: n=%lld out of range", (long long)n);
  hipLaunchKernelGGL(nn_overflow_compact_kernel, dim3(1), dim3(1024), 0, as_stream(stream), overflow_rows, n, idx, n_out);
  return ac_check_launch("ac_nn_overflow_compact");
}

// dst row r = src row idx[min(r, n_idx - 1)], r < n_rows (rows beyond the list repeat its last row: padding of the
// compact query copy to whole images).  One wave per row, 16-byte pieces when the row allows, dwords otherwise.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ src, int64_t row_bytes, const int* __restrict__ idx,
                                                          int64_t n_idx, int64_t n_rows, uint8_t* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const uint8_t* s = src + (size_t)idx[r < n_idx ? r : n_idx - 1] * row_bytes;
  uint8_t* d = dst + (size_t)r * row_bytes;
  if ((row_bytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    for (int64_t o = (int64_t)lane * 16; o < row_bytes; o += 1024) *reinterpret_cast<u32x4*>(d + o) = *reinterpret_cast<const u32x4*>(s + o);
  } else {
    for (int64_t o = (int64_t)lane * 4; o < row_bytes; o += 256) *reinterpret_cast<uint32_t*>(d + o) = *reinterpret_cast<const uint32_t*>(s + o);
  }
}

extern "C" int ac_gather_rows(const void* src, int64_t row_bytes, const int32_t* idx, int64_t n_idx, int64_t n_rows, void* dst,
                              void* stream) {
  AC_REQUIRE(src && idx && dst, "ac_gather_rows: null pointer");
  AC_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0, "ac_gather_rows: row_bytes=%lld must be a positive multiple of 4", (long long)row_bytes);
  AC_REQUIRE(n_idx > 0 && n_rows >= n_idx, "ac_gather_rows: n_idx=%lld n_rows=%lld", (long long)n_idx, (long long)n_rows);
  AC_REQUIRE((n_rows + 3) / 4 <= 0x7fffffffLL, "ac_gather_rows: too many rows");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint8_t*>(src), row_bytes, idx, n_idx, n_rows, static_cast<uint8_t*>(dst));
  return ac_check_launch("ac_gather_rows");
}

// best_d2[idx[i]] = the exhaustive kernel's min d2 of compact query i (uint32 float bits): those queries take the
// exhaustive answer as it stands, so they equal the un-pruned path bit for bit.
__global__ void nn_scatter_kernel(const unsigned* __restrict__ acc_bits, const int* __restrict__ idx, int64_t n, double* __restrict__ best) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) best[idx[i]] = (double)__uint_as_float(acc_bits[i]);
}

extern "C" int ac_nn_scatter_exhaustive(const uint32_t* acc_bits, const int32_t* idx, int64_t n, double* best_d2, void* stream) {
  AC_REQUIRE(acc_bits && idx && best_d2, "ac_nn_scatter_exhaustive: null pointer");
  if (n <= 0) return AC_OK;
  hipLaunchKernelGGL(nn_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), acc_bits, idx, n, best_d2);
  return ac_check_launch("ac_nn_scatter_exhaustive");
}

// ------------------------------------------------------------ alpha softmax
// One block per image.  fp64 throughout (utils.py:245, 253-254): alpha = k*exp(w/tau)/sum.
// The max is subtracted before exp - mathematically identical, and finite where the
// reference's un-shifted exp overflows.  tau == 0 (math.isclose(tau, 0) with its default
// abs_tol = 0 is true only for exact zero): indicator of the maximum over #ties.
__global__ __launch_bounds__(256) void alpha_softmax_kernel(const float* __restrict__ w, int P, double tau, double k,
                                                            double* __restrict__ alpha, float* __restrict__ alpha_f32) {
  __shared__ double red[4];
  __shared__ double bcast;
  const float* wi = w + (size_t)blockIdx.x * P;
  double* ai = alpha + (size_t)blockIdx.x * P;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float m = -__builtin_inff();
  for (int p = tid; p < P; p += 256) m = fmaxf(m, wi[p]);
  m = wave_max(m);
  if (lane == 0) red[wv] = (double)m;
  __syncthreads();
  if (tid == 0) bcast = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  const double mx = bcast;
  __syncthreads();
  double s = 0.0;
  const bool hard = (tau == 0.0);
  const double inv_tau = hard ? 0.0 : 1.0 / tau;
  for (int p = tid; p < P; p += 256) {
    double v = hard ? ((double)wi[p] == mx ? 1.0 : 0.0) : k * exp(inv_tau * ((double)wi[p] - mx));
    ai[p] = v;
    s += v;
  }
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  if (tid == 0) bcast = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const double tot = bcast;
  // alpha_f32: examples/main.py:294's `.float()` of the same value (round to nearest even), for the weighted average
  float* af = alpha_f32 ? alpha_f32 + (size_t)blockIdx.x * P : nullptr;
  for (int p = tid; p < P; p += 256) {
    const double a = ai[p] / tot;
    ai[p] = a;
    if (af) af[p] = (float)a;
  }
}

extern "C" int ac_alpha_softmax_f32(const float* w, int64_t N, int P, double tau, double k, double* alpha, float* alpha_f32,
                                    void* stream) {
  AC_REQUIRE(w && alpha, "ac_alpha_softmax: null pointer");
  AC_REQUIRE(P > 0 && N >= 0 && N <= 0x7fffffff, "ac_alpha_softmax: bad shape N=%lld P=%d", (long long)N, P);
  AC_REQUIRE(tau >= 0.0, "ac_alpha_softmax: tau=%g must be >= 0", tau);
  if (N == 0) return AC_OK;
  hipLaunchKernelGGL(alpha_softmax_kernel, dim3((unsigned)N), dim3(256), 0, as_stream(stream), w, P, tau, k, alpha, alpha_f32);
  return ac_check_launch("alpha_softmax");
}

extern "C" int ac_alpha_softmax(const float* w, int64_t N, int P, double tau, double k, double* alpha, void* stream) {
  return ac_alpha_softmax_f32(w, N, P, tau, k, alpha, nullptr, stream);
}

// --------------------------------------------------------- weighted average
// X[i][d] = sum_p alpha[i][p] * Z[i][p][d].  HBM-bound (reads Z once).  Block = (image, 256 columns):
// 64 threads x 4 columns wide (a wave-row reads 1 KB contiguous of one patch row) and 4 patch
// phases deep; every thread keeps 4 independent 16-byte loads in flight (patches p, p+4, p+8, p+12 of
// its phase), the phases meet in LDS in a fixed order.  No atomics: X is bitwise reproducible.
__global__ __launch_bounds__(256) void weighted_avg_kernel(const float* __restrict__ alpha, const float* __restrict__ Z,
                                                           int P, int D, float* __restrict__ X) {
  __shared__ f32x4 red[4][64];
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int d = (blockIdx.x * 64 + lane) * 4;
  const bool live = d < D;
  const float* a = alpha + (size_t)img * P;
  const float* z = Z + ((size_t)img * P) * D + (live ? d : 0);
  f32x4 acc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  int p = phase;
  for (; p + 12 < P; p += 16) {
    f32x4 v[4];
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = *reinterpret_cast<const f32x4*>(z + (size_t)(p + 4 * u) * D); w[u] = a[p + 4 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[u][j] = fmaf(w[u], v[u][j], acc[u][j]);
  }
  for (int u = 0; p < P; p += 4, ++u) {                  // tail: at most 3 more patches of this phase
    f32x4 v = *reinterpret_cast<const f32x4*>(z + (size_t)p * D);
    const float w = a[p];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[u & 3][j] = fmaf(w, v[j], acc[u & 3][j]);
  }
  f32x4 s;
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
  red[phase][lane] = s;
  __syncthreads();
  if (phase == 0 && live) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
    *reinterpret_cast<f32x4*>(X + (size_t)img * D + d) = o;
  }
}

extern "C" int ac_weighted_avg(const float* alpha, const float* Z, int64_t N, int P, int D, float* X, void* stream_) {
  hipStream_t stream = as_stream(stream_);
  AC_REQUIRE(alpha && Z && X, "ac_weighted_avg: null pointer");
  AC_REQUIRE(P > 0 && D > 0 && D % 4 == 0, "ac_weighted_avg: bad shape P=%d D=%d (D %% 4 == 0)", P, D);
  AC_REQUIRE(N >= 0 && N <= 65535, "ac_weighted_avg: N=%lld out of range (<= 65535 per call)", (long long)N);
  if (N == 0) return AC_OK;
  hipLaunchKernelGGL(weighted_avg_kernel, dim3((D / 4 + 63) / 64, (unsigned)N), dim3(256), 0, stream, alpha, Z, P, D, X);
  return ac_check_launch("weighted_avg");
}

// -------------------------------------------------------------------- pdist
// Condensed Euclidean distances in float64 (scipy pdist on the float64 copy of X,
// examples/test.py:186-195): d(i,j) = sqrt(sum_k (x_ik - x_jk)^2), differences and
// accumulation in fp64.  32x32 pairs per block, rows staged through LDS in 64-wide K slices.
__global__ __launch_bounds__(256) void pdist_kernel(const float* __restrict__ X, int64_t N, int D,
                                                    double* __restrict__ out) {
  __shared__ float sa[32][65], sb[32][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;                                   // upper triangle of 32x32 tiles
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;                // each thread: rows {ti, ti+16} x cols {tj, tj+16}
  double acc[2][2] = {{0, 0}, {0, 0}};
  for (int k0 = 0; k0 < D; k0 += 64) {
    for (int e = tid; e < 32 * 64; e += 256) {
      int rr = e >> 6, kk = e & 63;
      int64_t ra = (int64_t)bi * 32 + rr, rb = (int64_t)bj * 32 + rr;
      sa[rr][kk] = (ra < N && k0 + kk < D) ? X[ra * D + k0 + kk] : 0.f;
      sb[rr][kk] = (rb < N && k0 + kk < D) ? X[rb * D + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) {
      double a0 = sa[ti][kk], a1 = sa[ti + 16][kk], b0 = sb[tj][kk], b1 = sb[tj + 16][kk];
      double d;
      d = a0 - b0; acc[0][0] = fma(d, d, acc[0][0]);
      d = a0 - b1; acc[0][1] = fma(d, d, acc[0][1]);
      d = a1 - b0; acc[1][0] = fma(d, d, acc[1][0]);
      d = a1 - b1; acc[1][1] = fma(d, d, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      int64_t i = (int64_t)bi * 32 + ti + 16 * u, j = (int64_t)bj * 32 + tj + 16 * v;
      if (i < j && j < N) out[i * N - i * (i + 1) / 2 + (j - i - 1)] = sqrt(acc[u][v]);
    }
}

// Small N (every MVTec category): the 32x32-pair tiles above give only (N/32)^2/2 workgroups - 6 at
// N = 83, 0.45 ms for 14 MFLOP.  Here one WAVE owns a 2x2 block of pairs and strides the D axis with its
// 64 lanes (coalesced 256-byte row segments), fp64 partial sums per lane, fixed-order wave reduction:
// ~850 waves at N = 83, deterministic, rows come from L2.
__global__ __launch_bounds__(256) void pdist_small_kernel(const float* __restrict__ X, int64_t N, int D,
                                                          double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t a = blockIdx.y, b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t T = (N + 1) / 2;
  if (b < a || b >= T) return;                                  // wave-uniform
  const int64_t i0 = 2 * a, i1 = min(2 * a + 1, N - 1), j0 = 2 * b, j1 = min(2 * b + 1, N - 1);
  const float *xi0 = X + i0 * D, *xi1 = X + i1 * D, *xj0 = X + j0 * D, *xj1 = X + j1 * D;
  double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
  for (int k = lane; k < D; k += 64) {
    const double p0 = xi0[k], p1 = xi1[k], q0 = xj0[k], q1 = xj1[k];
    double d;
    d = p0 - q0; s00 = fma(d, d, s00);
    d = p0 - q1; s01 = fma(d, d, s01);
    d = p1 - q0; s10 = fma(d, d, s10);
    d = p1 - q1; s11 = fma(d, d, s11);
  }
  s00 = wave_sum(s00); s01 = wave_sum(s01); s10 = wave_sum(s10); s11 = wave_sum(s11);
  if (lane == 0) {
    auto put = [&](int64_t i, int64_t j, double s) {
      if (i < j && j < N) out[i * N - i * (i + 1) / 2 + (j - i - 1)] = sqrt(s);
    };
    put(i0, j0, s00);
    if (2 * b + 1 < N) put(i0, j1, s01);
    if (2 * a + 1 < N) put(i1, j0, s10);
    if (2 * a + 1 < N && 2 * b + 1 < N) put(i1, j1, s11);
  }
}

extern "C" int ac_pdist(const float* X, int64_t N, int D, double* out, void* stream) {
  AC_REQUIRE(X && (out || N < 2), "ac_pdist: null pointer");
  AC_REQUIRE(D > 0 && N >= 0 && N <= 32 * 65535LL, "ac_pdist: bad shape N=%lld D=%d", (long long)N, D);
  if (N < 2) return AC_OK;
  if (N <= 1024) {
    const unsigned T = (unsigned)((N + 1) / 2);
    hipLaunchKernelGGL(pdist_small_kernel, dim3((T + 3) / 4, T), dim3(256), 0, as_stream(stream), X, N, D, out);
    return ac_check_launch("pdist_small");
  }
  unsigned t = (unsigned)((N + 31) / 32);
  hipLaunchKernelGGL(pdist_kernel, dim3(t, t), dim3(256), 0, as_stream(stream), X, N, D, out);
  return ac_check_launch("pdist");
}

#ifdef AC_G7_TIMING
// Diagnostic builds only (gemm_hl7.h): per persistent workgroup, waves 0 and 4: [tiles, prologue, k loop, vmcnt wait at X,
// lgkmcnt + barrier wait at X, epilogue + next tile id, whole kernel, stages per tile] in shader-clock cycles.
extern "C" int ac_debug_wait_times7(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm7::ac_wait_times7), sizeof(gemm7::ac_wait_times7)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef AC_TILE_TIMING
// Diagnostic builds only (gemm_hl5.h): per-workgroup [k-loop cycles, X wait, Y wait, stages] of the last
// gemm_hl5 launch made from this translation unit.
extern "C" int ac_debug_wait_times(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm5::ac_wait_times), sizeof(gemm5::ac_wait_times)) != hipSuccess) return -1;
  return 0;
}
#endif

#ifdef AC_TILE_TIMING
// Diagnostic builds only (see gemm_hl2.h): the stamp table of THIS translation unit's gemm2 launches (the all-pairs
// kernel).  reset != 0 restarts the launch-slot counter; otherwise copies [8 slots][8192 blocks][4 stamps].
extern "C" int ac_debug_tile_times_alpha(unsigned long long* out, int reset) {
  if (reset) { gemm2::tt_counter() = 0; return 0; }
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm2::ac_tile_times), sizeof(gemm2::ac_tile_times)) != hipSuccess) return -1;
  return gemm2::tt_counter();
}
#endif
