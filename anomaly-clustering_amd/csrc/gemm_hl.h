// Split-fp16 ("hl32") MFMA GEMM core for gfx950:  C[m][n] = sum_k A[m][k] * B[n][k]
//
// Both operands are K-contiguous hl32 matrices (common.h).  Every K-step issues
// three v_mfma_f32_32x32x16_f16 per 32x32 output tile (hi*lo, lo*hi, hi*hi) into
// one fp32 accumulator - an fp32-grade product at a third of the fp16 MFMA rate,
// still ~5x the fp32-MFMA rate (DESIGN.md "Arithmetic").
//
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.
// The wave's M index lives on the accumulator REGISTERS, its N index on the LANES
// (C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// so callers put the axis they want to reduce in-lane on A and the axis they want
// contiguous in memory on B.
//
// Staging: each stage is 128 rows x 128 bytes per operand (one full cache line per
// row: 32 hi + 32 lo halves), register-prefetched one stage ahead and written to a
// double-buffered LDS image [plane hi|lo][128 rows][64 B] whose 16-byte chunks are
// XOR-swizzled with (row>>2)&3 so that every ds_read_b128 lane group hits 16
// distinct slots (MI355X LDS banking: 4 x 16-lane groups, 64 banks).
#pragma once
#include "common.h"
#include <type_traits>

namespace gemm {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE = BM * 64;            // bytes of one [128][64B] plane
constexpr int STAGE = 4 * PLANE;          // A_hi, A_lo, B_hi, B_lo
constexpr int LDS_BYTES = 2 * STAGE;      // 64 KB, double buffered
constexpr int FLUSH_STAGES = 16;          // MFMA accumulator flushed every 16 stages (512 k)

__device__ inline int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// XCD-aware, grouped rasterisation of a 1-D grid onto (tile_m, tile_n).
// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of the
// grouped order so that tiles sharing A/B panels share an L2.
__device__ inline void raster(int id, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  int total = tiles_m * tiles_n;
  int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
  int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  int per_group = group_m * tiles_n;
  int g = lin / per_group, within = lin - g * per_group;
  int first = g * group_m;
  int gsz = min(group_m, tiles_m - first);
  tn = within / gsz;
  tm = first + (within - tn * gsz);
}

// Register image of one wave's 64x64 accumulator, for the two fp16 MFMA shapes.  Epilogues are
// written against this interface (NT tiles per side, NR registers per tile, row()/col() of an element,
// min_over_row_groups()) so the same epilogue serves both kernels:
//   FragT<32>: v_mfma_f32_32x32x16_f16, 2x2 tiles, C/D map col = lane&31, row = (i&3)+8(i>>2)+4(lane>>5)
//   FragT<16>: v_mfma_f32_16x16x32_f16, 4x4 tiles, C/D map col = lane&15, row = 4(lane>>4)+i
// (the chip sustains a ~14 % higher clock on the 16x16x32 shape: csrc/probe/mfma_rate.hip)
template <int SHAPE> struct FragT;
template <> struct FragT<32> {
  static constexpr int NT = 2, NR = 16, TW = 32;
  f32x16 acc[2][2];
  static __device__ inline int row(int mi, int i, int lane) { return mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }
  static __device__ inline int col(int ni, int lane) { return ni * 32 + (lane & 31); }
  // lanes with equal col() hold different rows: combine them (the result is valid in every lane)
  static __device__ inline float min_over_row_groups(float v) { return fminf(v, __shfl_xor(v, 32, 64)); }
  static __device__ inline bool first_row_group(int lane) { return lane < 32; }
};
template <> struct FragT<16> {
  static constexpr int NT = 4, NR = 4, TW = 16;
  f32x4 acc[4][4];
  static __device__ inline int row(int mi, int i, int lane) { return mi * 16 + 4 * (lane >> 4) + i; }
  static __device__ inline int col(int ni, int lane) { return ni * 16 + (lane & 15); }
  static __device__ inline float min_over_row_groups(float v) {
    v = fminf(v, __shfl_xor(v, 16, 64));
    return fminf(v, __shfl_xor(v, 32, 64));
  }
  static __device__ inline bool first_row_group(int lane) { return lane < 16; }
};
using Frag = FragT<32>;

// Epilogues that can exploit C = C^T (A and B are the same matrix) specialise this to true and
// provide `bool symmetric` plus `transposed(frag, m_wave, n_wave, lane, float* lds_scratch)`:
// only tiles with tile_m <= tile_n are computed, and every off-diagonal tile is reduced in both
// directions (the scratch is this wave's 32x65-float slice of the then-idle staging LDS).
// kStaged: the epilogue also provides `staged(const float* tile, m_wave, n_wave, lane)`, which reads the
// wave's 64x64 result back from LDS ([row][EPI_PITCH floats]) one row segment per lane, so that it can
// emit full 128-byte lines with 16-byte stores (gemm_hl2.h uses it; the 64 KB kernel cannot).
template <class Epi> struct EpiTraits { static constexpr bool kSymmetric = false; static constexpr bool kStaged = false; };
constexpr int EPI_PITCH = 68;   // floats per staged row: 16-byte aligned, conflict-free for ds_read_b128

// Swapped orientation (gemm_hl2.h only; an epilogue opts in with `static constexpr bool kSwap = true`
// and a `swapped(frag, m_wave, n_wave, lane)` method).  The MFMA gets the B fragment as its first
// operand, so the accumulator tile is C^T: lane l then holds, of tile (mi, ni), row mi*16 + (l&15) and
// FOUR CONSECUTIVE columns - and because the kernel also permutes which B row feeds which MFMA slot,
// tiles 2p and 2p+1 together give the lane EIGHT consecutive columns: 32p + 8(l>>4) + 0..7 =
// acc[mi][2p][0..3], acc[mi][2p+1][0..3].  A row-major epilogue needs no LDS transpose: fp32 output is two
// 16-byte stores per (mi, p), hl32 output one 16-byte store per plane, and a wave-instruction covers 16
// rows x 64 (hl32) or 128 (fp32) contiguous bytes.
template <class E, class = void> struct HasSwap : std::false_type {};
template <class E> struct HasSwap<E, std::void_t<decltype(E::kSwap)>> : std::bool_constant<E::kSwap> {};
struct SwapFrag {
  __device__ static int row(int mi, int lane) { return mi * 16 + (lane & 15); }
  __device__ static int col8(int p, int lane) { return 32 * p + 8 * (lane >> 4); }
};

// Epi must provide:  __device__ void operator()(const Frag&, int m_wave, int n_wave, int lane) const
//   m_wave / n_wave = first row / column of this wave's 64x64 sub-tile.
// Element (mi, ni, i) of lane l sits at row m_wave + mi*32 + (i&3) + 8*(i>>2) + 4*(l>>5),
// column n_wave + ni*32 + (l&31).
template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm_hl_kernel(const uint8_t* __restrict__ A,
                                                          const uint8_t* __restrict__ B, int M, int N, int K,
                                                          int tiles_m, int tiles_n, int group_m, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  raster(blockIdx.x, tiles_m, tiles_n, group_m, tm, tn);
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      // The grid is the upper triangle (tm <= tn) of a T x T tile matrix, folded into a
      // ceil(T/2) x (T+1) rectangle so every XCD gets the same amount of work: folded row p holds
      // row p (T-p tiles) followed by row T-1-p (p+1 tiles).
      const int T = tiles_n - 1, p = tm, c = tn;
      if (c < T - p) { tm = p; tn = p + c; }
      else {
        if (2 * p == T - 1) return;                // odd T: the middle row is paired with itself
        tm = T - 1 - p; tn = tm + (c - (T - p));
      }
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- global -> register staging map: 8 threads cover one 128-byte row group
  const int lrow = tid >> 3, c8 = tid & 7;
  const size_t ld = (size_t)K * 4;
  const uint8_t* ga[4];
  const uint8_t* gb[4];
  int lds_off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int row = p * 32 + lrow;
    ga[p] = A + (size_t)min(m0 + row, M - 1) * ld + c8 * 16;
    gb[p] = B + (size_t)min(n0 + row, N - 1) * ld + c8 * 16;
    lds_off[p] = (c8 >> 2) * PLANE + swz(row, c8 & 3);
  }

  // Two-level accumulation: the MFMA accumulator `f` only ever holds FLUSH_STAGES*32 k-terms;
  // it is added into `tot` (VALU) and cleared every FLUSH_STAGES stages.  With K = 4096 and
  // strongly correlated rows (a.b ~ 1e3) a single fp32 running sum loses ~2e-3; blocked
  // summation keeps the error at the level of a blocked fp32 CPU sgemm (measured).
  Frag f, tot;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) { f.acc[mi][ni][i] = 0.f; tot.acc[mi][ni][i] = 0.f; }

  u32x4 ra[4], rb[4];
  const int nk = K / BK;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = *reinterpret_cast<const u32x4*>(ga[p]);
    rb[p] = *reinterpret_cast<const u32x4*>(gb[p]);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    *reinterpret_cast<u32x4*>(smem + lds_off[p]) = ra[p];
    *reinterpret_cast<u32x4*>(smem + 2 * PLANE + lds_off[p]) = rb[p];
  }
  __syncthreads();

  // per-lane LDS read offsets (bytes inside a plane) for the two k-steps of a stage
  int a_off[2][2], b_off[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a_off[s][t] = swz(wm * 64 + t * 32 + r, s * 2 + h);
      b_off[s][t] = swz(wn * 64 + t * 32 + r, s * 2 + h);
    }

  for (int kt = 0; kt < nk; ++kt) {
    const uint8_t* cur = smem + (kt & 1) * STAGE;
    const bool more = kt + 1 < nk;
    if (more) {
      const size_t ko = (size_t)(kt + 1) * 128;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = *reinterpret_cast<const u32x4*>(ga[p] + ko);
        rb[p] = *reinterpret_cast<const u32x4*>(gb[p] + ko);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const half8*>(cur + a_off[s][t]);
        al[t] = *reinterpret_cast<const half8*>(cur + PLANE + a_off[s][t]);
        bh[t] = *reinterpret_cast<const half8*>(cur + 2 * PLANE + b_off[s][t]);
        bl[t] = *reinterpret_cast<const half8*>(cur + 3 * PLANE + b_off[s][t]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], f.acc[mi][ni], 0, 0, 0);
          f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], f.acc[mi][ni], 0, 0, 0);
          f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], f.acc[mi][ni], 0, 0, 0);
        }
    }
    if ((kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int i = 0; i < 16; ++i) { tot.acc[mi][ni][i] += f.acc[mi][ni][i]; f.acc[mi][ni][i] = 0.f; }
    }
    if (more) {
      uint8_t* nxt = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<u32x4*>(nxt + lds_off[p]) = ra[p];
        *reinterpret_cast<u32x4*>(nxt + 2 * PLANE + lds_off[p]) = rb[p];
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) tot.acc[mi][ni][i] += f.acc[mi][ni][i];
  epi(tot, m0 + wm * 64, n0 + wn * 64, lane);
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric && tm != tn)                 // all waves are past the loop's last barrier: LDS is free
      epi.transposed(tot, m0 + wm * 64, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65));
  }
}

// Host launcher.  M, N arbitrary (tails are clamped on load; the epilogue masks them), K % 32 == 0.
template <class Epi>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 8) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl: K=%d must be a positive multiple of 32", K);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // folded upper triangle (see the kernel)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl: symmetric epilogue needs M == N");
      const int T = tiles_m;
      tiles_m = (T + 1) / 2;
      tiles_n = T + 1;
    }
  }
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl: grid too large");
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_hl_kernel<Epi>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_hl_kernel<Epi>, dim3((unsigned)total), dim3(256), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, epi);
  return ac_check_launch("gemm_hl");
}

}  // namespace gemm
