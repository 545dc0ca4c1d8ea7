// Shared pieces of the split-fp16 ("hl32") MFMA GEMMs for gfx950:  C[m][n] = sum_k A[m][k] * B[n][k]
//
// Both operands are K-contiguous hl32 matrices (common.h).  Every 32-deep K-step issues three fp16 MFMAs
// per output tile (hi*lo, lo*hi, hi*hi) into one fp32 accumulator - an fp32-grade product at a third of
// the fp16 MFMA rate, still ~5x the fp32-MFMA rate (DESIGN.md "Arithmetic").
//
// This header holds what every kernel generation shares: the XCD-aware rasterisation, the accumulator
// fragment interface the epilogues are written against, and the epilogue traits.  The kernels themselves:
//   gemm_hl2.h   256x128 tile, LDS-DMA ring, staggered schedule, optional two-level accumulation
//   gemm_hl5.h   256x256 tile, continuous MFMA stream (the ViT linear layers)
//   earlier / negative-result generations live outside the package (scratch/gemm_generations/; -DAC_EXPERIMENTAL builds only)
// The wave's M index lives on the accumulator REGISTERS, its N index on the LANES, so callers put the
// axis they want to reduce in-lane on A and the axis they want contiguous in memory on B.
#pragma once
#include "common.h"
#include <type_traits>

namespace gemm {

constexpr int FLUSH_STAGES = 16;          // two-level sum: MFMA accumulator flushed every 16 stages (512 k)

// XCD-aware, grouped rasterisation of a 1-D grid onto (tile_m, tile_n).
// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of the
// grouped order so that tiles sharing A/B panels share an L2.
// Blocks are dealt round-robin to the 8 XCDs (block id & 7 names the XCD, id >> 3 its slot there): position of
// block `id` in the linear tile order.
//   W == 0: every XCD walks ONE contiguous eighth of the order, so the 32 workgroups an XCD runs at a time are 32
//     consecutive tiles and consecutive rounds stay on the same row panels.
//   W > 0: windows of W consecutive tiles are dealt to the XCDs in turn, so at any time the 8 XCDs work on 8 NEIGHBOURING
//     windows (what one XCD misses in its L2 the Infinity Cache may already hold for another); the tail (total % 8W
//     tiles) keeps the identity order.
// Measured on the all-pairs kernel, round 4, interleaved rounds on one box each (profiles/r04_allpairs_xcd_window_*.txt),
// W = 32 against W = 0, register-only epilogue: the TRIANGLE of a symmetric launch +0.7 % (N = 83), +1.5 % (167), 0 (585)
// - its L2 hit rate falls from 77 % to 55 % and the fabric traffic doubles, 94 -> 181 GB per launch - but RECTANGLES
// (two-way blocks: what a rank of a multi-GPU run and the sub-blocks of a large category launch) -3.5 % at 11 x 11 images,
// -1.0 % at 42 x 42, -0.5 % at 400 x 400, config 5b at N = 2000 -1.6 %; W = 16 / 64 / 128 lose everywhere; the ViT linear
// layers do not move.  Hence: rectangles W = 32, the triangle W = 0.  Only the ORDER of the tiles changes: bit-identical.
#ifndef AC_XCD_WINDOW_RECT
#define AC_XCD_WINDOW_RECT 32
#endif
#ifndef AC_XCD_WINDOW_TRI
#define AC_XCD_WINDOW_TRI 0
#endif
template <int W>
__host__ __device__ inline int xcd_lin(int id, long long total) {
  const int xcd = id & 7, slot = id >> 3;
  if constexpr (W > 0) {
    const long long full = total / (8 * W) * (8 * W);
    if (id >= full) return id;
    return ((slot / W) * 8 + xcd) * W + slot % W;
  } else {
    const int q = (int)(total >> 3), r = (int)(total & 7);
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
}

template <int W>
__device__ inline void raster_w(int id, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  int total = tiles_m * tiles_n;
  int lin = xcd_lin<W>(id, total);
  int per_group = group_m * tiles_n;
  int g = lin / per_group, within = lin - g * per_group;
  int first = g * group_m;
  int gsz = min(group_m, tiles_m - first);
  tn = within / gsz;
  tm = first + (within - tn * gsz);
}
__device__ inline void raster(int id, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  raster_w<AC_XCD_WINDOW_RECT>(id, tiles_m, tiles_n, group_m, tm, tn);
}
// the same with the XCD order chosen at run time (gemm_hl7.h): windowed (W = 32) or one contiguous eighth per XCD
__device__ inline void raster_rt(int id, int tiles_m, int tiles_n, int group_m, bool windowed, int& tm, int& tn) {
  if (windowed) raster_w<32>(id, tiles_m, tiles_n, group_m, tm, tn);
  else raster_w<0>(id, tiles_m, tiles_n, group_m, tm, tn);
}

// Register image of one wave's 64x64 accumulator, for the two fp16 MFMA shapes.  Epilogues are
// written against this interface (NT tiles per side, NR registers per tile, row()/col() of an element,
// min_over_row_groups()) so the same epilogue serves both kernels:
//   FragT<32>: v_mfma_f32_32x32x16_f16, 2x2 tiles, C/D map col = lane&31, row = (i&3)+8(i>>2)+4(lane>>5)
//   FragT<16>: v_mfma_f32_16x16x32_f16, 4x4 tiles, C/D map col = lane&15, row = 4(lane>>4)+i
// (the chip sustains a ~14 % higher clock on the 16x16x32 shape: scratch/probe/mfma_rate.hip)
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
// An epilogue may also take the wave's whole accumulator at once (`static constexpr bool kWholeTile = true` and
// `swapped_tile(acc[4][NT], m_wave, n_wave, lane)`; gemm_hl5.h calls it instead of NT/4 swapped() calls) - for
// epilogues that read memory and want their loads in flight across the 64-column fragments.
template <class E, class = void> struct HasWholeTile : std::false_type {};
template <class E> struct HasWholeTile<E, std::void_t<decltype(E::kWholeTile)>> : std::bool_constant<E::kWholeTile> {};
struct SwapFrag {
  __device__ static int row(int mi, int lane) { return mi * 16 + (lane & 15); }
  __device__ static int col8(int p, int lane) { return 32 * p + 8 * (lane >> 4); }
};

}  // namespace gemm
