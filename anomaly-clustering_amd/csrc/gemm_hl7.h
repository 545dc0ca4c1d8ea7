// Split-fp16 MFMA GEMM, 256x192x32 tiles on 8 waves with TWO-LEVEL accumulation: the all-pairs distance kernel (round 5).
//
// Why this shape.  The all-pairs kernel needs a second fp32 accumulator per output element (gemm_hl.h: the MFMA
// accumulator is flushed every 512 k-terms, which is what keeps w at 1.8e-5 from the fp64 truth), so its per-wave state
// is 2 x (rows x columns / 64) registers and gemm_hl5.h's 256x256 tile (2 x 128) cannot hold it.  gemm_hl2.h's 256x128
// tile can (2 x 64), at 128 bytes of LDS-DMA and 341 bytes of fragment reads per MFMA - and round 4's power readings
//  showed that on this chip those BYTES are paid in clock: the DMA stream alone takes the kernel
// from 2.05 to 1.76 GHz.  256x192 is the largest tile whose two accumulator sets still fit 256 registers:
//   wave (wm, wn) of a 4 x 2 grid owns 64 rows x 96 columns = 4 x 6 MFMA tiles of 16x16 (v_mfma_f32_16x16x32_f16)
//   acc 96 + tot 96 registers, the four A (row) fragments of the stage RESIDENT (hi + lo: 32), the B (column) fragment
//   STREAMED one column step ahead (hi + lo, double buffered: 16)                                   = 240 of 256
//   -> 99.6 bytes of DMA (-22 %) and 284 bytes of fragment reads (-17 %) per MFMA.
// (gemm_hl5.h with NT = 6 and its B fragments resident is the same tile at 256 registers + 134 spilled: the loop had to
// be turned round.)
//
// Schedule (gemm_hl5.h's, with the operands' roles exchanged): a wave runs one continuous stream of COLUMN steps (4 row
// tiles x 3 passes = 12 MFMAs each); the next B fragment is read during the current step, and in the LAST step of a
// stage every A fragment is re-read for the next stage right after its final MFMA.  LDS 136 KB: A (256 rows, 32 KB per
// stage) in a 2-slot ring - it goes to registers a stage ahead - and B (192 rows, 24 KB) in a 3-slot ring, filled by
// LDS-DMA from waves 0-3 (round 6: buffer loads with scalar burst / stage offsets, see the DMA map in the kernel).  ONE barrier per stage t, X(t), before the last column step: everybody's share of
// A(t+1), B(t+1) has landed; nobody reads B(t) any more and A(t) went to registers a stage ago
//   -> A(t+2) goes into A(t)'s slot (one stage of cover), B(t+3) into B(t)'s slot (two stages).
// Tiles: persistent workgroups pull them from gemm_hl2.h's per-XCD-slot queues (dynamic), or one workgroup per tile.
// Between two tiles of a persistent workgroup (stamps: scratch/wait_timing_g7.py, profiles/r05_allpairs_g7_timing.txt):
//   * the ticket of the next tile is pulled three stages before the k loop ends and resolved right after it, so ONE
//     barrier (the one that retires the ring) also hands over the tile id;
//   * a tile whose eight 64 x 96 blocks all reduce from registers (inside the problem, off the diagonal: 98 % of them at
//     N = 83) gets its squared norms by DMA into a small LDS buffer together with its first stage, so its epilogue issues no
//     global load, and the DMA-issuing waves start the next tile's stream right behind their own reductions, without a
//     barrier;
//   * the group of the previous tile is cached, so the scalar tile arithmetic between two tiles is a subtraction and a
//     division instead of a square root and a search (6.9k -> 2.1k cycles per tile).
//
// Arithmetic: per output element the same chain as gemm_hl2.h's FLUSH kernel - stages in k order, passes hi*lo, lo*hi,
// hi*hi, the MFMA accumulator added into the second one every 16 stages - so the two kernels agree to the bit
// (tests/test_gemm_generations_gpu.py::test_all_pairs_tiles_agree_bitwise, scratch/ab_pairmin.py).
//
// Measured (MI355X, N = 83, same box, interleaved; profiles/r05_allpairs_256x192_ab.txt): -3.6 ... -4.7 % against the
// 256x128 kernel; fabric traffic 88 -> 67 ... 75 GB per launch, MFMA-pipe occupancy 73.5 -> 76.5 ... 78.5 % at the same or a
// slightly LOWER clock (the package stays at its power cap: the joules the operand stream no longer takes go into more
// MFMAs per second, not into clock - and every cycle the tile transition gave up after the first build came back as
// clock: early ticket, LDS norms, cached tile arithmetic together -3 % cycles, -0.3 % time).  Not kept
// (scratch/exp_g7_stagger_and_knobs.patch, scratch/exp_g7_early_prologue.patch): a half-stage stagger of waves 4-7
// (+10 ... +18 %), DMA issue from all eight waves (+8 %), priority variants (0 ... +1 %), the next tile's first stages
// issued before the epilogue (equal time, +19 % fabric traffic).
// Round 6 (profiles/r06_allpairs_buffer_dma_ab.txt): the transfers as buffer loads, 83 -> 16 non-MFMA VALU per stage in the
// loader waves, -0.9 ... -1.2 % same-box; what the DMA path costs, taken apart by the AC_ABLATE7 bits 8 / 16 / 32: instructions
// 0.7 ms, L2 -> LDS bandwidth 0.8, L2 misses 3.2, fresh operands under the MFMAs 1.6.  MT = 8 (four waves, 512 registers):
// hipcc spills an accumulator set in the k loop (profiles/r06_allpairs_512reg_resource_report.txt) - compile-only.
#pragma once
#include "gemm_hl2.h"

namespace gemm7 {

using gemm::EpiTraits;
using gemm2::gptr_t;
using gemm2::lptr_t;
using gemm2::tile_off;
using Frag = gemm::FragT<16>;

constexpr int BM = 256, BN = 192, BK = 32;
// MT = 16-row MFMA tiles per wave: 4 -> eight waves of 64 x 96 (two per SIMD, 256 registers each: the product);
// 8 -> FOUR waves of 128 x 96, one per SIMD with the whole 512-register file (round 6, VERDICT r05 item 2: -32 % fragment
// bytes per MFMA; -DAC_G7_MT=8 builds it).
constexpr int threads_of(int MT) { return 2048 / MT; }
constexpr int NTN = 6;                                   // 16-column MFMA tiles per wave (4 row tiles always)
constexpr int A_HALF = BM * 128, B_HALF = BN * 128;      // 32 KB, 24 KB per stage
constexpr int A_SLOTS = 2, B_SLOTS = 3;
constexpr int LDS_BYTES = A_SLOTS * A_HALF + B_SLOTS * B_HALF;   // 136 KB
// behind the ring: two buffers for the squared norms of a tile's 256 rows + 192 columns (the epilogue of a tile whose
// blocks all reduce from registers reads them from LDS; they arrive with the tile's first stage), then the next tile id
constexpr int NORM_BYTES = 2048, LDS_NORMS = 2 * NORM_BYTES;
constexpr int LDS_EXTRA = LDS_NORMS + 16;
constexpr int LDS_ABLATE = 4096;              // AC_ABLATE7 & 32 only
constexpr int FLUSH_STAGES = gemm::FLUSH_STAGES;

// XCD order of a RECTANGLE's tiles (two-way and one-way blocks; gemm_hl.h xcd_lin): windows of 32 tiles dealt to the XCDs in turn
// (the 8 XCDs side by side on neighbouring windows) for SMALL rectangles, one contiguous eighth of the grouped order per XCD
// for LARGE ones.  Round 4 had measured the windows best at every size for the 256x128 kernel.  For this kernel (round 5,
// profiles/r05_allpairs_rect_xcd_order_ab.txt, two-way blocks of n x n images, same box): windows -3.4 % at 11 x 11, -2.4 % at
// 21 x 21, -0.5 % at 42 x 42 - and +2.4 % at 100 x 100, +3.4 % at 300 x 300, where the L2 hit rate of the launch is 35 % with
// windows and 79 % without (fabric traffic 596 against 189 GB at 100 x 100).  The launcher switches at 160 tiles per CU; it
// travels in bit 16 of the group argument.  (-DAC_G7_WINDOW_RECT=0 / 32 pins one order for A/B builds.)
constexpr int RECT_WINDOW_TILES_PER_CU = 160;
// Timing-only diagnostic builds (results are wrong): -DAC_ABLATE7=1 no DMA in the k loop, 2 no fragment reads in the
// k loop, 4 no MFMAs; round 6, on the DMA itself: 8 every transfer issued against a descriptor of zero records (nothing
// fetched; zeros land in LDS), 16 every stage fetches the tile's first k slab (all bytes move, out of the L2), 32 the
// transfers land in 4 KB of LDS behind the rings (combinable).  0 in every shipped build.
#ifndef AC_ABLATE7
#define AC_ABLATE7 0
#endif
#ifndef AC_G7_FLAT_DMA
#define AC_G7_FLAT_DMA 0
#endif
#ifndef AC_G7_PIN_AGPR
#define AC_G7_PIN_AGPR 0
#endif
#ifndef AC_G7_KATTR
#define AC_G7_KATTR
#endif

// -DAC_G7_TIMING (diagnostic builds; read back with ac_debug_wait_times7, alpha.hip): waves 0 and 4 of every persistent
// workgroup add up the shader-clock cycles they spend in the prologue of a tile, in its k loop, inside that at X waiting
// for their own DMA (vmcnt) and then for the other waves (lgkmcnt + barrier), and in the epilogue up to the next tile id.
#ifdef AC_G7_TIMING
static __device__ unsigned long long ac_wait_times7[1024 * 2 * 8];
#define G7_T0() const unsigned long long t0_ = __builtin_readcyclecounter()
#define G7_TACC(v) v += __builtin_readcyclecounter() - t0_
#else
#define G7_T0()
#define G7_TACC(v)
#endif

// ---- upper triangle of a symmetric problem on 256-row x 192-column tiles.  Tile (p, c) holds rows [256 p, +256) and
// columns [192 c, +192); it contains an element with column >= row iff c >= c0(p) = floor(4 p / 3).
__host__ __device__ inline int tri_c0(int p) { return (4 * p) / 3; }
// tiles in the tile rows above p: p Tn - sum_{i < p} floor(4 i / 3); with p = 3 a + b the sum is 6 a^2 - 3 a + 4 a b + [b == 2].
// 32-bit: the launcher refuses problems with Tm * Tn >= 2^31.
__host__ __device__ inline int tri_rows_before(int p, int Tn) {
  const int a = p / 3, b = p - 3 * a;
  return p * Tn - (6 * a * a - 3 * a + 4 * a * b + (b == 2 ? 1 : 0));
}
// Linear order made for the L2, as gemm_hl2.h's: groups of G tile rows; inside a group the ACTUAL column runs slowest
// and the row fastest, so that a contiguous run of the order (what one XCD works on at a time) covers G row panels and
// the few column panels they share.
// This runs once per tile on the scalar unit, between two tiles of a persistent workgroup, with the MFMA pipe idle -
// and hipcc keeps ~150 scalar registers of this kernel spilled in vector lanes, so every operand costs a v_readlane.
// The group of the PREVIOUS tile is therefore remembered (TriCache): a workgroup's next tile is ~32 positions further on
// in the same XCD's run and nearly always in the same group of ~G Tn tiles, where the answer is one subtraction and
// one division by the group's row count; the square root and the search run on a group change only.
struct TriCache {
  int lo = 0, n = 0;            // linear ids [lo, lo + n) are the cached group's tiles (n = 0: nothing cached)
  int p0 = 0, gsz = 1;          // its first tile row, its rows
  int c_head = 0, c_full = 0;   // its first column; from c_full on every row of the group is valid
  int head = 0;                 // tiles in the ragged head (columns c_head .. c_full - 1)
  float inv = 1.f;              // 1 / gsz
};
__host__ __device__ inline void tri_tile(int lin, int Tm, int Tn, int G, int& tm, int& tn, TriCache& tc) {
  if ((unsigned)(lin - tc.lo) >= (unsigned)tc.n) {
    // row p with F(p) <= lin < F(p + 1): F(p) = p (Tn + 1) - (2/3) p^2 - {0, 1/3, 1/3}[p % 3], so the root of the
    // quadratic is within a row of the answer (single precision is enough: the loops below settle the last step)
    const float t = (float)(Tn + 1);
    int p = (int)(0.75f * (t - sqrtf(fmaxf(t * t - (8.0f / 3.0f) * (float)lin, 0.0f))));
    p = max(0, min(p, Tm - 1));
    while (p > 0 && tri_rows_before(p, Tn) > lin) --p;
    while (p + 1 < Tm && tri_rows_before(p + 1, Tn) <= lin) ++p;
    tc.p0 = p - p % G;
    tc.gsz = min(G, Tm - tc.p0);
    tc.lo = tri_rows_before(tc.p0, Tn);
    tc.n = tri_rows_before(tc.p0 + tc.gsz, Tn) - tc.lo;
    tc.c_head = tri_c0(tc.p0);
    tc.c_full = tri_c0(tc.p0 + tc.gsz - 1);
    tc.head = 0;
    for (int c = tc.c_head; c < tc.c_full; ++c) tc.head += min(tc.gsz, (3 * c + 2) / 4 - tc.p0 + 1);   // <= 4 (G - 1) / 3 + 1 columns
    tc.inv = 1.0f / (float)tc.gsz;
  }
  int w = lin - tc.lo;                                   // index inside the group
  if (w >= tc.head) {                                    // full columns: gsz tiles each
    w -= tc.head;
    const int q = (int)(((float)w + 0.5f) * tc.inv);     // w / gsz, exact: gsz <= G is small, w < 2^22
    tn = tc.c_full + q;
    tm = tc.p0 + (w - q * tc.gsz);
    return;
  }
  int c = tc.c_head;
  for (;;) {                                             // ragged head: column c holds the rows p0 .. with c0(row) <= c
    const int v = min(tc.gsz, (3 * c + 2) / 4 - tc.p0 + 1);
    if (w < v) break;
    w -= v;
    ++c;
  }
  tm = tc.p0 + w;
  tn = c;
}

// `total`: the launch's tile count (the launcher passes it: tri_rows_before(Tm, Tn) for a symmetric problem)
template <class Epi>
__device__ inline void tile_of(int id, int total, int tiles_m, int tiles_n, int group_m, const Epi& epi, int& tm, int& tn,
                               TriCache& tc) {
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      tri_tile(gemm2::xcd_chunk_lin(id, total), tiles_m, tiles_n, group_m & 0xffff, tm, tn, tc);
      return;
    }
  }
  gemm::raster_rt(id, tiles_m, tiles_n, group_m & 0xffff, (group_m >> 16) & 1, tm, tn);
}

template <class Epi, int MT = 4>
__global__ __launch_bounds__(threads_of(MT), MT == 4 ? 2 : 1) AC_G7_KATTR void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int id_lo, int id_hi, int total, Epi epi, unsigned* queue) {
  static_assert(LDS_BYTES + LDS_EXTRA <= 160 * 1024, "LDS");
  static_assert(!gemm::HasSwap<Epi>::value && EpiTraits<Epi>::kSymmetric, "gemm_hl7 serves the pair-min epilogue");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const Aring = smem;
  uint8_t* const Bring = smem + A_SLOTS * A_HALF;
  int* const lds_next = reinterpret_cast<int*>(smem + LDS_BYTES + LDS_NORMS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;       // MT = 4: 4 x 2 waves, MT = 8: 2 x 2
  constexpr int WROWS = 16 * MT;                 // rows of a wave's block
  const int nk = K / BK;

  // DMA map (gemm_hl5.h's): a half-tile is 32 (A) / 24 (B) wave-instructions of 1 KB = 8 rows each, issued by waves
  // 0-3 only - the SIMD's arbiter serves its older wave first, so their issue stalls hide behind the partner's MFMAs.
  // Wave w issues rows 32 j + 8 w + (l >> 3); lane l fetches logical chunk (l & 7) ^ ((row >> 1) & 7) of its row.
  // Round 6: the transfers are BUFFER loads (buffer_load_dwordx4 ... offen lds).  One descriptor per operand and tile
  // (base = the tile's first row, num_records = the bytes of its rows inside the problem: the range check drops the
  // rows behind the edge - they used to be clamped to the last row, and the epilogue never looks at them either way),
  // ONE per-lane byte offset for the whole launch (row 8 w + (l >> 3) of a burst, swizzled chunk: the same for both
  // operands, since both have K * 4 bytes per row), the burst (32 j rows) and the stage (128 t bytes) in the scalar
  // offset: no vector instruction per transfer.  (Rounds 1-5: global_load_lds with a 64-bit address per lane and
  // transfer, rebuilt at every burst because hoisted it costs 28 registers - 76 VALU per 64 MFMAs in the loader waves'
  // stage block, v_mad_u64_u32 among them.  -DAC_G7_FLAT_DMA=1 builds that form for A/B.)
  constexpr int LW = 4;                          // DMA-issuing waves (all eight: +7 %, round-5 A/B)
  const bool loader = MT == 8 || wave < LW;      // (four waves: every one of them)
  if (MT == 4 && loader) __builtin_amdgcn_s_setprio(2);
  const unsigned ld32 = (unsigned)K * 4u;
  int m0 = 0, n0 = 0;
#if AC_G7_FLAT_DMA
  auto dma = [&](const uint8_t* src, int first, int last, int t, uint8_t* dst, int pieces) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int r = (wave & (LW - 1)) * 8 + (l >> 3);
    const uint8_t* base = src + (size_t)t * 128 + (unsigned)((l & 7) ^ ((r >> 1) & 7)) * 16u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j >= pieces) break;
      const unsigned row = (unsigned)min(first + r + 8 * LW * j, last);
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (uint64_t)row * ld32), (lptr_t)(dst + (j * LW + (wave & (LW - 1))) * 1024), 16, 0, 0);
    }
  };
  auto dmaA = [&](int t, int slot) { if (loader) dma(A, m0, M - 1, t, Aring + slot * A_HALF, BM / (8 * LW)); };
  auto dmaB = [&](int t, int slot) { if (loader) dma(B, n0, N - 1, t, Bring + slot * B_HALF, BN / (8 * LW)); };
  auto set_tile = [&]() {};
#else
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  const int dma_r = (wave & (LW - 1)) * 8 + (lane >> 3);
  const int dma_voff = dma_r * (int)ld32 + ((lane & 7) ^ ((dma_r >> 1) & 7)) * 16;     // < 32 rows: K < 2^14 keeps it in 31 bits
  // The descriptor of an operand and tile as three scalars (base, bytes); dma() builds the 128-bit value from them
  // behind readfirstlane, so that hipcc can PROVE it wave-uniform - carried round the tile loop as a descriptor it
  // lands in vector registers and every transfer is wrapped in a waterfall loop (v_readfirstlane x 4, compare, saveexec).
  struct TileBuf { unsigned lo, hi, bytes; };
  TileBuf tbA = {0, 0, 0}, tbB = {0, 0, 0};
  auto tile_buf = [&](const uint8_t* src, int first, int rows_total, int rows_tile) {
    const int rows = max(0, min(rows_total - first, rows_tile));
    const uint64_t base = reinterpret_cast<uint64_t>(src) + (uint64_t)(unsigned)first * ld32;
    TileBuf b = {(unsigned)base, (unsigned)(base >> 32), (unsigned)rows * ld32};
    if (AC_ABLATE7 & 8) b.bytes = 0;             // timing only: every transfer issued, none fetched
    return b;
  };
  auto set_tile = [&]() {                        // after m0 / n0 changed, before the next tile's first transfer
    tbA = tile_buf(A, m0, M, BM);
    tbB = tile_buf(B, n0, N, BN);
  };
  auto dma = [&](const TileBuf& tb, int t, uint8_t* dst, int pieces) {
    if (AC_ABLATE7 & 16) t = 0;                  // timing only: every stage fetches the tile's first k slab (L2 hits)
    const uint64_t base = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)tb.lo) |
                          (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)tb.hi) << 32;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0,
                                                        __builtin_amdgcn_readfirstlane((int)tb.bytes), 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j >= pieces) break;
      // (AC_ABLATE7 & 32, timing only: every transfer lands in 4 KB behind everything else - the rings keep the data of
      // the launch's first prologue, so the MFMAs go on multiplying real numbers)
      uint8_t* const d = (AC_ABLATE7 & 32) ? smem + LDS_BYTES + LDS_EXTRA + (wave & (LW - 1)) * 1024
                                           : dst + (j * LW + (wave & (LW - 1))) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)d, 16, dma_voff, (int)(8u * LW * j * ld32) + t * 128, 0, 0);
    }
  };
  auto dmaA = [&](int t, int slot) { if (loader) dma(tbA, t, Aring + slot * A_HALF, BM / (8 * LW)); };
  auto dmaB = [&](int t, int slot) { if (loader) dma(tbB, t, Bring + slot * B_HALF, BN / (8 * LW)); };
#endif
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // fragment offsets: 16-row tile t of this wave's rows / columns is base + t * 2048 bytes (the XOR swizzle term
  // (row >> 1) & 7 is unchanged by +16 rows; 96 wn is a multiple of 16 too); the lo plane of a row is its hi plane's
  // chunk + 4, i.e. the same offset with bit 6 flipped (slot bases are multiples of 128)
  const int a_hi = tile_off(wm * WROWS + (lane & 15), (lane >> 4));
  const int b_hi = tile_off(wn * (NTN * 16) + (lane & 15), (lane >> 4));
  auto frag = [&](int slot_off, int off_hi, int t, half8& hi, half8& lo) {      // slot_off: byte offset of the ring slot in smem
    const int oh = slot_off + off_hi, ol = oh ^ 64;      // t * 2048 stays an immediate offset of the ds_read
    lo = *reinterpret_cast<const half8*>(smem + ol + t * 2048);   // lo first: the first pass of a step needs only it
    hi = *reinterpret_cast<const half8*>(smem + oh + t * 2048);
  };
  constexpr int B0 = A_SLOTS * A_HALF;           // the B ring's offset in smem

  // dynamic walk (queue != nullptr): gemm_hl2.h's per-XCD-slot queues with stealing; thread 0 pulls, LDS hands over
  auto take = [&]() -> unsigned { return gemm2::queue_take(queue); };
  auto resolve = [&](unsigned ticket) { *lds_next = gemm2::queue_resolve(queue, ticket, id_lo, id_hi); };

  int id, tm = 0, tn = 0;
  TriCache tcache;
  if (queue) {
    if (tid == 0) resolve(take());
    __syncthreads();
    id = __builtin_amdgcn_readfirstlane(*lds_next);   // block-uniform, and known to be: the tile arithmetic stays scalar
    __syncthreads();                               // everyone has read it before thread 0 may overwrite it
    if (id < 0) return;
  } else {
    id = id_lo + blockIdx.x;
    if (id >= id_hi) return;
  }

#ifdef AC_G7_TIMING
  unsigned long long tt_pro = 0, tt_loop = 0, tt_vm = 0, tt_bar = 0, tt_epi = 0, tt_tiles = 0, tt_e1 = 0, tt_e2 = 0;
  const unsigned long long tt_start = __builtin_readcyclecounter();
#endif
  // The first stages of a tile - A(0) B(0) A(1) B(1) B(2), in that order - go out as soon as the wave has finished the
  // epilogue of the tile in front of it.  A "fast" tile (every 64 x 96 block inside the problem and off the diagonal: its
  // epilogue works from registers) gets its squared norms the same way, FIRST in the queue (waves 0 and 1, one DMA
  // instruction each: 256 row norms, 192 column norms) into one of two small LDS buffers: its epilogue then issues no
  // global load, and behind a fast tile the DMA-issuing waves need no barrier before they start the next tile's stream.
  // (Measured and NOT kept, round 5: issuing these transfers BEFORE the epilogue - the cold start of a tile then runs under
  // the reductions of the tile before it, -1.2 % cycles per tile - raises the L2-miss traffic of the launch from 75 to 89 GB
  // (the L2 hit rate falls from 76.8 to 72.1 %: the first stages are fetched out of step with the tiles that share their
  // panels), the clock follows it down (1594 -> 1581 MHz) and the time does not move: profiles/r05_allpairs_early_prologue_pmc.txt.)
  bool fast_next = false;      // the tile whose prologue was issued last is fast (its norms are in LDS buffer norm_par)
  int norm_par = 1;
  auto issue_prologue = [&]() {
    fast_next = epi.tile_is_fast(m0, n0, BM, BN, NTN * 16);
    norm_par ^= 1;
    set_tile();
    if (fast_next && wave < 2) {
      const float* src = wave == 0 ? epi.key_sq + m0 : epi.qry_sq + n0;
      const int l4 = wave == 0 || lane < BN / 4 ? lane * 4 : 0;           // column norms: 48 lanes carry data
      __builtin_amdgcn_global_load_lds((gptr_t)(src + l4), (lptr_t)(smem + LDS_BYTES + norm_par * NORM_BYTES + wave * 1024), 16, 0, 0);
    }
    dmaA(0, 0);
    dmaB(0, 0);
    if (nk > 1) { dmaA(1, 1); dmaB(1, 1); }
    if (nk > 2) dmaB(2, 2);
  };
  tile_of(id, total, tiles_m, tiles_n, group_m, epi, tm, tn, tcache);            // block-uniform
  m0 = tm * BM;
  n0 = tn * BN;
  issue_prologue();
  for (;;) {
#ifdef AC_G7_TIMING
    const unsigned long long tt_a = __builtin_readcyclecounter();
#endif
    const int m0c = m0, n0c = n0;                                 // this tile (m0, n0 move on to the next one before its epilogue)
    const bool fast_cur = fast_next;                              // ... is fast, its norms are in LDS buffer norm_cur
    const int norm_cur = norm_par;
    f32x4 acc[MT][NTN], tot[MT][NTN];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTN; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[mi][ni][i] = 0.f; tot[mi][ni][i] = 0.f; }

    // the first two of the five prologue transfers must have landed (only the waves that issued them wait)
    if (loader) {
      if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BM / (8 * LW) + 2 * (BN / (8 * LW))) : "memory");      // A(1) B(1) B(2) may fly
      else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BM / (8 * LW) + BN / (8 * LW)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    half8 ah[MT], al[MT], bh, bl;
#pragma unroll
    for (int t = 0; t < MT; ++t) frag(0, a_hi, t, ah[t], al[t]);
    frag(B0, b_hi, 0, bh, bl);
#ifdef AC_G7_TIMING
    const unsigned long long tt_b = __builtin_readcyclecounter();
    tt_pro += tt_b - tt_a;
#endif

    int sb = 0;                                    // B slot of the current stage (kt % B_SLOTS)
    auto stage = [&](int kt, auto steady) {
      constexpr bool STEADY = decltype(steady)::value;   // branch-free body while three more stages follow
      const bool more1 = STEADY || kt + 1 < nk, more2 = STEADY || kt + 2 < nk, more3 = STEADY || kt + 3 < nk;
      const int sb1 = sb == B_SLOTS - 1 ? 0 : sb + 1;    // (kt + 1) % B_SLOTS
#if AC_G7_PIN_AGPR                                   // (MT = 8 experiments: say where the MFMA accumulators live)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTN; ++ni) asm volatile("" : "+a"(acc[mi][ni]));
#endif
      const int Bs = B0 + sb * B_HALF, Bn = B0 + sb1 * B_HALF, An = ((kt + 1) & 1) * A_HALF;
      // One column step = three passes over the four row tiles (per element the passes stay in the order hi*lo, lo*hi,
      // hi*hi).  The lo half of the B fragment is dead after the first pass: the next step's fragment is read right
      // there - its lo half into the registers just freed, its hi half into the set the step after next will free - and
      // has the other two passes (8 MFMAs) to arrive.  sched_barrier pins that order: left alone, hipcc sinks the reads
      // to the end of the step to save registers and waits for them at once.
#pragma unroll
      for (int ni = 0; ni < NTN; ++ni) {
        const bool last = ni == NTN - 1;
        if (!(AC_ABLATE7 & 4)) {
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (last && more1) {
          // own share of A(kt+1), B(kt+1) landed: B(kt+2) (6 instructions, the youngest) may still fly
          G7_T0();
          if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BN / (8 * LW)) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          G7_TACC(tt_vm);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          barrier();                               // X(kt)
          G7_TACC(tt_bar);
        }
        half8 nh = bh;
        if (AC_ABLATE7 & 2) {
        } else if (!last) {
          frag(Bs, b_hi, ni + 1, nh, bl);
        } else if (more1) {
          frag(Bn, b_hi, 0, nh, bl);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (last && more1) {
          if (more2 && !(AC_ABLATE7 & 1)) dmaA(kt + 2, kt & 1);   // A(kt) went to registers a stage ago
          if (more3 && !(AC_ABLATE7 & 1)) dmaB(kt + 3, sb);       // nobody reads B(kt) any more
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          if (!(AC_ABLATE7 & 4)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh, acc[mi][ni], 0, 0, 0);
          if (last && more1 && !(AC_ABLATE7 & 2))
            al[mi] = *reinterpret_cast<const half8*>(smem + ((An + a_hi) ^ 64) + mi * 2048);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          if (!(AC_ABLATE7 & 4)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
          if (last && more1 && !(AC_ABLATE7 & 2)) ah[mi] = *reinterpret_cast<const half8*>(smem + (An + a_hi) + mi * 2048);
        }
        bh = nh;
        __builtin_amdgcn_sched_barrier(0);
      }
      // Two-level accumulation (gemm_hl.h): flush the MFMA accumulator every 512 k-terms.
      if ((kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NTN; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) { tot[mi][ni][i] += acc[mi][ni][i]; acc[mi][ni][i] = 0.f; }
      }
      sb = sb1;
    };
    // k loop.  Dynamic walk: thread 0 pulls the next tile's ticket three stages before the end and turns it into a
    // tile id after the last stage, so the round trip of the atomic hides under MFMA work and the id is in LDS when the
    // post-loop barrier opens - one barrier hands over the tile.
    int kt = 0;
    for (; kt + 3 < nk; ++kt) stage(kt, std::true_type{});
    unsigned ticket = 0;
    if (queue && tid == 0) ticket = take();
    for (; kt < nk; ++kt) stage(kt, std::false_type{});
    // EVERY lane consumes the ticket register here: hipcc's wait-count pass merges paths, and with the atomic and its
    // use in two separate `tid == 0` branches it would otherwise believe the register still pending all round the
    // tile loop and put an s_waitcnt vmcnt(0) in front of every reuse of that register - between the DMA bursts of the
    // next tile's prologue, for one.
    asm volatile("" : "+v"(ticket));
    if (queue && tid == 0) resolve(ticket);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();                                     // ring reads done: an epilogue / the next prologue may reuse LDS
#ifdef AC_G7_TIMING
    const unsigned long long tt_c = __builtin_readcyclecounter();
    tt_loop += tt_c - tt_b;
    ++tt_tiles;
#endif
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTN; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mi][ni][i] += tot[mi][ni][i];
    const int n_wave = n0c + wn * (NTN * 16);
    int next = -1;
    if (queue) {
      next = __builtin_amdgcn_readfirstlane(*lds_next);        // block-uniform: the tile arithmetic stays scalar
      if (next >= 0) {
        tile_of(next, total, tiles_m, tiles_n, group_m, epi, tm, tn, tcache);
        m0 = tm * BM;
        n0 = tn * BN;
      }
    }
#ifdef AC_G7_TIMING
    const unsigned long long tt_d = __builtin_readcyclecounter();
    tt_e1 += tt_d - tt_c;
#endif
#ifdef AC_G7_TIMING
    const unsigned long long tt_e = __builtin_readcyclecounter();
    tt_e2 += tt_e - tt_d;
#endif
    // ---- epilogue: the wave's block as MT / 4 pieces of 64 x 96, at rows m_wave, columns n_wave
    const bool sym = epi.symmetric;
    auto epilogue_half = [&](auto hm_c) {
      constexpr int hm = decltype(hm_c)::value;
      f32x4 (&acch)[4][NTN] = *reinterpret_cast<f32x4 (*)[4][NTN]>(&acc[4 * hm]);
      const int m_wave = m0c + wm * WROWS + 64 * hm;
      // symmetric problem: only elements with row <= column are used, each reduced in both directions (gemm_hl2.h).
      // A block wholly above the diagonal and inside the problem: both reductions from registers (epi.fast_wide);
      // otherwise the general code on a 64 x 64 and a 64 x 32 piece, per-element diagonal mask where a piece crosses it.
      const bool below = sym && m_wave > n_wave + (NTN * 16 - 1);
      if (fast_cur) {
        const float* nb = reinterpret_cast<const float*>(smem + LDS_BYTES + norm_cur * NORM_BYTES);
        f32x4 ks[4];
        float qs[NTN];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ks[mi] = *reinterpret_cast<const f32x4*>(nb + wm * WROWS + 64 * hm + 16 * mi + 4 * (lane >> 4));
#pragma unroll
        for (int ni = 0; ni < NTN; ++ni) qs[ni] = nb[256 + wn * (NTN * 16) + 16 * ni + (lane & 15)];
        epi.template fast_reduce<NTN>(acch, ks, qs, m_wave, n_wave, lane, sym || epi.two_way);
      } else if (!below) {
        const bool both = sym || epi.two_way;
        const bool clear = !sym || m_wave + 63 <= n_wave;
        if (!(clear && epi.template fast_wide<NTN>(acch, m_wave, n_wave, lane, both))) {
          float* scratch = reinterpret_cast<float*>(smem) + wave * (32 * 65);
#pragma unroll
          for (int h = 0; h < (NTN + 3) / 4; ++h) {
            Frag f;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int i = 0; i < 4; ++i) f.acc[mi][ni][i] = (4 * h + ni < NTN) ? acch[mi][(4 * h + ni) % NTN][i] : 0.f;
            const int n_sub = n_wave + 64 * h, n_lim = n_wave + NTN * 16, width = min(64, NTN * 16 - 64 * h);
            if (sym && m_wave > n_sub + width - 1) continue;          // this piece lies below the diagonal
            const bool cross = sym && m_wave + 63 > n_sub;           // some of its elements do
            epi(f, m_wave, n_sub, lane, cross, n_lim);
            if (both) epi.transposed(f, m_wave, n_sub, lane, scratch, cross, n_lim);
          }
        }
      }
    };
    epilogue_half(std::integral_constant<int, 0>{});
    if constexpr (MT == 8) epilogue_half(std::integral_constant<int, 1>{});

#ifdef AC_G7_TIMING
    tt_epi += __builtin_readcyclecounter() - tt_c;
    if (next < 0 && (tid == 0 || tid == threads_of(MT) / 2) && blockIdx.x < 1024) {
      unsigned long long* o = ac_wait_times7 + (blockIdx.x * 2 + (tid != 0)) * 8;
      o[0] = tt_tiles; o[1] = tt_pro; o[2] = tt_loop; o[3] = tt_vm; o[4] = tt_bar - tt_vm; o[5] = tt_epi;
      o[6] = __builtin_readcyclecounter() - tt_start; o[7] = nk | (tt_e1 / (tt_tiles ? tt_tiles : 1)) << 16 | (tt_e2 / (tt_tiles ? tt_tiles : 1)) << 40;
    }
#endif
    if (next < 0) break;                           // every queue is exhausted (or: one workgroup per tile)
    if (!fast_cur) {                               // the general epilogue used the ring as scratch: everyone must be out of it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
    }
    issue_prologue();
    id = next;
  }
}

// Host launcher: gemm_hl2.h's contract (M, N arbitrary, K % 32 == 0; part / nparts slices of the tile-id space).
template <class Epi, int MT = 4>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 6, int part = 0, int nparts = 1, bool dynamic = true) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl7: K=%d must be a positive multiple of 32", K);
  if (K > (1 << 20)) return ac_fail(AC_EINVAL, "gemm_hl7: K=%d too large (a tile's rows must fit a 31-bit buffer offset)", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl7: bad part %d/%d", part, nparts);
  if (group_m < 1 || group_m > 0xffff) return ac_fail(AC_EINVAL, "gemm_hl7: group_m=%d out of range", group_m);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl7: grid too large");
  if (epi.symmetric) {
    if (M != N) return ac_fail(AC_EINVAL, "gemm_hl7: symmetric epilogue needs M == N");
    total = tri_rows_before(tiles_m, tiles_n);
  }
  static unsigned long long attr_done = 0;       // per instantiation, one bit per device
  if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&kernel<Epi, MT>), LDS_BYTES + LDS_EXTRA + ((AC_ABLATE7 & 32) ? LDS_ABLATE : 0), &attr_done)) return rc;
  const long long lo = (total * part / nparts) & ~7LL;
  const long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  const int n_cu = ac_device_cus();
  long long grid = hi - lo;
  unsigned* queue = nullptr;
  if (dynamic && hi - lo >= 4ll * n_cu) {        // dynamic walk: worth it (and only meaningful) when every CU gets several tiles
    if (int rc = gemm2::queue_acquire(&queue, stream)) return rc;
    grid = n_cu;
  }
#ifdef AC_G7_WINDOW_RECT
  const bool windowed = AC_G7_WINDOW_RECT != 0;
#else
  // from the WHOLE id space, not from this part's slice: the two orders are different bijections of the same ids, so every
  // part of a sliced launch must pick the same one (or some tiles run twice and others never)
  const bool windowed = total < (long long)RECT_WINDOW_TILES_PER_CU * n_cu;
#endif
  hipLaunchKernelGGL((kernel<Epi, MT>), dim3((unsigned)grid), dim3(threads_of(MT)), LDS_BYTES + LDS_EXTRA + ((AC_ABLATE7 & 32) ? LDS_ABLATE : 0), stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m | (windowed ? 1 << 16 : 0), (int)lo, (int)hi, (int)total, epi, queue);
  return ac_check_launch("gemm_hl7");
}

}  // namespace gemm7
