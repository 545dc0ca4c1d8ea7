// Split-fp16 MFMA GEMM, 256x256x32 tiles on 8 waves (2 per SIMD).
//
// Why: tile-phase timing of gemm_hl2.h (tests/tile_timing.py) shows its k loop runs
// at 1.25-1.5 us per stage where the MFMAs need 0.85: with a 3-slot ring only two 48 KB stages are in
// flight per CU, and at the ~2.5 us global->LDS latency seen under load that is 68 % of the 56 GB/s per
// CU the tile consumes (Little's law) - exactly the measured efficiency.  A 256x256 tile needs half the
// bytes per FLOP, so the same ~100 KB in flight covers the latency with room to spare.
//
// Layout: wave (wm, wn) of a 4 x 2 grid owns 64 rows x 128 columns:
//   acc   4 x 8 tiles of 16x16 (v_mfma_f32_16x16x32_f16)            128 registers
//   B     all 8 column fragments (hi + lo) of the stage               64
//   A     one row fragment (hi + lo), double buffered                 16
// and runs ONE continuous stream of row steps (24 MFMAs each): the next A fragment is read during the
// current step; in the LAST step of a stage each B fragment is re-read for the next stage right after
// its final MFMA (21 MFMAs before its first use), so there is no fragment-load bubble and no second B
// register set.  No room for a second accumulator: the all-pairs kernel, which needs two-level
// accumulation for fp32-grade distances, stays on gemm_hl2.h.  (Tried: folding -(|a|^2 + |b|^2)/2 into the
// operands as an extra k stage per 1024 features so that one accumulator holds -|a-b|^2/2: 6.5e-5 from the
// fp64 truth against 1.8e-5 with the second accumulator - the error follows the +-175 swing inside a chunk,
// and the stage density that would fix it costs more MFMA work than this kernel saves.)
//
// LDS (160 KB) = five 32 KB half-tiles: A in a 3-slot ring, B in a 2-slot ring, filled by
// global_load_lds, 8 instructions per half-tile from each of the waves 0-3.  ONE barrier per stage t,
// X(t), before the last row step: everybody's share of A(t+1), B(t+1) has landed; nobody reads A(t) any
// more and every wave took B(t) into registers a stage ago
//   -> A(t+3) goes into A(t)'s slot (two stages of latency cover), B(t+2) into B(t)'s slot (one stage).
// (A second barrier right after the B re-reads bought B 1.75 stages of cover but cost 3 %: the counter
// waits are ~50 cycles per stage either way, the barrier is what is expensive - see DESIGN.md section 4.)
#pragma once
#include "gemm_hl2.h"

namespace gemm5 {

using gemm::EpiTraits;
using gemm2::gptr_t;
using gemm2::lptr_t;
using gemm2::tile_off;
using Frag = gemm::FragT<16>;

constexpr int BM = 256, BK = 32;
constexpr int threads_of(int WN) { return 4 * WN * 64; }     // waves: 4 along M x WN along N
constexpr int HALF = 256 * 128;                 // the A half-tile: 256 rows x 128 B
constexpr int A_SLOTS = 3, B_SLOTS = 2;
// NT = 16-column MFMA tiles per wave, WN = waves along N (4 waves along M always):
//   NT 8, WN 2 -> 256x256, 8 waves, 2 per SIMD (the shape described above)
//   NT 4, WN 2 -> 256x128, 64x64 per wave (room for the second accumulator of the two-level sum, FLUSH)
//   NT 4, WN 3 -> 256x192, TWELVE waves, 3 per SIMD at <= 168 registers: a third MFMA-issuing wave per SIMD
constexpr int bn_of(int NT, int WN = 2) { return NT * 16 * WN; }
constexpr int lds_bytes(int NT, int WN = 2) { return A_SLOTS * HALF + B_SLOTS * bn_of(NT, WN) * 128; }   // 160 / 128 / 144 KB
constexpr int FLUSH_STAGES = 16;

// Timing-only diagnostic builds (results are wrong): -DAC_ABLATE5=1 no DMA in the k loop, 2 no fragment
// reads in the k loop, 4 no MFMAs, 8 no barriers in the k loop (combinable).  0 in every shipped build.
#ifndef AC_ABLATE5
#define AC_ABLATE5 0
#endif
// -DAC_TILE_TIMING: wave 0 of every workgroup accumulates the shader-clock cycles it spends waiting at
// X (DMA landed + barrier) and at Y (fragment reads done + barrier), and its total k-loop cycles.
#ifdef AC_TILE_TIMING
static __device__ unsigned long long ac_wait_times[8192 * 8];
#define AC_T0() const unsigned long long t0_ = __builtin_readcyclecounter()
#define AC_TACC(v) v += __builtin_readcyclecounter() - t0_
#else
#define AC_T0()
#define AC_TACC(v)
#endif

// Map a linear tile id to (tm, tn); false for a hole of the folded symmetric grid.  Symmetric problems
// run the upper triangle (tm <= tn) of the T x T tile matrix folded into a ceil(T/2) x (T+1) rectangle:
// folded row p holds row p (T-p tiles) followed by row T-1-p (p+1 tiles).
template <class Epi>
__device__ inline bool tile_of_square(int id, int tiles_m, int tiles_n, int group_m, const Epi& epi, int& tm, int& tn) {
  gemm::raster(id, tiles_m, tiles_n, group_m, tm, tn);
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      const int T = tiles_n - 1, p = tm, c = tn;
      if (c < T - p) { tm = p; tn = p + c; }
      else {
        if (2 * p == T - 1) return false;          // odd T: the middle row is paired with itself
        tm = T - 1 - p; tn = tm + (c - (T - p));
      }
    }
  }
  return true;
}

// PLAIN: the operands are plain fp16 rows (128-byte line = 64 consecutive k values), not hl32: a stage is two
// MFMAs per tile (first 32 k, second 32 k) instead of the three split-fp16 passes; the caller passes K / 2 as K so
// that row bytes and stage count come out right.  (The approximate first pass of the pruned nearest-neighbour
// search, alpha.hip.)
template <class Epi, int NT, bool FLUSH, int WN = 2, bool PLAIN = false>
__global__ __launch_bounds__(threads_of(WN), (WN == 2 ? 1 : WN)) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int id_lo, int id_hi, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const Aring = smem;
  uint8_t* const Bring = smem + A_SLOTS * HALF;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef AC_TILE_TIMING
  // phase stamps of thread 0 (launch slot in the upper half of group_m): entry, first stage visible, k loop done,
  // epilogue stores done - read back with ac_debug_tile_times (tests/tile_timing.py)
  const int tt_slot = (group_m >> 16) & 7;
  group_m &= 0xffff;
#define G5_STAMP(k)                                                                                       \
  if (tid == 0 && blockIdx.x < gemm2::TT_BLOCKS)                                                          \
    gemm2::ac_tile_times[((size_t)tt_slot * gemm2::TT_BLOCKS + blockIdx.x) * 4 + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define G5_STAMP(k)
#endif
  G5_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  constexpr bool SWAP = gemm::HasSwap<Epi>::value;   // accumulate C^T tiles (see gemm_hl.h)
  static_assert(WN == 2 || !EpiTraits<Epi>::kSymmetric, "symmetric epilogues run on the 2-wave-wide kernels");
  const int nk = K / BK;
  const size_t ld = (size_t)K * 4;

  int tm = 0, tn = 0;
  const int id = id_lo + blockIdx.x;
  constexpr int BN = bn_of(NT, WN), HALF_B = BN * 128, NB = BN / 32;   // NB: B instructions per loader wave
  if (id >= id_hi) return;
  if constexpr (NT == 8) { if (!tile_of_square(id, tiles_m, tiles_n, group_m, epi, tm, tn)) return; }
  else if constexpr (WN == 2) { if (!gemm2::tile_of(id, tiles_m, tiles_n, group_m, M, N, epi, tm, tn)) return; }
  else gemm::raster(id, tiles_m, tiles_n, group_m, tm, tn);                                  // block-uniform
  const int m0 = tm * BM, n0 = tn * BN;

  // DMA map: a half-tile is 32 wave-instructions of 1 KB (8 rows each).  Only waves 0-3 issue them
  // (see below): wave w issues g = 4j + w, j = 0..7.  Lane l lands in slot l of the instruction: row
  // 8g + (l>>3), physical chunk l&7, so it fetches logical chunk (l&7) ^ ((row>>1)&7) of that row.
  // Swapped orientation: LDS row r of the B tile holds global row perm(r) (gemm_hl.h).
  //
  // Who issues: the SIMD's arbiter serves its older wave first, so of the two waves of a SIMD (w, w+4)
  // wave w runs ahead and parks at the next barrier, and whatever wave w+4 has to do besides MFMAs while
  // it catches up alone is exposed (wait-time stamps, tests/wait_timing.py: wave 0 idles ~800 cycles at
  // each barrier, wave 4 never waits; stage time = 2 x MFMA time + the younger wave's other work).  The
  // older wave's other work is free - the younger one issues MFMAs meanwhile - so ALL the DMA goes to
  // waves 0-3, which also get the explicit higher priority.
  const bool loader = wave < 4;
  if (loader) __builtin_amdgcn_s_setprio(2);
  // Row r = 32j + 8(w&3) + (l>>3); (r>>1)&7 and with it the chunk swizzle do not depend on j, so the
  // per-lane state is one row number and one chunk offset and the address is rebuilt per instruction
  // (3 VALU operations on a wave whose stalls are hidden anyway) instead of living in 16 registers.
  const int row0 = (wave & 3) * 8 + (lane >> 3);
  const unsigned lc16 = (unsigned)((lane & 7) ^ ((row0 >> 1) & 7)) * 16u;
  auto dmaA = [&](int t, int slot) {
    if (!loader) return;
    const uint8_t* base = A + (size_t)t * 128 + lc16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t off = (size_t)min(m0 + row0 + 32 * j, M - 1) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(Aring + slot * HALF + (j * 4 + (wave & 3)) * 1024), 16, 0, 0);
    }
  };
  auto dmaB = [&](int t, int slot) {
    if (!loader) return;
    const uint8_t* base = B + (size_t)t * 128 + lc16;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int row = row0 + 32 * j;
      const int brow = SWAP ? ((row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3)) : row;
      const size_t off = (size_t)min(n0 + brow, N - 1) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(Bring + slot * HALF_B + (j * 4 + (wave & 3)) * 1024), 16, 0, 0);
    }
  };
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // fragment offsets: 16-row tile t of this wave's rows is base + t * 2048 bytes (the XOR swizzle term
  // (row >> 1) & 7 is unchanged by +16 rows)
  const int a_hi = tile_off(wm * 64 + (lane & 15), (lane >> 4)), a_lo = tile_off(wm * 64 + (lane & 15), 4 + (lane >> 4));
  const int b_hi = tile_off(wn * (NT * 16) + (lane & 15), (lane >> 4)), b_lo = tile_off(wn * (NT * 16) + (lane & 15), 4 + (lane >> 4));

  f32x4 acc[4][NT], tot[FLUSH ? 4 : 1][FLUSH ? NT : 1];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[mi][ni][i] = 0.f; if constexpr (FLUSH) tot[mi][ni][i] = 0.f; }

  // prologue: A(0) B(0) A(1) B(1) A(2) in that order; the first two must have landed
  dmaA(0, 0);
  dmaB(0, 0);
  if (nk > 1) { dmaA(1, 1); dmaB(1, 1); }
  if (nk > 2) dmaA(2, 2);
  // leave A(1) B(1) A(2) = 8 + NB + 8 (or A(1) B(1) = 8 + NB) in flight
  if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 + NB) : "memory");
  else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NB) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  half8 bh[NT], bl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bh[t] = *reinterpret_cast<const half8*>(Bring + b_hi + t * 2048);
    bl[t] = *reinterpret_cast<const half8*>(Bring + b_lo + t * 2048);
  }
  half8 ah = *reinterpret_cast<const half8*>(Aring + a_hi), al = *reinterpret_cast<const half8*>(Aring + a_lo);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  barrier();                                     // B(0) is in registers everywhere: its slot may be refilled
  G5_STAMP(1);
  auto mfma3 = [&](f32x4& c, const half8& xh, const half8& xl, const half8& yh, const half8& yl) {
    if constexpr (PLAIN) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yl, c, 0, 0, 0);
    } else if constexpr (SWAP) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, xh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xh, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, c, 0, 0, 0);
    }
  };

  int sa = 0;                                    // A slot of the current stage (kt % 3)
#ifdef AC_TILE_TIMING
  unsigned long long tx = 0, ty = 0, tx1 = 0, ty1 = 0, td = 0;   // (ty*: no second barrier any more)
  const unsigned long long tloop = __builtin_readcyclecounter();
#endif
  auto stage = [&](int kt, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;   // branch-free body while three more stages follow
    const bool more1 = STEADY || kt + 1 < nk, more2 = STEADY || kt + 2 < nk, more3 = STEADY || kt + 3 < nk;
    const int sa1 = sa == 2 ? 0 : sa + 1;        // (kt + 1) % 3
    const uint8_t* As = Aring + sa * HALF;
    const uint8_t* An = Aring + sa1 * HALF;
    const uint8_t* Bn = Bring + ((kt + 1) & 1) * HALF_B;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (mi == 3 && more1) {
        // own share of A(kt+1), B(kt+1) landed: A(kt+2) (8 instructions, the youngest) may still fly
        AC_T0();
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        AC_TACC(tx1);
        if (!(AC_ABLATE5 & 8)) barrier();        // X(kt)
        AC_TACC(tx);
        if (more2 && !(AC_ABLATE5 & 1)) dmaB(kt + 2, kt & 1);   // B(kt) went to registers a stage ago
        if (more3 && !(AC_ABLATE5 & 1)) { AC_T0(); dmaA(kt + 3, sa); AC_TACC(td); }
      }
      half8 nh = ah, nl = al;
      if (AC_ABLATE5 & 2) {
      } else if (mi < 3) {
        nh = *reinterpret_cast<const half8*>(As + a_hi + (mi + 1) * 2048);
        nl = *reinterpret_cast<const half8*>(As + a_lo + (mi + 1) * 2048);
      } else if (more1) {
        nh = *reinterpret_cast<const half8*>(An + a_hi);
        nl = *reinterpret_cast<const half8*>(An + a_lo);
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        if (!(AC_ABLATE5 & 4)) mfma3(acc[mi][ni], ah, al, bh[ni], bl[ni]);
        if (mi == 3 && more1 && !(AC_ABLATE5 & 2)) {
          bh[ni] = *reinterpret_cast<const half8*>(Bn + b_hi + ni * 2048);
          bl[ni] = *reinterpret_cast<const half8*>(Bn + b_lo + ni * 2048);
        }
      }
      ah = nh; al = nl;
    }
    // Two-level accumulation (gemm_hl.h): flush the MFMA accumulator every 512 k-terms.
    if constexpr (FLUSH) {
      if ((kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) { tot[mi][ni][i] += acc[mi][ni][i]; acc[mi][ni][i] = 0.f; }
      }
    }
    sa = sa1;
  };
  int kt = 0;
  for (; kt + 3 < nk; ++kt) stage(kt, std::true_type{});
  for (; kt < nk; ++kt) stage(kt, std::false_type{});
  barrier();                                     // ring reads done before an epilogue reuses LDS
  G5_STAMP(2);
  if constexpr (FLUSH) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mi][ni][i] += tot[mi][ni][i];
  }
#ifdef AC_TILE_TIMING
  if ((tid == 0 || tid == 256) && blockIdx.x < 4096) {   // waves 0 and 4: the two waves of one SIMD
    unsigned long long* o = ac_wait_times + (blockIdx.x * 2 + (tid >> 8)) * 8;
    o[0] = __builtin_readcyclecounter() - tloop;
    o[1] = tx; o[2] = ty; o[3] = nk; o[4] = tx1; o[5] = ty1; o[6] = td;
  }
#endif

  // epilogue: the wave's 64 x (NT*16) accumulator as NT/4 64x64 fragments of the common epilogue interface
  const int m_wave = m0 + wm * 64;
  if constexpr (SWAP && gemm::HasWholeTile<Epi>::value) {
    epi.template swapped_tile<NT>(acc, m_wave, n0 + wn * (NT * 16), lane);
#ifdef AC_TILE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G5_STAMP(3);
#endif
    return;
  }
#pragma unroll
  for (int no = 0; no < NT / 4; ++no) {
    Frag f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) f.acc[mi][ni] = acc[mi][no * 4 + ni];
    const int n_wave = n0 + wn * (NT * 16) + no * 64;
    if constexpr (SWAP) {
      epi.swapped(f, m_wave, n_wave, lane);
    } else if constexpr (EpiTraits<Epi>::kSymmetric) {
      if (epi.symmetric) {
        // Row block rb against column block cb of the symmetric matrix: rb < cb reduce both ways,
        // rb == cb one way, rb > cb (lower part of a diagonal tile) is its mirror image's job.
        // NT == 8: 64-blocks of a square tile; NT == 4: 128-blocks of a 256x128 tile (gemm_hl2.h).
        const int rb = NT == 8 ? tm * 4 + wm : 2 * tm + (wm >> 1), cb = NT == 8 ? tn * 4 + wn * 2 + no : tn;
        // Every product a.b is formed with the LOWER patch row as the A operand and reduced in both
        // directions - also inside the diagonal blocks, where only the elements on or above the diagonal are
        // used.  The three split-fp16 passes are accumulated in a fixed order, so a.b and b.a differ in the
        // last bit; with one orientation per unordered pair the result does not depend on where tile and block
        // boundaries fall, which is what makes blocked / sharded runs bit-identical to a single launch.
        const bool diag = rb == cb;
        if (rb < cb || (diag && m_wave <= n_wave + 63)) {
          epi(f, m_wave, n_wave, lane, diag);
          epi.transposed(f, m_wave, n_wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65), diag);
        }
      } else {
        epi(f, m_wave, n_wave, lane);
        // two-way rectangle (A and B are different row sets of one symmetric problem): every tile is also
        // reduced the other way, into the epilogue's second matrix
        if (epi.two_way) epi.transposed(f, m_wave, n_wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65));
      }
    } else {
      epi(f, m_wave, n_wave, lane);
    }
  }
#ifdef AC_TILE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G5_STAMP(3);
#endif
}

template <class Epi, int NT = 8, bool FLUSH = false, int WN = 2, bool PLAIN = false>
#ifndef AC_G5_GROUP_M
#define AC_G5_GROUP_M 4          // L2 window of the linear-layer launches, in tile rows (same-box A/B knob)
#endif
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = AC_G5_GROUP_M, int part = 0, int nparts = 1) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl5: K=%d must be a positive multiple of 32", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl5: bad part %d/%d", part, nparts);
  constexpr int BN = bn_of(NT, WN), LDS_BYTES = lds_bytes(NT, WN), THREADS = threads_of(WN);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  long long total = (long long)tiles_m * tiles_n;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // upper triangle (tile_of_square / gemm2::tile_of)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl5: symmetric epilogue needs M == N");
      if (NT == 8) {                               // folded into a ceil(T/2) x (T+1) rectangle
        const int Tm = tiles_m;
        tiles_m = (Tm + 1) / 2;
        tiles_n = Tm + 1;
        total = (long long)tiles_m * tiles_n;
      } else {
        total = gemm2::tri_rows_before(tiles_m, tiles_n);
      }
    }
  }
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl5: grid too large");
  static unsigned long long attr_done = 0;       // per instantiation, one bit per device
  if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&kernel<Epi, NT, FLUSH, WN, PLAIN>), LDS_BYTES, &attr_done)) return rc;
  long long lo = (total * part / nparts) & ~7LL;
  long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  hipLaunchKernelGGL((kernel<Epi, NT, FLUSH, WN, PLAIN>), dim3((unsigned)(hi - lo)), dim3(THREADS), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
#ifdef AC_TILE_TIMING
                     group_m | ((gemm2::tt_counter()++ & 7) << 16),
#else
                     group_m,
#endif
                     (int)lo, (int)hi, epi);
  return ac_check_launch("gemm_hl5");
}

}  // namespace gemm5
