// Split-fp16 MFMA GEMM, second generation: 256x128x32 tiles, 8 waves, LDS-DMA ring, staggered
// two-barrier schedule, persistent over tiles.
//
// Same contract and epilogue interface as gemm_hl.h (C[m][n] = sum_k A[m][k] B[n][k] on hl32
// operands, M on the accumulator registers, N on the lanes, 64x64 per wave), different engine:
//   * 512 threads = 8 waves as 4 (M) x 2 (N); one workgroup per CU, two waves per SIMD;
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB = 8 full 128-byte lines
//     per wave-instruction), no VGPR round trip and no ds_write;
//   * LDS image of a tile: [row][128 B] = the hl32 line of that row; its eight 16-byte chunks are
//     XOR-swizzled with (row>>1)&7 - applied to the per-lane SOURCE address, because the DMA
//     destination is lane-linear - which makes every ds_read_b128 lane group conflict-free
//     (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * ring of 3 stages (3 x 48 KB); the loads of stage t+2 are issued while stage t is multiplied;
//     waits are counted (s_waitcnt vmcnt(6): one stage's 6 wave-instructions stay in flight) and
//     barriers are raw s_barrier (a __syncthreads() would drain the DMA queue);
//   * STAGGER: a stage is split into P1 = {16 fragment reads, 6 DMA issues} and P2 = {48 MFMAs on
//     registers}, separated by barriers; waves 0-3 and waves 4-7 (w and w+4 share a SIMD) run the
//     same code one barrier apart, so on every SIMD one wave multiplies while its partner loads:
//       barrier #:   0      1        2        3
//       A:            P1(0) | P2(0)  | P1(1)  | P2(1) ...
//       B:          (wait)  | P1(0)  | P2(0)  | P1(1) ...
//     Visibility: every wave waits for its own DMA share of stage t+1 before the barrier that precedes
//     group A's P1(t+1) (A at the end of P2(t), B at the end of P1(t)).  Reuse: the ring slot of stage
//     t+2 was last read in P1(t-1), which both groups left (lgkmcnt(0) + barrier) before either issues
//     into it.
//   * PERSISTENT (A/B knob): the grid is one workgroup per CU; each walks tiles id, id+grid, ...  The first two
//     stages of the NEXT tile are issued before the epilogue of the current one, so the epilogue and the
//     DMA latency of the next tile's prologue overlap (K = 768 GEMMs spent ~30 % of their time there).
//   * DYNAMIC (round 4, the all-pairs launches): persistent workgroups PULL tiles from per-XCD-slot queues with
//     stealing - see g_tile_queues below.
// Measured (MI355X, K = 4096 all-pairs): 66 % MFMA-pipe utilisation at the 1.72-1.81 GHz the chip
// sustains under this load; LDS-DMA vs register staging and block-issue vs interleaved DMA issue were
// all within 2 % of each other (profiles/).
#pragma once
#include "gemm_hl.h"

namespace gemm2 {

using gemm::EpiTraits;
using Frag = gemm::FragT<16>;   // v_mfma_f32_16x16x32_f16: 4x4 tiles of 16x16 per wave

constexpr int BM = 256, BN = 128, BK = 32, NSTAGE = 3, THREADS = 512;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 48 KB
constexpr int LDS_BYTES = NSTAGE * STAGE;                                           // 144 KB
constexpr int LDS_EXTRA = 16;                                                       // the next tile id of a dynamic walk
constexpr int FLUSH_STAGES = 16;

// DYNAMIC tile queues (round 4; launch(..., dynamic = true)).  One workgroup per CU walks tiles it PULLS: eight queues,
// one per XCD slot (blockIdx & 7), queue x handing out the block ids x, x + 8, x + 16, ... in order - exactly the tiles
// and the order the hardware dispatcher gives that XCD in a one-workgroup-per-tile launch, so the L2 locality of the
// rasterisation is unchanged - and a workgroup whose own queue is empty steals from the others.  Why: equal tile
// counts per XCD do not finish together (sampled stamps, profiles/r04_allpairs_tile_timing.txt: the eight XCDs end
// 1.35 ms apart on a 42 ms launch); stealing evens that out, and the workgroup launch between two tiles goes away.
// Counters: QUEUE_SLOTS sets of 8 tickets in device memory, one set per launch in rotation; the launcher zeroes the
// chosen set ON THE LAUNCH STREAM right before the kernel (round 5: a kernel that died can no longer leave stale
// tickets behind, and nothing depends on the previous user of the set having left cleanly).  Two launches share a set
// only if QUEUE_SLOTS dynamic launches of this translation unit are in flight on the device at once.  Every workgroup
// leaves when all eight queues are exhausted: no spinning.
constexpr int QUEUE_SLOTS = 64, QUEUE_STRIDE = 16;
static __device__ unsigned g_tile_queues[QUEUE_SLOTS * QUEUE_STRIDE];

// Thread 0 of a workgroup pulls in two steps, so that the ticket's round trip overlaps other memory latency instead of
// stalling its wave: queue_take() issues the returning atomic on the workgroup's OWN queue; queue_resolve() - later -
// turns the ticket into a block id and steals from the other queues only when the own one is exhausted (the tail of the
// launch); -1 = every queue is exhausted.
// Round 5 (advice on round 4): the atomic is the compiler's own builtin, so hipcc KNOWS the result register is pending and
// places the s_waitcnt in front of its first use - round 4 issued it from inline assembly, whose output register the
// compiler was free to copy or spill before the value had arrived.  What keeps the builtin from being rewritten: with a
// wave-uniform address hipcc's atomic optimiser turns the add into a wave reduction whose v_readfirstlane consumes the
// result at once (a full memory round trip with wave 0 parked right after the k loop); an opaque zero from a one-line
// asm makes the address divergent in the compiler's eyes and the add stays one global_atomic_add_u32 ... sc0.
__device__ inline unsigned queue_take(unsigned* queue) {
  unsigned zero;
  asm("v_mov_b32 %0, 0" : "=v"(zero));
  return __hip_atomic_fetch_add(queue + (blockIdx.x & 7) + zero, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int queue_resolve(unsigned* queue, unsigned ticket, int id_lo, int id_hi) {
  // the first USE of the ticket stays here: without this hipcc hoists the id arithmetic up to the atomic (same branch
  // of thread 0) and waits for the round trip on the spot
  asm volatile("" : "+v"(ticket) :: "memory");
  const int x = blockIdx.x & 7;
  long long cand = (long long)id_lo + 8ll * ticket + x;
  int got = cand < id_hi ? (int)cand : -1;
  for (int j = 1; j < 8 && got < 0; ++j) {
    const int y = (x + j) & 7;
    cand = (long long)id_lo + 8ll * atomicAdd(queue + y, 1u) + y;
    if (cand < id_hi) got = (int)cand;
  }
  return got;
}
// Host side: the next counter set of THIS translation unit in rotation, zeroed on the launch stream.
static inline int queue_acquire(unsigned** queue, hipStream_t stream) {
  static unsigned next_slot = 0;                 // one per translation unit, next to its g_tile_queues
  static void* base_of[64] = {};                 // per device: the address of this module's counters there
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ac_fail(AC_ELAUNCH, "tile queues: hipGetDevice failed");
  void* base = __atomic_load_n(&base_of[dev], __ATOMIC_RELAXED);
  if (!base) {
    if (hipGetSymbolAddress(&base, HIP_SYMBOL(g_tile_queues)) != hipSuccess || !base)
      return ac_fail(AC_ELAUNCH, "tile queues: hipGetSymbolAddress(g_tile_queues) failed");
    __atomic_store_n(&base_of[dev], base, __ATOMIC_RELAXED);
  }
  const unsigned slot = __atomic_fetch_add(&next_slot, 1u, __ATOMIC_RELAXED) % QUEUE_SLOTS;
  *queue = static_cast<unsigned*>(base) + slot * QUEUE_STRIDE;
  if (hipMemsetAsync(*queue, 0, 8 * sizeof(unsigned), stream) != hipSuccess) return ac_fail(AC_ELAUNCH, "tile queues: hipMemsetAsync failed");
  return AC_OK;
}

// Timing-only diagnostic builds (results are wrong): -DAC_ABLATE=1 no DMA issue in the loop, 2 no
// fragment reads after stage 0, 4 no MFMAs (combinable).  0 in every shipped build.
#ifndef AC_ABLATE
#define AC_ABLATE 0
#endif
// Wave priority of the two half-stages (same-box A/B knobs, scratch/ab_pairmin.py; round-3 result in DESIGN.md
// section 4): P2 = the 48-MFMA segment, P1 = fragment reads + DMA issue of the partner wave.
#ifndef AC_G2_P2PRIO
#define AC_G2_P2PRIO 1
#endif
#ifndef AC_G2_P1PRIO
#define AC_G2_P1PRIO 0
#endif

// Diagnostic build -DAC_TILE_TIMING: every workgroup records 100 MHz timestamps (entry, first stage
// visible, k loop done, epilogue stores done) into a per-TU buffer; the `stagger` argument then carries
// the launch slot instead.  Read back with ac_debug_tile_times (vit.hip).  Never defined in shipped builds.
#ifdef AC_TILE_TIMING
constexpr int TT_SLOTS = 8, TT_BLOCKS = 8192;
static __device__ unsigned long long ac_tile_times[TT_SLOTS * TT_BLOCKS * 4];
inline int& tt_counter() { static int c = 0; return c; }
#ifndef AC_TT_STRIDE
#define AC_TT_STRIDE 1        // stamp every AC_TT_STRIDE-th workgroup (8: a sample over the whole of a 65 000-tile launch)
#endif
#define AC_STAMP(k)                                                                                   \
  if (tid == 0 && blockIdx.x % AC_TT_STRIDE == 0 && blockIdx.x / AC_TT_STRIDE < gemm2::TT_BLOCKS)   \
    gemm2::ac_tile_times[((size_t)(stagger & 7) * gemm2::TT_BLOCKS + blockIdx.x / AC_TT_STRIDE) * 4 + (k)] = __builtin_amdgcn_s_memrealtime()
#else
#define AC_STAMP(k)
#endif

__device__ inline int tile_off(int row, int lc) { return row * 128 + ((lc ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Tiles of the upper triangle of a symmetric problem on 256-row x 128-column tiles: in units of 128-blocks a
// tile holds row blocks {2tm, 2tm+1} and column block tn, so tile row p needs the columns tn >= 2p:
// cnt(p) = Tn - 2p tiles, F(p) = p Tn - p (p - 1) tiles in the rows above p.
__host__ __device__ inline long long tri_rows_before(int p, int Tn) { return (long long)p * Tn - (long long)p * (p - 1); }

// Linear order of those tiles, made for the L2: groups of G tile rows; inside a group the ACTUAL column tn runs
// slowest and the row fastest, so the 32 workgroups an XCD runs at one time (a contiguous run of this order)
// cover G row panels x 32/G column panels that they all share.  (Round 1 folded the triangle into a rectangle
// first; a window of folded columns then touched 2G - 2 + 32/G different column panels.  Same-box A/B,
// profiles/r02_raster_ab.txt: -2 % time, G = 6 best.)
// Equal work per XCD and per GPU slice needs no folding: the order is cut into equal runs by tile count.
__host__ __device__ inline void tri_tile(int lin, int Tm, int Tn, int G, int& tm, int& tn) {
  // row p with F(p) <= lin < F(p + 1): root of p^2 - (Tn + 1) p + lin = 0, then the group it belongs to
  const double b = (double)Tn + 1.0;
  int p = (int)((b - sqrt(fmax(b * b - 4.0 * (double)lin, 0.0))) * 0.5);
  p = max(0, min(p, Tm - 1));
  while (p > 0 && tri_rows_before(p, Tn) > lin) --p;
  while (p + 1 < Tm && tri_rows_before(p + 1, Tn) <= lin) ++p;
  const int p0 = (p / G) * G, gsz = min(G, Tm - p0);
  int w = lin - (int)tri_rows_before(p0, Tn);           // index inside the group
  // column c = tn - 2 p0 has v(c) = min(gsz, c/2 + 1) valid rows; the ragged head is 2 (gsz - 1) columns and
  // gsz (gsz - 1) tiles long, every later column is full
  const int head = gsz * (gsz - 1);
  int c, r;
  if (w >= head) {
    w -= head;
    c = 2 * (gsz - 1) + w / gsz;
    r = w - (w / gsz) * gsz;
  } else {
    c = 0;
    for (;;) {                                          // at most 2 (G - 1) steps
      const int v = c / 2 + 1;
      if (w < v) break;
      w -= v;
      ++c;
    }
    r = w;
  }
  tm = p0 + r;
  tn = 2 * p0 + c;
}

__host__ __device__ inline int xcd_chunk_lin(int id, long long total) {
  return gemm::xcd_lin<AC_XCD_WINDOW_TRI>(id, total);
}

// Map a linear tile id to (tm, tn).  Symmetric problems: tiles_m / tiles_n are the full tile counts Tm / Tn and
// the id space is the triangle above.
template <class Epi>
__device__ inline bool tile_of(int id, int tiles_m, int tiles_n, int group_m, int M, int N, const Epi& epi,
                               int& tm, int& tn) {
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      const long long total = tri_rows_before(tiles_m, tiles_n);
      tri_tile(xcd_chunk_lin(id, total), tiles_m, tiles_n, group_m, tm, tn);
      return true;
    }
  }
  gemm::raster(id, tiles_m, tiles_n, group_m, tm, tn);
  return true;
}

template <class Epi, bool FLUSH>
__global__ __launch_bounds__(THREADS, 2) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int id_lo, int id_hi, int stagger, Epi epi, unsigned* queue) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool groupB = wave >= 4;
  constexpr bool SWAP = gemm::HasSwap<Epi>::value;   // accumulate C^T tiles (see gemm_hl.h)
  // (`stagger`: launch slot of the -DAC_TILE_TIMING stamps.  A start stagger of the first tile round across
  // the CUs, meant to turn the lockstep epilogue bursts into a steady stream, measured -1...-6 %: removed.)
  (void)stagger;
  AC_STAMP(0);
  const size_t ld = (size_t)K * 4;
  const int nk = K / BK;

  // per-lane fragment offsets inside a stage: [16-row tile][plane].  One 16x16x32 MFMA consumes the
  // whole 32-deep stage: lane l holds row l&15, k = 8(l>>4) .. +7, i.e. 16-byte chunk l>>4 of the row.
  int a_off[4][2], b_off[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      a_off[t][p] = tile_off(wm * 64 + t * 16 + (lane & 15), p * 4 + (lane >> 4));
      b_off[t][p] = A_BYTES + tile_off(wn * 64 + t * 16 + (lane & 15), p * 4 + (lane >> 4));
    }

  // ---- DMA map: wave-instruction g (0..47) fills LDS bytes [g*1024, +1024) of a stage:
  // g < 32 -> rows 8g..8g+7 of the A tile, else rows 8(g-32).. of the B tile.  Wave w issues
  // g = j*8 + w, j = 0..5.  Lane l lands in slot l: row (l>>3), physical chunk (l&7).
  const uint8_t* src[6];
  int tm = 0, tn = 0;
  auto setup = [&](int id) -> bool {              // tile pointers of block-uniform tile `id`
    if (!tile_of(id, tiles_m, tiles_n, group_m, M, N, epi, tm, tn)) return false;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int g = j * 8 + wave;
      const bool isA = j < 4;
      const int row = (isA ? g : g - 32) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      // swapped orientation: LDS row r of the B tile holds global row perm(r) (bits 4|3:2 -> 2|4:3 within
      // each group of 32), which makes a lane's two tiles of a pair cover 8 consecutive columns
      const int brow = SWAP ? ((row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3)) : row;
      const uint8_t* base = isA ? A + (size_t)min(tm * BM + row, M - 1) * ld : B + (size_t)min(tn * BN + brow, N - 1) * ld;
      src[j] = base + lc * 16;
    }
    return true;
  };
  auto issue = [&](int kt) {
    uint8_t* stage = smem + (kt % NSTAGE) * STAGE;
    const size_t ko = (size_t)kt * 128;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + ko), (lptr_t)(stage + (j * 8 + wave) * 1024), 16, 0, 0);
  };
  auto wait_one_stage_in_flight = [&](bool more) {
    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // dynamic walk (queue != nullptr): thread 0 pulls the next block id - own queue first, then the others' - and hands it
  // to the workgroup through LDS; -1 = every queue is exhausted
  int* const lds_next = reinterpret_cast<int*>(smem + LDS_BYTES);
  auto take = [&]() -> unsigned { return queue_take(queue); };                                        // thread 0 only
  auto resolve = [&](unsigned ticket) { *lds_next = queue_resolve(queue, ticket, id_lo, id_hi); };   // thread 0 only

  // first tile of this workgroup
  int id;
  if (queue) {
    if (tid == 0) resolve(take());
    __syncthreads();
    id = *lds_next;
    __syncthreads();                               // everyone has read it before thread 0 may overwrite it
    if (id < 0) return;
    setup(id);
  } else {
    id = id_lo + blockIdx.x;
    while (id < id_hi && !setup(id)) id += gridDim.x;
    if (id >= id_hi) return;
  }
  issue(0);
  if (nk > 1) issue(1);

  for (;;) {
    const int m0 = tm * BM, n0 = tn * BN, tm_cur = tm, tn_cur = tn;
    Frag f, tot;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) { f.acc[mi][ni][i] = 0.f; if (FLUSH) tot.acc[mi][ni][i] = 0.f; }

    wait_one_stage_in_flight(nk > 1);
    barrier();                                      // #0: stage 0 visible to everyone
    AC_STAMP(1);
    if (groupB) barrier();
    half8 ah[4], al[4], bh[4], bl[4];
    for (int kt = 0; kt < nk; ++kt) {
      // ---------------- P1(kt)
      const uint8_t* cur = smem + (kt % NSTAGE) * STAGE;
      if (!(AC_ABLATE & 2) || kt == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          ah[t] = *reinterpret_cast<const half8*>(cur + a_off[t][0]);
          al[t] = *reinterpret_cast<const half8*>(cur + a_off[t][1]);
          bh[t] = *reinterpret_cast<const half8*>(cur + b_off[t][0]);
          bl[t] = *reinterpret_cast<const half8*>(cur + b_off[t][1]);
        }
      }
      if (kt + 2 < nk && !(AC_ABLATE & 1)) issue(kt + 2);
      if (groupB) wait_one_stage_in_flight(kt + 2 < nk);      // own share of stage kt+1 landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- P2(kt)
      __builtin_amdgcn_s_setprio(AC_G2_P2PRIO);
      if (!(AC_ABLATE & 4))
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          if constexpr (SWAP) {
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ni], ah[mi], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], al[mi], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ni], ah[mi], f.acc[mi][ni], 0, 0, 0);
          } else {
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ni], f.acc[mi][ni], 0, 0, 0);
          }
        }
      __builtin_amdgcn_s_setprio(AC_G2_P1PRIO);
      // Two-level accumulation (see gemm_hl.h): flush the MFMA accumulator every 512 k-terms.
      if (FLUSH && (kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int i = 0; i < 4; ++i) { tot.acc[mi][ni][i] += f.acc[mi][ni][i]; f.acc[mi][ni][i] = 0.f; }
      }
      if (!groupB) wait_one_stage_in_flight(kt + 2 < nk);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
    }
    if (!groupB) barrier();                         // both groups are past their last ring read
    AC_STAMP(2);
    if (FLUSH) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int i = 0; i < 4; ++i) f.acc[mi][ni][i] += tot.acc[mi][ni][i];
    }

    // next tile: its first two stages go out before this tile's epilogue (ring slots 0/1 are free;
    // epilogues that use LDS themselves issue after they are done instead)
    int next = id + gridDim.x;
    bool has_next = false;
    constexpr bool kUsesLds = !SWAP && (EpiTraits<Epi>::kStaged || EpiTraits<Epi>::kSymmetric);
    unsigned ticket = 0;
    if (queue) {
      if (tid == 0) ticket = take();                 // in flight beside the epilogue's own loads; resolved after the epilogue
    } else {
      while (next < id_hi && !(has_next = setup(next))) next += gridDim.x;
      if (has_next && !kUsesLds) { issue(0); if (nk > 1) issue(1); }
    }

    const int m_wave = m0 + wm * 64, n_wave = n0 + wn * 64;
    if constexpr (SWAP) {
      epi.swapped(f, m_wave, n_wave, lane);
    } else if constexpr (EpiTraits<Epi>::kStaged) {
      float* t = reinterpret_cast<float*>(smem) + wave * (64 * gemm::EPI_PITCH);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            t[Frag::row(mi, i, lane) * gemm::EPI_PITCH + Frag::col(ni, lane)] = f.acc[mi][ni][i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      epi.staged(t, m_wave, n_wave, lane);
    } else if constexpr (EpiTraits<Epi>::kSymmetric) {
      if (epi.symmetric) {
        // this wave's 64 rows lie in 128-row block rb, its columns in block cb = tn:
        // rb < cb: off-diagonal, reduce both ways; rb == cb: diagonal block, one way; rb > cb: the
        // lower half of a diagonal tile - its mirror image is computed by tile (tm, tn+1)
        const int rb = 2 * tm_cur + (wm >> 1), cb = tn_cur;
        // Every product a.b is formed with the LOWER patch row as the A operand and reduced in both
        // directions - also inside the diagonal blocks, where only the elements on or above the diagonal are
        // used.  The three split-fp16 passes are accumulated in a fixed order, so a.b and b.a differ in the
        // last bit; with one orientation per unordered pair the result does not depend on where tile and block
        // boundaries fall, which is what makes blocked / sharded runs bit-identical to a single launch.
        const bool diag = rb == cb;
        if (rb < cb || (diag && m_wave <= n_wave + 63)) {
          // off the diagonal and inside the problem: both reductions from registers (epi.fast); else the general code
          if (diag || !epi.fast(f, m_wave, n_wave, lane, true)) {
            epi(f, m_wave, n_wave, lane, diag);
            epi.transposed(f, m_wave, n_wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65), diag);
          }
        }
      } else if (!epi.fast(f, m_wave, n_wave, lane, epi.two_way)) {
        epi(f, m_wave, n_wave, lane);
        // two-way rectangle (A and B are different row sets of one symmetric problem): every tile is also
        // reduced the other way, into the epilogue's second matrix
        if (epi.two_way) epi.transposed(f, m_wave, n_wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65));
      }
    } else {
      epi(f, m_wave, n_wave, lane);
    }
#ifdef AC_TILE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    AC_STAMP(3);
#endif
    if (queue) {
      if (tid == 0) resolve(ticket);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // thread 0's block id is in LDS; everyone is out of the ring
      barrier();
      next = *lds_next;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();                                   // ... and has read the ticket before the next pull overwrites it
      if (next < 0) break;
      setup(next);
      issue(0);
      if (nk > 1) issue(1);
      id = next;
      continue;
    }
    if (!has_next) break;
    if (kUsesLds) {                                // the epilogue used the ring: everyone must be out of it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
      issue(0);
      if (nk > 1) issue(1);
    }
    id = next;
  }
}

// Host launcher.  M, N arbitrary (tails are clamped on load; the epilogue masks them), K % 32 == 0.
// part / nparts: run only the part-th of nparts contiguous slices of the tile-id space (slice
// boundaries are multiples of 8 so the XCD round-robin stays aligned) - used to split one symmetric
// all-pairs problem over several GPUs.
template <class Epi, bool FLUSH = false>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 8, int part = 0, int nparts = 1, bool persistent = false, bool dynamic = false) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl2: K=%d must be a positive multiple of 32", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl2: bad part %d/%d", part, nparts);
  if (group_m < 1) return ac_fail(AC_EINVAL, "gemm_hl2: group_m=%d must be positive", group_m);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  long long total = (long long)tiles_m * tiles_n;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // upper triangle only (see tile_of)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl2: symmetric epilogue needs M == N");
      total = tri_rows_before(tiles_m, tiles_n);
    }
  }
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl2: grid too large");
  static unsigned long long attr_done = 0;       // per instantiation, one bit per device
  if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&kernel<Epi, FLUSH>), LDS_BYTES + LDS_EXTRA, &attr_done)) return rc;
  long long lo = (total * part / nparts) & ~7LL;
  long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  // One workgroup per tile.  (The kernel can also walk tiles persistently, one workgroup per CU with the
  // next tile's first stages issued under the epilogue: same-box A/B equal within noise on the K = 768 ViT
  // GEMMs, 4 % slower on the K = 4096 all-pairs kernel - the static walk loses at the tail what the
  // overlapped prologue gains - so the hardware dispatcher keeps the job; `persistent` is for A/B runs.)
  const int n_cu = ac_device_cus();
  long long grid = (persistent && hi - lo > n_cu) ? n_cu : hi - lo;
  // dynamic walk: worth it (and only meaningful) when every CU gets several tiles
  unsigned* queue = nullptr;
  if (dynamic && hi - lo >= 4ll * n_cu) {
    if (int rc = queue_acquire(&queue, stream)) return rc;
    grid = n_cu;
  }
  int stagger_arg = 0;
#ifdef AC_TILE_TIMING
  stagger_arg = tt_counter()++ & 7;
#endif
  hipLaunchKernelGGL((kernel<Epi, FLUSH>), dim3((unsigned)grid), dim3(THREADS), LDS_BYTES + LDS_EXTRA, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, (int)lo, (int)hi, stagger_arg, epi, queue);
  return ac_check_launch("gemm_hl2");
}

}  // namespace gemm2
