// Split-fp16 MFMA GEMM, second generation: 256x128x32 tiles, 8 waves, LDS-DMA ring.
//
// Same contract and epilogue interface as gemm_hl.h (C[m][n] = sum_k A[m][k] B[n][k] on hl32
// operands, M on the accumulator registers, N on the lanes, 64x64 per wave), different staging:
//   * 512 threads = 8 waves as 4 (M) x 2 (N); one workgroup per CU, two waves per SIMD;
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB = 8 full 128-byte
//     lines per wave-instruction), no VGPR round trip and no ds_write;
//   * a ring of 3 stages (3 x 48 KB): the loads of stage t+2 are issued while stage t is being
//     multiplied; the only wait in the loop is a counted s_waitcnt vmcnt(6) (one stage's 6
//     wave-instructions stay in flight) followed by one raw s_barrier per stage;
//   * LDS image of a tile: [row][128 B] = the hl32 line of that row; its eight 16-byte chunks
//     are XOR-swizzled with (row>>1)&7 - applied to the per-lane SOURCE address, because the DMA
//     destination is lane-linear - which makes every ds_read_b128 lane group conflict-free.
// Bytes staged per MFMA FLOP drop by 25 % against the 128x128 tile, and the per-stage
// ds_write + vmcnt(0) + barrier sequence of the register-staged kernel disappears.
#pragma once
#include "gemm_hl.h"
#include <type_traits>

namespace gemm2 {

using gemm::EpiTraits;
using gemm::Frag;

constexpr int BM = 256, BN = 128, BK = 32, NSTAGE = 3, THREADS = 512;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 48 KB
constexpr int LDS_BYTES = NSTAGE * STAGE;                                           // 144 KB
constexpr int FLUSH_STAGES = 16;

__device__ inline int tile_off(int row, int lc) { return row * 128 + ((lc ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <class Epi, bool FLUSH, int SCHED>
__global__ __launch_bounds__(THREADS, 2) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int block_offset, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  gemm::raster(blockIdx.x + block_offset, tiles_m, tiles_n, group_m, tm, tn);
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      // C = C^T on a grid of 256-row x 128-column tiles: in units of 128-blocks a tile holds row
      // blocks {2tm, 2tm+1} and column block tn; the upper triangle needs the tiles with tn >= 2tm.
      // Tile row p has Tn - 2p of them; rows p and Tm-1-p together always have 2Tn - 2Tm + 2, so the
      // launch grid is that folded rectangle and every XCD gets the same amount of work.
      const int Tm = (M + BM - 1) / BM, Tn = (N + BN - 1) / BN;
      const int p = tm, c = tn, n1 = Tn - 2 * p;
      if (c < n1) { tm = p; tn = 2 * p + c; }
      else {
        if (2 * p == Tm - 1) return;                // odd Tm: the middle row is paired with itself
        tm = Tm - 1 - p; tn = 2 * tm + (c - n1);
      }
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA map: wave-instruction g (0..47) fills LDS bytes [g*1024, +1024) of a stage:
  // g < 32 -> rows 8g..8g+7 of the A tile, else rows 8(g-32).. of the B tile.  Wave w issues
  // g = j*8 + w, j = 0..5.  Lane l lands in slot l: row (l>>3), physical chunk (l&7).
  const size_t ld = (size_t)K * 4;
  const uint8_t* src[6];
  int dst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int g = j * 8 + wave;
    const bool isA = j < 4;
    const int row = (isA ? g : g - 32) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    const uint8_t* base = isA ? A + (size_t)min(m0 + row, M - 1) * ld : B + (size_t)min(n0 + row, N - 1) * ld;
    src[j] = base + lc * 16;
    dst[j] = g * 1024;
  }
  auto issue = [&](int kt) {
    uint8_t* stage = smem + (kt % NSTAGE) * STAGE;
    const size_t ko = (size_t)kt * 128;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + ko), (lptr_t)(stage + dst[j]), 16, 0, 0);
  };

  Frag f, tot;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) { f.acc[mi][ni][i] = 0.f; if (FLUSH) tot.acc[mi][ni][i] = 0.f; }

  // per-lane fragment offsets inside a stage: [kstep][tile][plane]
  int a_off[2][2][2], b_off[2][2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        a_off[s][t][p] = tile_off(wm * 64 + t * 32 + r, p * 4 + s * 2 + h);
        b_off[s][t][p] = A_BYTES + tile_off(wn * 64 + t * 32 + r, p * 4 + s * 2 + h);
      }

  const int nk = K / BK;
  if constexpr (SCHED == 2) {
    // ---- staggered schedule with REGISTER staging (global -> VGPR -> ds_write_b128), 2 LDS slots.
    // Same two-barrier, half-a-stage-apart structure as SCHED 1, but the loads of stage kt+2 are plain
    // global_load_dwordx4 (a few cycles of issue each) and land in LDS one stage later with 6
    // ds_write_b128 per thread; an LDS-DMA wave-instruction costs the issuing wave ~100 cycles, 600 per
    // stage, which made P1 (loads + fragment reads) longer than P2 (24 MFMAs).
    //   P1(kt): read the 16 fragments of stage kt (slot kt&1); write the staged registers of stage
    //           kt+1 into slot (kt+1)&1 (last read in P1(kt-1): both groups left it a barrier ago);
    //           issue the global loads of stage kt+2 into the same registers.
    //   P2(kt): 24 MFMAs on registers.
    const bool groupB = wave >= 4;
    const uint8_t* gsrc[6];
    int ldst[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int c = j * THREADS + tid;
      const bool isA = j < 4;
      const int cc = isA ? c : c - 4 * THREADS;
      const int row = cc >> 3, lc = cc & 7;
      const uint8_t* base = isA ? A + (size_t)min(m0 + row, M - 1) * ld : B + (size_t)min(n0 + row, N - 1) * ld;
      gsrc[j] = base + lc * 16;
      ldst[j] = (isA ? 0 : A_BYTES) + tile_off(row, lc);
    }
    u32x4 st[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) st[j] = *reinterpret_cast<const u32x4*>(gsrc[j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(smem + ldst[j]) = st[j];
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) st[j] = *reinterpret_cast<const u32x4*>(gsrc[j] + 128);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // #0: stage 0 visible to everyone
    asm volatile("" ::: "memory");
    if (groupB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    for (int kt = 0; kt < nk; ++kt) {
      const uint8_t* cur = smem + (kt & 1) * STAGE;
      half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ah[s][t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][0]);
          al[s][t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][1]);
          bh[s][t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][0]);
          bl[s][t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][1]);
        }
      if (kt + 1 < nk) {
        uint8_t* nxt = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(nxt + ldst[j]) = st[j];
      }
      if (kt + 2 < nk) {
        const size_t ko = (size_t)(kt + 2) * 128;
#pragma unroll
        for (int j = 0; j < 6; ++j) st[j] = *reinterpret_cast<const u32x4*>(gsrc[j] + ko);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], bl[s][ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][mi], bh[s][ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], bh[s][ni], f.acc[mi][ni], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      if (FLUSH && (kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot.acc[mi][ni][i] += f.acc[mi][ni][i]; f.acc[mi][ni][i] = 0.f; }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!groupB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  } else {
    issue(0);
    if (nk > 1) issue(1);
  }

  if constexpr (SCHED == 2) {
  } else if constexpr (SCHED == 1) {
    // ---- staggered schedule: two barriers per stage, the halves of the workgroup half a stage apart.
    // A stage is split into P1 = {16 fragment reads, 6 DMA issues of stage kt+2} and P2 = {24 MFMAs on
    // registers}.  Waves 0-3 (group A) and waves 4-7 (group B; w and w+4 share a SIMD) run the same
    // code one barrier apart, so on every SIMD one wave is in P2 while its partner is in P1: the matrix
    // pipe no longer idles while all eight waves issue loads in lockstep.
    //   barrier #:   0      1        2        3        4
    //   A:            P1(0) | P2(0)  | P1(1)  | P2(1)  | ...
    //   B:          (wait)  | P1(0)  | P2(0)  | P1(1)  | ...
    // Visibility: every wave has waited for its own DMA share of stage kt+1 before the barrier that
    // precedes group A's P1(kt+1) (A: end of P2(kt), B: end of P1(kt)).  Reuse: the ring slot of stage
    // kt+2 was last read in P1(kt-1), which both groups have left (lgkmcnt(0) + barrier) before either
    // issues into it.
    const bool groupB = wave >= 4;
    auto wait_share = [&](int kt) {                 // own share of stage kt+1 landed
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // #0: stage 0 visible to everyone
    asm volatile("" ::: "memory");
    if (groupB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    for (int kt = 0; kt < nk; ++kt) {
      const uint8_t* cur = smem + (kt % NSTAGE) * STAGE;
      half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ah[s][t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][0]);
          al[s][t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][1]);
          bh[s][t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][0]);
          bl[s][t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][1]);
        }
      if (kt + 2 < nk) issue(kt + 2);
      if (groupB) wait_share(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], bl[s][ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][mi], bh[s][ni], f.acc[mi][ni], 0, 0, 0);
            f.acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], bh[s][ni], f.acc[mi][ni], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
      if (FLUSH && (kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot.acc[mi][ni][i] += f.acc[mi][ni][i]; f.acc[mi][ni][i] = 0.f; }
      }
      if (!groupB) wait_share(kt);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!groupB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed for this wave once at most the 6 loads of stage kt+1 are still in flight;
    // the barrier makes every wave's share visible and proves everyone is done reading stage kt-1.
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) issue(kt + 2);           // overwrites the ring slot of stage kt-1
    // (tried: spreading the 6 DMA issues between the MFMAs with sched_group_barrier - 7 % slower on the
    //  ViT GEMMs and 2x slower with the flush accumulators live; the block issue stays)

    const uint8_t* cur = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][0]);
        al[t] = *reinterpret_cast<const half8*>(cur + a_off[s][t][1]);
        bh[t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][0]);
        bl[t] = *reinterpret_cast<const half8*>(cur + b_off[s][t][1]);
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
    if (FLUSH && (kt & (FLUSH_STAGES - 1)) == FLUSH_STAGES - 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int i = 0; i < 16; ++i) { tot.acc[mi][ni][i] += f.acc[mi][ni][i]; f.acc[mi][ni][i] = 0.f; }
    }
  }
  if (FLUSH) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 16; ++i) f.acc[mi][ni][i] += tot.acc[mi][ni][i];
  }
  if constexpr (EpiTraits<Epi>::kStaged) {
    // every wave is past its last ring read after this barrier (no DMA is in flight: vmcnt(0) above)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float* t = reinterpret_cast<float*>(smem) + wave * (64 * gemm::EPI_PITCH);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          t[(mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * gemm::EPI_PITCH + ni * 32 + r] = f.acc[mi][ni][i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    epi.staged(t, m0 + wm * 64, n0 + wn * 64, lane);
  } else if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      // this wave's 64 rows lie in 128-row block rb, its columns in block cb = tn:
      // rb < cb: off-diagonal, reduce both ways; rb == cb: diagonal block, one way; rb > cb: the
      // lower half of a diagonal tile - its mirror image is computed by tile (tm, tn+1)
      const int rb = 2 * tm + (wm >> 1), cb = tn;
      __builtin_amdgcn_s_barrier();                 // every wave is done with the ring
      asm volatile("" ::: "memory");
      if (rb <= cb) epi(f, m0 + wm * 64, n0 + wn * 64, lane);
      if (rb < cb)
        epi.transposed(f, m0 + wm * 64, n0 + wn * 64, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65));
    } else {
      epi(f, m0 + wm * 64, n0 + wn * 64, lane);
    }
  } else {
    epi(f, m0 + wm * 64, n0 + wn * 64, lane);
  }
}

template <class Epi, bool FLUSH = false, int SCHED = 0>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 8, int part = 0, int nparts = 1) {
  // part / nparts: launch only the part-th of nparts contiguous slices of the block-id space (slice
  // boundaries are multiples of 8 so the XCD round-robin stays aligned) - used to split one
  // symmetric all-pairs problem over several GPUs.
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl2: K=%d must be a positive multiple of 32", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl2: bad part %d/%d", part, nparts);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // folded upper triangle (see the kernel)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl2: symmetric epilogue needs M == N");
      const int Tm = tiles_m, Tn = tiles_n;
      tiles_m = (Tm + 1) / 2;
      tiles_n = 2 * Tn - 2 * Tm + 2;
    }
  }
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl2: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel<Epi, FLUSH, SCHED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  long long lo = (total * part / nparts) & ~7LL;
  long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  hipLaunchKernelGGL((kernel<Epi, FLUSH, SCHED>), dim3((unsigned)(hi - lo)), dim3(THREADS), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, (int)lo, epi);
  return ac_check_launch("gemm_hl2");
}

}  // namespace gemm2
