// Split-fp16 MFMA GEMM, 256x128x32 tiles, 8 waves, NO block-wide barrier in the k loop (EXPERIMENTAL,
// ac_set_gemm_policy 106 for the all-pairs kernel, -DAC_EXPERIMENTAL builds).
//
// Why: the wait-time stamps of gemm_hl5.h show what a barrier costs here.  The two
// waves of a SIMD share the MFMA pipe and the arbiter serves the older one first; it reaches the barrier
// early and parks, and whatever the younger wave still has to do besides MFMAs then runs with the pipe
// idle: stage time = 2 x MFMA time + the non-MFMA work of the wave that arrives last.  Here the waves
// synchronise through two monotonic counters in LDS instead, so the leader is only ever held up by a true
// data dependency and keeps issuing MFMAs while its partner reads fragments:
//   landed  +1 by each of the four loader waves (0-3) when its share of the next stage has landed
//           (stage s is complete when landed >= 4 (s + 1))
//   passed  +1 by every wave at the X point of a stage (before its last row step), i.e. when it no longer
//           reads A(t) and took B(t) into registers a stage ago (all past X(t) when passed >= 8 (t + 1))
// A wave enters the last row step of stage t - which reads A(t+1) and B(t+1) - once landed >= 4 (t + 2).
// A loader refills the slots of stage t with stage t+3 once passed >= 8 (t + 1); it tries at X(t) and, if
// its partners are not there yet, again after each of the following row steps, so it never parks for
// them unless it gets a whole stage ahead.  With three slots per operand (144 KB) a refill that is issued
// up to a stage late still lands a stage before it is needed.
//
// Everything else is gemm_hl5.h at NT = 4: wave (wm, wn) of a 4 x 2 grid owns 64 x 64, B fragments
// resident and re-read in place in the last row step, A fragment streamed, optional two-level
// accumulation (FLUSH), symmetric / swapped epilogues.  All spins are bounded: a protocol error gives
// wrong numbers, not a hung GPU.
#pragma once
#include "gemm_hl5.h"

namespace gemm6 {

using gemm::EpiTraits;
using gemm2::gptr_t;
using gemm2::lptr_t;
using gemm2::tile_off;
using Frag = gemm::FragT<16>;

constexpr int BM = 256, BN = 128, BK = 32, THREADS = 512, NT = 4, SLOTS = 3;
constexpr int HALF_A = BM * 128, HALF_B = BN * 128;
constexpr int CNT_OFF = SLOTS * (HALF_A + HALF_B);      // 144 KB of rings, then the counters
constexpr int LDS_BYTES = CNT_OFF + 64;
constexpr int FLUSH_STAGES = 16;
constexpr int SPIN_MAX = 1 << 16;

template <class Epi, bool FLUSH>
__global__ __launch_bounds__(THREADS, 1) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int id_lo, int id_hi, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const Aring = smem;
  uint8_t* const Bring = smem + SLOTS * HALF_A;
  volatile unsigned* const cnt = reinterpret_cast<volatile unsigned*>(smem + CNT_OFF);   // [0] landed, [1] passed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr bool SWAP = gemm::HasSwap<Epi>::value;
  const int nk = K / BK;
  const size_t ld = (size_t)K * 4;

  int tm = 0, tn = 0;
  const int id = id_lo + blockIdx.x;
  if (id >= id_hi || !gemm2::tile_of(id, tiles_m, tiles_n, group_m, M, N, epi, tm, tn)) return;   // block-uniform
  const int m0 = tm * BM, n0 = tn * BN;

  const bool loader = wave < 4;
  if (loader) __builtin_amdgcn_s_setprio(2);
  // DMA map as in gemm_hl5.h: 1 KB wave-instructions of 8 rows; loader w issues g = 4j + w
  // (A: j = 0..7, B: j = 0..3); addresses rebuilt per instruction.
  const int row0 = (wave & 3) * 8 + (lane >> 3);
  const unsigned lc16 = (unsigned)((lane & 7) ^ ((row0 >> 1) & 7)) * 16u;
  auto dma_stage = [&](int t, int slot) {          // A(t) then B(t) into ring slot `slot`: 12 instructions
    const uint8_t* baseA = A + (size_t)t * 128 + lc16;
    const uint8_t* baseB = B + (size_t)t * 128 + lc16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t off = (size_t)min(m0 + row0 + 32 * j, M - 1) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)(baseA + off), (lptr_t)(Aring + slot * HALF_A + (j * 4 + (wave & 3)) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + 32 * j;
      const int brow = SWAP ? ((row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3)) : row;
      const size_t off = (size_t)min(n0 + brow, N - 1) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)(baseB + off), (lptr_t)(Bring + slot * HALF_B + (j * 4 + (wave & 3)) * 1024), 16, 0, 0);
    }
  };
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  auto signal = [&](int idx) {                     // one increment per wave
    if (lane == 0) atomicAdd(const_cast<unsigned*>(cnt) + idx, 1u);
  };
  auto reached = [&](int idx, int target) -> bool {
    const unsigned v = __builtin_amdgcn_readfirstlane(cnt[idx]);
    return (int)(v - (unsigned)target) >= 0;
  };
  auto wait_for = [&](int idx, int target) {
    for (int it = 0; it < SPIN_MAX && !reached(idx, target); ++it) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  const int a_hi = tile_off(wm * 64 + (lane & 15), (lane >> 4)), a_lo = tile_off(wm * 64 + (lane & 15), 4 + (lane >> 4));
  const int b_hi = tile_off(wn * 64 + (lane & 15), (lane >> 4)), b_lo = tile_off(wn * 64 + (lane & 15), 4 + (lane >> 4));

  f32x4 acc[4][NT], tot[FLUSH ? 4 : 1][FLUSH ? NT : 1];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[mi][ni][i] = 0.f; if constexpr (FLUSH) tot[mi][ni][i] = 0.f; }

  // prologue: stages 0, 1, 2 go out; stage 0 must have landed (counted as 4 landed signals)
  if (tid == 0) { cnt[0] = 4u; cnt[1] = 0u; }
  if (loader) {
    dma_stage(0, 0);
    if (nk > 1) dma_stage(1, 1);
    if (nk > 2) dma_stage(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  barrier();
  half8 bh[NT], bl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bh[t] = *reinterpret_cast<const half8*>(Bring + b_hi + t * 2048);
    bl[t] = *reinterpret_cast<const half8*>(Bring + b_lo + t * 2048);
  }
  half8 ah = *reinterpret_cast<const half8*>(Aring + a_hi), al = *reinterpret_cast<const half8*>(Aring + a_lo);

  auto mfma3 = [&](f32x4& c, const half8& xh, const half8& xl, const half8& yh, const half8& yl) {
    if constexpr (SWAP) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, xh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, xh, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, c, 0, 0, 0);
    }
  };

  int sa = 0;                                      // ring slot of the current stage (kt % 3)
  int next_refill = 0;                             // loaders: refill r = stage r+3 into the slot of stage r
  int x_passed = -1;                               // last stage whose X point this wave has passed
  // refill `next_refill` if it exists, this wave is past its X point and so is everybody else
  auto try_refill = [&]() {
    if (next_refill + 3 < nk && next_refill <= x_passed && reached(1, 8 * (next_refill + 1))) {
      asm volatile("" ::: "memory");
      dma_stage(next_refill + 3, next_refill % 3);
      ++next_refill;
    }
  };
  for (int kt = 0; kt < nk; ++kt) {
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    const int sa1 = sa == 2 ? 0 : sa + 1;
    const uint8_t* As = Aring + sa * HALF_A;
    const uint8_t* An = Aring + sa1 * HALF_A;
    const uint8_t* Bn = Bring + sa1 * HALF_B;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (mi == 3 && more1) {                      // ---- X(kt)
        if (loader) {
          // every refill up to kt-1 must be out before the counted wait below (normally it is)
          while (next_refill < kt && next_refill + 3 < nk) {
            wait_for(1, 8 * (next_refill + 1));
            dma_stage(next_refill + 3, next_refill % 3);
            ++next_refill;
          }
          // own share of stage kt+1 landed; stage kt+2 (12 instructions, the youngest) may still fly
          if (more2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          signal(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of A(kt) (and of B(kt), long ago) done
        signal(1);
        x_passed = kt;
        wait_for(0, 4 * (kt + 2));                 // everybody's share of stage kt+1 has landed
        if (loader) try_refill();
      }
      half8 nh = ah, nl = al;
      if (mi < 3) {
        nh = *reinterpret_cast<const half8*>(As + a_hi + (mi + 1) * 2048);
        nl = *reinterpret_cast<const half8*>(As + a_lo + (mi + 1) * 2048);
      } else if (more1) {
        nh = *reinterpret_cast<const half8*>(An + a_hi);
        nl = *reinterpret_cast<const half8*>(An + a_lo);
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        mfma3(acc[mi][ni], ah, al, bh[ni], bl[ni]);
        if (mi == 3 && more1) {
          bh[ni] = *reinterpret_cast<const half8*>(Bn + b_hi + ni * 2048);
          bl[ni] = *reinterpret_cast<const half8*>(Bn + b_lo + ni * 2048);
        }
      }
      ah = nh; al = nl;
      if (loader) try_refill();                    // deferred refill: the partners may have caught up
    }
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
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  barrier();                                       // ring reads done before an epilogue reuses LDS
  if constexpr (FLUSH) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mi][ni][i] += tot[mi][ni][i];
  }

  Frag f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.acc[mi][ni] = acc[mi][ni];
  const int m_wave = m0 + wm * 64, n_wave = n0 + wn * 64;
  if constexpr (SWAP) {
    epi.swapped(f, m_wave, n_wave, lane);
  } else if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {
      const int rb = 2 * tm + (wm >> 1), cb = tn;   // see gemm_hl2.h
      if (rb <= cb) epi(f, m_wave, n_wave, lane);
      if (rb < cb) epi.transposed(f, m_wave, n_wave, lane, reinterpret_cast<float*>(smem) + wave * (32 * 65));
    } else {
      epi(f, m_wave, n_wave, lane);
    }
  } else {
    epi(f, m_wave, n_wave, lane);
  }
}

template <class Epi, bool FLUSH = false>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 8, int part = 0, int nparts = 1) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl6: K=%d must be a positive multiple of 32", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl6: bad part %d/%d", part, nparts);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // folded upper triangle (gemm2::tile_of)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl6: symmetric epilogue needs M == N");
      const int Tm = tiles_m, Tn = tiles_n;
      tiles_m = (Tm + 1) / 2;
      tiles_n = 2 * Tn - 2 * Tm + 2;
    }
  }
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl6: grid too large");
  static bool attr_set = false;                    // per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel<Epi, FLUSH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  long long lo = (total * part / nparts) & ~7LL;
  long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  hipLaunchKernelGGL((kernel<Epi, FLUSH>), dim3((unsigned)(hi - lo)), dim3(THREADS), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, (int)lo, (int)hi, epi);
  return ac_check_launch("gemm_hl6");
}

}  // namespace gemm6
