// Split-fp16 MFMA GEMM, third generation (EXPERIMENTAL, ac_set_gemm_policy 103 in -DAC_EXPERIMENTAL builds): 256x256x32 tiles,
// 4 waves, ONE wave per SIMD with the whole 512-register file.
//
// gemm_hl2.h is co-limited by operand delivery (18.4 ms of global->LDS traffic
// against 19.9 ms of MFMAs on the all-pairs problem).  A 256x256 tile halves the bytes per FLOP.
// It does not fit the 8-wave / 2-per-SIMD layout (128 accumulator registers per wave plus fragments
// exceed 256), so this kernel gives each wave 128x128 of the tile:
//   acc   8x8 tiles of 16x16 (v_mfma_f32_16x16x32_f16)          256 registers
//   B     all 8 column fragments (hi + lo) of the stage           64
//   A     one row fragment (hi + lo), double buffered             16   (24 MFMAs per fragment cover the
//                                                                       next fragment's LDS read)
// With no SIMD partner a wave has to hide its own loads under its own MFMAs, so the k loop is one
// continuous stream of "row steps" (one A fragment x 8 B fragments = 24 MFMAs):
//   * the next A fragment is read during the current step, across stage boundaries too;
//   * in the LAST step of a stage each B fragment is re-read for the next stage right after its final
//     MFMA, ~21 MFMAs before its first use - no fragment-load bubble and no second B register set;
//   * the stage barrier therefore sits before that last step (7/8 through the stage): from there on
//     nobody reads A(t) or the LDS copy of B(t) again;
//   * 2 LDS-DMA instructions per step: steps 7,0,1,2 fetch B(t+2 | t+1), steps 3..6 fetch A(t+2).
// LDS (160 KB): A in a 3-slot ring (two stages of latency cover), B in a 2-slot ring (one stage).
#pragma once
#include "gemm_hl2.h"

namespace gemm3 {

using gemm::EpiTraits;
using gemm2::gptr_t;
using gemm2::lptr_t;
using gemm2::tile_off;
using Frag = gemm::FragT<16>;

constexpr int BM = 256, BN = 256, BK = 32, THREADS = 256;
constexpr int HALF = 256 * 128;                 // one operand half-tile: 256 rows x 128 B
constexpr int A_SLOTS = 3, B_SLOTS = 2;
constexpr int LDS_BYTES = (A_SLOTS + B_SLOTS) * HALF;   // 160 KB

// Timing-only diagnostic builds (results are wrong): -DAC_ABLATE3=1 no DMA in the loop, 2 no fragment
// reads in the loop, 4 no MFMAs (combinable).  0 in every shipped build.
#ifndef AC_ABLATE3
#define AC_ABLATE3 0
#endif

template <class Epi>
__global__ __launch_bounds__(THREADS, 1) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const Aring = smem;
  uint8_t* const Bring = smem + A_SLOTS * HALF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned ld = (unsigned)K * 4u;
  const int nk = K / BK;

  int tm, tn;
  gemm::raster(blockIdx.x, tiles_m, tiles_n, group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // DMA map: a half-tile is 32 wave-instructions of 1 KB (8 rows each).  Wave w issues g = j*4 + w,
  // j = 0..7.  Lane l lands in slot l of the instruction: row 8g + (l>>3), physical chunk l&7, so it
  // fetches logical chunk (l&7) ^ ((row>>1)&7) of that row.  Offsets are 32-bit (launch() checks).
  unsigned offA[8], offB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const unsigned lc = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
    offA[j] = (unsigned)min(m0 + row, M - 1) * ld + lc * 16u;
    offB[j] = (unsigned)min(n0 + row, N - 1) * ld + lc * 16u;
  }
  auto dmaA = [&](int kt, int slot, int j) {
    if (AC_ABLATE3 & 1) return;
    __builtin_amdgcn_global_load_lds((gptr_t)(A + (size_t)kt * 128 + offA[j]),
                                     (lptr_t)(Aring + slot * HALF + (j * 4 + wave) * 1024), 16, 0, 0);
  };
  auto dmaB = [&](int kt, int slot, int j) {
    if (AC_ABLATE3 & 1) return;
    __builtin_amdgcn_global_load_lds((gptr_t)(B + (size_t)kt * 128 + offB[j]),
                                     (lptr_t)(Bring + slot * HALF + (j * 4 + wave) * 1024), 16, 0, 0);
  };
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // fragment offsets: tile t of this wave's 128 rows starts at base + t * 2048 bytes (16 rows); the XOR
  // swizzle term (row >> 1) & 7 is unchanged by +16 rows, so one base per plane suffices
  const int a_hi = tile_off(wm * 128 + (lane & 15), (lane >> 4)), a_lo = tile_off(wm * 128 + (lane & 15), 4 + (lane >> 4));
  const int b_hi = tile_off(wn * 128 + (lane & 15), (lane >> 4)), b_lo = tile_off(wn * 128 + (lane & 15), 4 + (lane >> 4));

  f32x4 acc[8][8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[mi][ni][i] = 0.f;

  // prologue: B(0), A(0), A(1); the first two must have landed before the fragments are read
  {
    const bool ab = AC_ABLATE3 & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(B + offB[j]), (lptr_t)(Bring + (j * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(A + offA[j]), (lptr_t)(Aring + (j * 4 + wave) * 1024), 16, 0, 0);
    if (nk > 1 && !ab) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dmaA(1, 1, j);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  barrier();
  half8 bh[8], bl[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    bh[t] = *reinterpret_cast<const half8*>(Bring + b_hi + t * 2048);
    bl[t] = *reinterpret_cast<const half8*>(Bring + b_lo + t * 2048);
  }
  half8 ah = *reinterpret_cast<const half8*>(Aring + a_hi), al = *reinterpret_cast<const half8*>(Aring + a_lo);
  if (nk > 1) { dmaB(1, 1, 0); dmaB(1, 1, 1); }            // "step 7 of stage -1"

  int sa = 0;                                    // A slot of the current stage (kt % 3)
  auto stage = [&](int kt, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;   // branch-free body while two more stages follow
    const bool more1 = STEADY || kt + 1 < nk, more2 = STEADY || kt + 2 < nk;
    const int sa1 = sa == 2 ? 0 : sa + 1;        // (kt + 1) % 3
    const int sa2 = sa == 0 ? 2 : sa - 1;        // (kt + 2) % 3
    const uint8_t* As = Aring + sa * HALF;
    const uint8_t* An = Aring + sa1 * HALF;
    const uint8_t* Bn = Bring + ((kt + 1) & 1) * HALF;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      if (mi == 7 && more1) {
        // own share of A(kt+1) and B(kt+1) landed (A(kt+2), the 8 youngest, may still fly); after the
        // barrier everybody's has, and nobody reads A(kt) or the LDS copy of B(kt) any more
        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
      }
      half8 nh = ah, nl = al;
      if (!(AC_ABLATE3 & 2)) {
        if (mi < 7) {
          nh = *reinterpret_cast<const half8*>(As + a_hi + (mi + 1) * 2048);
          nl = *reinterpret_cast<const half8*>(As + a_lo + (mi + 1) * 2048);
        } else if (more1) {
          nh = *reinterpret_cast<const half8*>(An + a_hi);
          nl = *reinterpret_cast<const half8*>(An + a_lo);
        }
      }
      // B(kt+1): second .. fourth quarter in steps 0..2 (the first went out in step 7 of stage kt-1)
      if (mi < 3 && more1) { dmaB(kt + 1, (kt + 1) & 1, 2 * mi + 2); dmaB(kt + 1, (kt + 1) & 1, 2 * mi + 3); }
      if (mi >= 3 && mi < 7 && more2) { dmaA(kt + 2, sa2, 2 * (mi - 3)); dmaA(kt + 2, sa2, 2 * (mi - 3) + 1); }
      if (mi == 7 && more2) { dmaB(kt + 2, kt & 1, 0); dmaB(kt + 2, kt & 1, 1); }
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        if (!(AC_ABLATE3 & 4)) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
        }
        if (mi == 7 && more1 && !(AC_ABLATE3 & 2)) {
          bh[ni] = *reinterpret_cast<const half8*>(Bn + b_hi + ni * 2048);
          bl[ni] = *reinterpret_cast<const half8*>(Bn + b_lo + ni * 2048);
        }
      }
      ah = nh; al = nl;
    }
    sa = sa1;
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) stage(kt, std::true_type{});
  for (; kt < nk; ++kt) stage(kt, std::false_type{});
  barrier();                                     // ring reads done before the epilogue reuses LDS

  // epilogue: the 128x128 accumulator as four 64x64 fragments of the common epilogue interface
#pragma unroll
  for (int mo = 0; mo < 2; ++mo)
#pragma unroll
    for (int no = 0; no < 2; ++no) {
      Frag f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) f.acc[mi][ni] = acc[mo * 4 + mi][no * 4 + ni];
      const int m_wave = m0 + wm * 128 + mo * 64, n_wave = n0 + wn * 128 + no * 64;
      if constexpr (EpiTraits<Epi>::kStaged) {
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      } else {
        epi(f, m_wave, n_wave, lane);
      }
    }
}

// True when the shape fits this kernel's 32-bit operand offsets.
inline bool fits(int M, int N, int K) {
  return (unsigned long long)(M > N ? M : N) * (unsigned long long)K * 4ull < 0xffff0000ull;
}

template <class Epi>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 4) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl3: K=%d must be a positive multiple of 32", K);
  if (!fits(M, N, K)) return ac_fail(AC_EINVAL, "gemm_hl3: operand larger than 4 GiB");
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl3: grid too large");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel<Epi>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((kernel<Epi>), dim3((unsigned)total), dim3(THREADS), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, epi);
  return ac_check_launch("gemm_hl3");
}

}  // namespace gemm3
