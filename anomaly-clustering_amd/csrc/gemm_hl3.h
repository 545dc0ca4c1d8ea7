// Split-fp16 MFMA GEMM, third generation (EXPERIMENTAL, selected with AC_GEMM=5): 256x256x32 tiles,
// 4 waves, ONE wave per SIMD with the whole 512-register file.
//
// gemm_hl2.h is co-limited by operand delivery (18.4 ms of global->LDS traffic
// against 19.9 ms of MFMAs on the all-pairs problem).  A 256x256 tile halves the bytes per FLOP.
// It does not fit the 8-wave / 2-per-SIMD layout (128 accumulator registers per wave plus fragments
// exceed 256), so this kernel gives each wave 128x128 of the tile:
//   acc   8x8 tiles of 16x16 (v_mfma_f32_16x16x32_f16)          256 VGPRs
//   B     all 8 column fragments of the stage, hi + lo           64
//   A     one row fragment (hi + lo), double buffered            16   (streamed: 24 MFMAs per fragment
//                                                                      cover the next fragment's LDS read)
// With no SIMD partner the wave hides its own loads under its own MFMAs.
// LDS (160 KB): A in a 3-slot ring, B in a 2-slot ring of 32 KB half-tiles.  B(t) lives in registers
// for the whole stage, so its slot is recycled right after the fragment load; A(t+2) and B(t+2) are
// issued two stages ahead.  Two barriers per stage (3072 MFMA cycles).
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
constexpr int A_SLOTS = 2, B_SLOTS = 2;
constexpr int LDS_BYTES = (A_SLOTS + B_SLOTS) * HALF;   // 160 KB

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
  const size_t ld = (size_t)K * 4;
  const int nk = K / BK;

  int tm, tn;
  gemm::raster(blockIdx.x, tiles_m, tiles_n, group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // Staging map (register staged: an LDS-DMA wave-instruction costs the issuing wave 60-100 cycles, and a
  // lone wave on its SIMD has no partner to cover that; plain 16-byte global loads and ds_write_b128 fit the
  // gaps between MFMAs).  A half-tile is 2048 chunks of 16 B; thread t moves chunks c = j*256 + t, j = 0..7:
  // row c >> 3, logical chunk c & 7 (8 consecutive threads = one 128-byte line).
  const uint8_t* srcA[8];
  const uint8_t* srcB[8];
  int ldst[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = j * THREADS + tid, row = c >> 3, lc = c & 7;
    srcA[j] = A + (size_t)min(m0 + row, M - 1) * ld + lc * 16;
    srcB[j] = B + (size_t)min(n0 + row, N - 1) * ld + lc * 16;
    ldst[j] = tile_off(row, lc);
  }
  u32x4 stA[8], stB[8];
  auto gload = [&](int kt) {
    const size_t ko = (size_t)kt * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      stA[j] = *reinterpret_cast<const u32x4*>(srcA[j] + ko);
      stB[j] = *reinterpret_cast<const u32x4*>(srcB[j] + ko);
    }
  };
  auto lstore = [&](int kt) {
    uint8_t* As = Aring + (kt & 1) * HALF;
    uint8_t* Bs = Bring + (kt & 1) * HALF;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      *reinterpret_cast<u32x4*>(As + ldst[j]) = stA[j];
      *reinterpret_cast<u32x4*>(Bs + ldst[j]) = stB[j];
    }
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

  gload(0);
  lstore(0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();                                   // stage kt visible; everyone has left stage kt-1
    const uint8_t* As = Aring + (kt & 1) * HALF;
    const uint8_t* Bs = Bring + (kt & 1) * HALF;
    if (kt + 1 < nk) gload(kt + 1);              // in flight under this stage's 192 MFMAs
    half8 bh[8], bl[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      bh[t] = *reinterpret_cast<const half8*>(Bs + b_hi + t * 2048);
      bl[t] = *reinterpret_cast<const half8*>(Bs + b_lo + t * 2048);
    }
    half8 ah = *reinterpret_cast<const half8*>(As + a_hi), al = *reinterpret_cast<const half8*>(As + a_lo);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      half8 nh, nl;
      if (mi < 7) {
        nh = *reinterpret_cast<const half8*>(As + a_hi + (mi + 1) * 2048);
        nl = *reinterpret_cast<const half8*>(As + a_lo + (mi + 1) * 2048);
      }
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
      }
      if (mi < 7) { ah = nh; al = nl; }
    }
    if (kt + 1 < nk) lstore(kt + 1);             // the other slot: last read in stage kt-1
  }

  // epilogue: the 128x128 accumulator as four 64x64 fragments of the common epilogue interface
  barrier();
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

template <class Epi>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 4) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl3: K=%d must be a positive multiple of 32", K);
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
