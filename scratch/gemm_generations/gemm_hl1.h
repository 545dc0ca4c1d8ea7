// First generation of the split-fp16 GEMM: 128x128x32 tiles, 4 waves, register-staged double-buffered LDS,
// v_mfma_f32_32x32x16_f16, two-level accumulation.  Superseded by gemm_hl2.h / gemm_hl5.h; kept for same-box
// A/B runs in -DAC_EXPERIMENTAL builds only (not part of libac_hip.so).
#pragma once
#include "gemm_hl.h"

namespace gemm {

// Staging: each stage is 128 rows x 128 bytes per operand, register-prefetched one stage ahead and written to
// a double-buffered LDS image [plane hi|lo][128 rows][64 B] whose 16-byte chunks are XOR-swizzled with
// (row>>2)&3 so that every ds_read_b128 lane group hits 16 distinct slots.
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE = BM * 64;            // bytes of one [128][64B] plane
constexpr int STAGE = 4 * PLANE;          // A_hi, A_lo, B_hi, B_lo
constexpr int LDS_BYTES = 2 * STAGE;      // 64 KB, double buffered

__device__ inline int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

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
