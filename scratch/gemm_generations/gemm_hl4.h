// Split-fp16 MFMA GEMM, loader-specialised variants of gemm_hl2.h (EXPERIMENTAL, ac_set_gemm_policy 104 / 105).
//
// Same 256x128x32 tile and LDS-DMA ring as gemm_hl2.h, but the global->LDS traffic is issued by four
// dedicated LOADER waves (one per SIMD) instead of by the eight MFMA waves.  Ablation of gemm_hl2.h
//  showed that a wave stalls ~90 cycles per global_load_lds it issues; with 6 per
// wave per stage that is a third of the wave's stage time and it cannot all hide behind the SIMD
// partner's MFMAs.  A loader wave absorbs those stalls for its SIMD: 12 instructions per stage
// (~1100 cycles) against 1536 MFMA cycles.  Three waves per SIMD leave 168 registers per wave, so
// there is no second accumulator here (single-level accumulation only).
//
// MODE 0 (policy 104): the staggered two-group schedule of gemm_hl2.h, two barriers per stage.
// MODE 1 (policy 105): every MFMA wave runs one continuous stream - the fragments of stage t+1 are read
//   from LDS while the MFMAs of stage t run, each into registers whose stage-t value has just had its
//   last use (A row fragments after their row; B fragments in the last row), plus 24 spare registers for
//   the three fragments that die too late.  One barrier per stage.
#pragma once
#include "gemm_hl2.h"

namespace gemm4 {

using gemm::EpiTraits;
using gemm2::gptr_t;
using gemm2::lptr_t;
using gemm2::tile_of;
using gemm2::tile_off;
using Frag = gemm::FragT<16>;

constexpr int BM = 256, BN = 128, BK = 32, NSTAGE = 3;
constexpr int CONSUMERS = 8, LOADERS = 4, THREADS = (CONSUMERS + LOADERS) * 64;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 48 KB
constexpr int LDS_BYTES = NSTAGE * STAGE;                                           // 144 KB

// MODE 0 barrier numbering: #0 follows the prologue; interval I_n lies between barriers n and n+1.
//   group A (waves 0-3): fragment reads of stage kt in I_2kt,   MFMAs in I_2kt+1
//   group B (waves 4-7): fragment reads of stage kt in I_2kt+1, MFMAs in I_2kt+2
//   loaders:             half h of stage t goes out in I_(2t+h-4) (its ring slot, stage t-3's, was last
//                        read in I_2t-5) and must have landed by the end of I_2t-1.
//   Every wave executes 2nk + 2 barriers.
// MODE 1: barrier #0 = "stage 0 landed"; barrier S(kt), kt = 0..nk-1, opens stage kt = "stage kt+1 landed,
//   nobody reads the LDS copy of stage kt any more" (it was read during stage kt-1).  After S(kt) the
//   loaders refill slot kt % 3 with stage kt+3.  Every wave executes nk + 1 barriers.
template <class Epi, int MODE>
__global__ __launch_bounds__(THREADS, 1) void kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                     int M, int N, int K, int tiles_m, int tiles_n, int group_m,
                                                     int id_lo, int id_hi, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t ld = (size_t)K * 4;
  const int nk = K / BK;
  auto barrier = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  constexpr bool SWAP = gemm::HasSwap<Epi>::value;   // accumulate C^T tiles (see gemm_hl.h)

  int tm = 0, tn = 0;
  const int id = id_lo + blockIdx.x;
  if (id >= id_hi || !tile_of(id, tiles_m, tiles_n, group_m, M, N, epi, tm, tn)) return;   // block-uniform

  if (wave >= CONSUMERS) {
    // ---------------------------------------------------------------- loader
    // wave-instruction g (0..47) fills LDS bytes [g*1024, +1024) of a stage: g < 32 -> rows 8g..8g+7 of
    // the A tile, else rows 8(g-32).. of the B tile.  Loader w issues g = 4j + w, j = 0..11 (half 0:
    // j < 6).  Lane l lands in slot l: row (l>>3), physical chunk (l&7).
    const int w = wave - CONSUMERS;
    const uint8_t* src[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int g = j * 4 + w;
      const bool isA = j < 8;
      const int row = (isA ? g : g - 32) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      const int brow = SWAP ? ((row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3)) : row;
      const uint8_t* base = isA ? A + (size_t)min(tm * BM + row, M - 1) * ld : B + (size_t)min(tn * BN + brow, N - 1) * ld;
      src[j] = base + lc * 16;
    }
    auto issue_half = [&](int t, int half) {         // `half` is a literal at every call site
      uint8_t* stage = smem + (t % NSTAGE) * STAGE;
      const size_t ko = (size_t)t * 128;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int jj = half * 6 + j;
        __builtin_amdgcn_global_load_lds((gptr_t)(src[jj] + ko), (lptr_t)(stage + (jj * 4 + w) * 1024), 16, 0, 0);
      }
    };
    if constexpr (MODE == 0) {
      issue_half(0, 0);
      issue_half(0, 1);
      if (nk > 1) {
        issue_half(1, 0);
        issue_half(1, 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      barrier();                                      // #0
      for (int n = 0; n <= 2 * nk; n += 2) {
        const int t = (n + 4) >> 1;
        if (t < nk) issue_half(t, 0);
        barrier();                                    // n + 1
        if (n + 1 > 2 * nk) break;
        if (t < nk) issue_half(t, 1);
        const int s = (n + 2) >> 1;                   // stage that must be complete after this interval
        if (s < nk) {
          if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier();                                    // n + 2
      }
    } else {
      const int pre = min(nk, 3);
      for (int t = 0; t < pre; ++t) { issue_half(t, 0); issue_half(t, 1); }
      if (pre == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (pre == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier();                                      // #0: stage 0 landed
      for (int kt = 0; kt < nk; ++kt) {
        // issued so far: stages <= kt+2; stage kt+1 must have landed before S(kt)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();                                    // S(kt)
        if (kt + 3 < nk) { issue_half(kt + 3, 0); issue_half(kt + 3, 1); }
      }
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wm = wave >> 1, wn = wave & 1;
  // fragment offsets inside a stage: 16-row tile t of this wave's 64 rows is base + t * 2048 (the XOR
  // swizzle term (row >> 1) & 7 is unchanged by +16 rows)
  const int a_hi = tile_off(wm * 64 + (lane & 15), (lane >> 4)), a_lo = tile_off(wm * 64 + (lane & 15), 4 + (lane >> 4));
  const int b_hi = A_BYTES + tile_off(wn * 64 + (lane & 15), (lane >> 4));
  const int b_lo = A_BYTES + tile_off(wn * 64 + (lane & 15), 4 + (lane >> 4));

  const int m0 = tm * BM, n0 = tn * BN;
  Frag f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int i = 0; i < 4; ++i) f.acc[mi][ni][i] = 0.f;

  half8 ah[4], al[4], bh[4], bl[4];
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
  if constexpr (MODE == 0) {
    const bool groupB = wave >= 4;
    barrier();                                        // #0: stage 0 visible to everyone
    if (groupB) barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const uint8_t* cur = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const half8*>(cur + a_hi + t * 2048);
        al[t] = *reinterpret_cast<const half8*>(cur + a_lo + t * 2048);
        bh[t] = *reinterpret_cast<const half8*>(cur + b_hi + t * 2048);
        bl[t] = *reinterpret_cast<const half8*>(cur + b_lo + t * 2048);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) mfma3(f.acc[mi][ni], ah[mi], al[mi], bh[ni], bl[ni]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
    }
    if (!groupB) barrier();                           // both groups are past their last ring read
  } else {
    // spare registers: the fragments whose stage-t value dies in the last row (A row 3, B columns 2, 3)
    // get their stage-t+1 value into the other set; the two sets swap roles every stage
    half8 sah, sal, sbh[2], sbl[2];
    barrier();                                        // #0
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = *reinterpret_cast<const half8*>(smem + a_hi + t * 2048);
      al[t] = *reinterpret_cast<const half8*>(smem + a_lo + t * 2048);
      bh[t] = *reinterpret_cast<const half8*>(smem + b_hi + t * 2048);
      bl[t] = *reinterpret_cast<const half8*>(smem + b_lo + t * 2048);
    }
    int slot = 0;
    auto stage = [&](int kt, auto parity, auto steady) {
      constexpr int P = decltype(parity)::value;
      constexpr bool STEADY = decltype(steady)::value;   // branch-free body while another stage follows
      half8& a3h = P ? sah : ah[3];      half8& a3l = P ? sal : al[3];
      half8& n3h = P ? ah[3] : sah;      half8& n3l = P ? al[3] : sal;
      half8& b2h = P ? sbh[0] : bh[2];   half8& b2l = P ? sbl[0] : bl[2];
      half8& b3h = P ? sbh[1] : bh[3];   half8& b3l = P ? sbl[1] : bl[3];
      half8& m2h = P ? bh[2] : sbh[0];   half8& m2l = P ? bl[2] : sbl[0];
      half8& m3h = P ? bh[3] : sbh[1];   half8& m3l = P ? bl[3] : sbl[1];
      const bool more = STEADY || kt + 1 < nk;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();                                      // S(kt)
      slot = slot == 2 ? 0 : slot + 1;
      const uint8_t* nxt = smem + slot * STAGE;       // stage kt+1
      auto rd = [&](int off) { return *reinterpret_cast<const half8*>(nxt + off); };
      // row 0 (+ the three spare-set reads)
      if (more) { n3h = rd(a_hi + 3 * 2048); n3l = rd(a_lo + 3 * 2048); }
      mfma3(f.acc[0][0], ah[0], al[0], bh[0], bl[0]);
      if (more) { m2h = rd(b_hi + 2 * 2048); m2l = rd(b_lo + 2 * 2048); }
      mfma3(f.acc[0][1], ah[0], al[0], bh[1], bl[1]);
      if (more) { m3h = rd(b_hi + 3 * 2048); m3l = rd(b_lo + 3 * 2048); }
      mfma3(f.acc[0][2], ah[0], al[0], b2h, b2l);
      mfma3(f.acc[0][3], ah[0], al[0], b3h, b3l);
      // row 1; A row 0 is dead
      if (more) { ah[0] = rd(a_hi); al[0] = rd(a_lo); }
      mfma3(f.acc[1][0], ah[1], al[1], bh[0], bl[0]);
      mfma3(f.acc[1][1], ah[1], al[1], bh[1], bl[1]);
      mfma3(f.acc[1][2], ah[1], al[1], b2h, b2l);
      mfma3(f.acc[1][3], ah[1], al[1], b3h, b3l);
      // row 2; A row 1 is dead
      if (more) { ah[1] = rd(a_hi + 2048); al[1] = rd(a_lo + 2048); }
      mfma3(f.acc[2][0], ah[2], al[2], bh[0], bl[0]);
      mfma3(f.acc[2][1], ah[2], al[2], bh[1], bl[1]);
      mfma3(f.acc[2][2], ah[2], al[2], b2h, b2l);
      mfma3(f.acc[2][3], ah[2], al[2], b3h, b3l);
      // row 3; A row 2 is dead, B columns 0 and 1 die after their MFMAs
      if (more) { ah[2] = rd(a_hi + 2 * 2048); al[2] = rd(a_lo + 2 * 2048); }
      mfma3(f.acc[3][0], a3h, a3l, bh[0], bl[0]);
      if (more) { bh[0] = rd(b_hi); bl[0] = rd(b_lo); }
      mfma3(f.acc[3][1], a3h, a3l, bh[1], bl[1]);
      if (more) { bh[1] = rd(b_hi + 2048); bl[1] = rd(b_lo + 2048); }
      mfma3(f.acc[3][2], a3h, a3l, b2h, b2l);
      mfma3(f.acc[3][3], a3h, a3l, b3h, b3l);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      stage(kt, P0{}, std::true_type{});
      stage(kt + 1, P1{}, std::true_type{});
    }
    if (kt < nk) stage(kt++, P0{}, std::false_type{});  // the last one or two stages
    if (kt < nk) stage(kt++, P1{}, std::false_type{});
    // nobody reads the ring after S(nk-1) and every DMA has landed: the epilogue may reuse LDS
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

template <class Epi, int MODE = 0>
inline int launch(const void* A, const void* B, int M, int N, int K, const Epi& epi, hipStream_t stream,
                  int group_m = 8, int part = 0, int nparts = 1) {
  if (M <= 0 || N <= 0) return AC_OK;
  if (K % BK != 0 || K <= 0) return ac_fail(AC_EINVAL, "gemm_hl4: K=%d must be a positive multiple of 32", K);
  if (nparts < 1 || part < 0 || part >= nparts) return ac_fail(AC_EINVAL, "gemm_hl4: bad part %d/%d", part, nparts);
  int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if constexpr (EpiTraits<Epi>::kSymmetric) {
    if (epi.symmetric) {                           // folded upper triangle (see gemm2::tile_of)
      if (M != N) return ac_fail(AC_EINVAL, "gemm_hl4: symmetric epilogue needs M == N");
      const int Tm = tiles_m, Tn = tiles_n;
      tiles_m = (Tm + 1) / 2;
      tiles_n = 2 * Tn - 2 * Tm + 2;
    }
  }
  long long total = (long long)tiles_m * tiles_n;
  if (total > 0x7fffffffLL) return ac_fail(AC_EINVAL, "gemm_hl4: grid too large");
  static bool attr_set = false;                    // per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel<Epi, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  long long lo = (total * part / nparts) & ~7LL;
  long long hi = (part == nparts - 1) ? total : ((total * (part + 1) / nparts) & ~7LL);
  if (hi <= lo) return AC_OK;
  hipLaunchKernelGGL((kernel<Epi, MODE>), dim3((unsigned)(hi - lo)), dim3(THREADS), LDS_BYTES, stream,
                     static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), M, N, K, tiles_m, tiles_n,
                     group_m, (int)lo, (int)hi, epi);
  return ac_check_launch("gemm_hl4");
}

}  // namespace gemm4
