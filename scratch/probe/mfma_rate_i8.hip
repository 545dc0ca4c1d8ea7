// Hardware probe (not part of the product): does the chip hold a higher clock on int8 MFMAs than on fp16 MFMAs?
// ("three int8 limbs with exact accumulation": the six int8 limb-pair MFMAs of a 64-column step
// take the cycles of the three fp16 passes of two 32-column steps; what would be gained is energy, i.e. clock.)
// One wave per SIMD, a 64x64 block per wave, operands re-read from LDS every step, random data:
//   f16 : 16x16x32 f16, per 32 columns 16 fragment reads + 48 MFMAs  (the shipped pattern: hl, lh, hh)
//   i8  : 16x16x64 i8,  per 64 columns 24 fragment reads + 96 MFMAs  (levels (1,1) | (1,2),(2,1) | (1,3),(2,2),(3,1))
// Both print time per MFMA; equal cycles per MFMA, so the ratio of the times is the ratio of the sustained clocks.
//   hipcc --offload-arch=gfx950 -O2 mfma_rate_i8.hip -o mfma_rate_i8 && ./mfma_rate_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256, 1) void kf16(const i32x4* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) i32x4 lds[4096];          // 64 KB of operands
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const half8* base = (const half8*)lds + ((it * 37 + wave * 11) & 47) * 64;
    half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = base[(t * 4 + 0) * 64 + lane];
      al[t] = base[(t * 4 + 1) * 64 + lane];
      bh[t] = base[(t * 4 + 2) * 64 + lane];
      bl[t] = base[(t * 4 + 3) * 64 + lane];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) s += acc[a][b][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 1) void ki8(const i32x4* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) i32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i32x4 c1[4][4], c2[4][4], c3[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) { c1[a][b][i] = 0; c2[a][b][i] = 0; c3[a][b][i] = 0; }
  for (int it = 0; it < iters; ++it) {
    const i32x4* base = lds + ((it * 37 + wave * 11) & 31) * 64;
    i32x4 a1[4], a2[4], a3[4], b1[4], b2[4], b3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a1[t] = base[(t * 6 + 0) * 64 + lane];
      a2[t] = base[(t * 6 + 1) * 64 + lane];
      a3[t] = base[(t * 6 + 2) * 64 + lane];
      b1[t] = base[(t * 6 + 3) * 64 + lane];
      b2[t] = base[(t * 6 + 4) * 64 + lane];
      b3[t] = base[(t * 6 + 5) * 64 + lane];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        c1[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[a], b1[b], c1[a][b], 0, 0, 0);
        c2[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[a], b2[b], c2[a][b], 0, 0, 0);
        c2[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2[a], b1[b], c2[a][b], 0, 0, 0);
        c3[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[a], b3[b], c3[a][b], 0, 0, 0);
        c3[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2[a], b2[b], c3[a][b], 0, 0, 0);
        c3[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3[a], b1[b], c3[a][b], 0, 0, 0);
      }
    if ((it & 63) == 63) {                       // keep the sums inside int32 whatever the bytes are (a real kernel
#pragma unroll                                   // would flush into wider totals here: 64 x 64 x 127^2 x 3 < 2^31)
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) { c1[a][b][i] >>= 12; c2[a][b][i] >>= 12; c3[a][b][i] >>= 12; }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) s += (float)(c1[a][b][i] + c2[a][b][i] + c3[a][b][i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;     // 0 random operands, 1 zeros
  std::vector<_Float16> hf(32768);
  std::vector<signed char> hi(65536);
  srand(1);
  for (auto& x : hf) x = mode ? (_Float16)0 : (_Float16)(((double)rand() / RAND_MAX - 0.5) * 0.25);
  for (auto& x : hi) x = mode ? 0 : (signed char)(rand() % 255 - 127);
  i32x4 *df, *di; float* o;
  CK(hipMalloc(&df, 65536)); CK(hipMalloc(&di, 65536)); CK(hipMalloc(&o, 256 * 256 * 4));
  CK(hipMemcpy(df, hf.data(), 65536, hipMemcpyHostToDevice));
  CK(hipMemcpy(di, hi.data(), 65536, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int it_f = 60000, it_i = 30000;           // the same number of MFMAs: 48 per f16 step, 96 per i8 step
  const double mfma = 256.0 * 4 * it_f * 48;      // per launch, whole chip
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    kf16<<<256, 256>>>(df, o, it_f);
    CK(hipEventRecord(e0)); kf16<<<256, 256>>>(df, o, it_f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s f16 16x16x32: %.1f ms  %.2f ns per MFMA per SIMD (%.0f TFLOP/s issued)   ", mode ? "zeros " : "random", ms,
           ms * 1e6 / (it_f * 48.0), mfma * 16 * 16 * 32 * 2 / ms / 1e9);
    ki8<<<256, 256>>>(di, o, it_i);
    CK(hipEventRecord(e0)); ki8<<<256, 256>>>(di, o, it_i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("i8 16x16x64: %.1f ms  %.2f ns per MFMA per SIMD (%.0f TOP/s issued)\n", ms, ms * 1e6 / (it_i * 96.0),
           mfma * 16 * 16 * 64 * 2 / ms / 1e9);
  }
  return 0;
}
