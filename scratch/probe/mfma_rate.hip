// Hardware probe (not part of the product): sustained rate of the two fp16 MFMA shapes under the
// chip's power/clock management, in the exact operand pattern of gemm_hl2.h (64x64 output per wave,
// 3 MFMAs per fragment pair, fragments re-read from LDS every k-step, 2 waves per SIMD, random data).
//   hipcc --offload-arch=gfx950 -O2 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 32x32x16: per k32 step 16 fragment reads, 24 MFMAs
__global__ __launch_bounds__(512, 2) void k32(const _Float16* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[32768];     // 64 KB of operands
  for (int i = threadIdx.x; i < 32768 / 8; i += 512) ((half8*)lds)[i] = ((const half8*)src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const half8* base = (const half8*)lds + ((it * 37 + wave * 11) & 63) * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = base[(s * 8 + t * 4 + 0) * 64 % 4096 + lane];
        al[t] = base[(s * 8 + t * 4 + 1) * 64 % 4096 + lane];
        bh[t] = base[(s * 8 + t * 4 + 2) * 64 % 4096 + lane];
        bl[t] = base[(s * 8 + t * 4 + 3) * 64 % 4096 + lane];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) s += acc[a][b][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 16x16x32: per k32 step 16 fragment reads, 48 MFMAs
__global__ __launch_bounds__(512, 2) void k16(const _Float16* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
  for (int i = threadIdx.x; i < 32768 / 8; i += 512) ((half8*)lds)[i] = ((const half8*)src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const half8* base = (const half8*)lds + ((it * 37 + wave * 11) & 63) * 64;
    half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = base[(t * 4 + 0) * 64 % 4096 + lane];
      al[t] = base[(t * 4 + 1) * 64 % 4096 + lane];
      bh[t] = base[(t * 4 + 2) * 64 % 4096 + lane];
      bl[t] = base[(t * 4 + 3) * 64 % 4096 + lane];
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
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  std::vector<_Float16> h(32768);
  srand(1);
  for (auto& x : h) x = (_Float16)(((double)rand() / RAND_MAX - 0.5) * 0.25);
  _Float16* d; float* o;
  CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 256 * 512 * 4));
  CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40000;                     // k32 steps per wave
  const double flop = 256.0 * 8 * iters * 24 * 32 * 32 * 16 * 2;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    k32<<<256, 512>>>(d, o, iters);            // warm
    CK(hipEventRecord(e0)); k32<<<256, 512>>>(d, o, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("32x32x16: %.1f ms  %.0f TF/s issued   ", ms, flop / ms / 1e9);
    k16<<<256, 512>>>(d, o, iters);
    CK(hipEventRecord(e0)); k16<<<256, 512>>>(d, o, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("16x16x32: %.1f ms  %.0f TF/s issued\n", ms, flop / ms / 1e9);
  }
  return 0;
}
