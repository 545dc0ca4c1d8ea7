// Hardware probe (not part of the product path): verifies, on a real gfx950,
//  1. the A/B/C lane maps of v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16
//     with exact small-integer data (asymmetric operands),
//  2. the same for v_mfma_f32_32x32x2_f32,
//  3. the numerical quality of the 3-pass split-fp16 product
//        a.b ~= ah.bh + (ah.bl' + al'.bh) * 2^-11       (bl', al' pre-scaled by 2^11)
//     against an fp64 reference and against an fp32 fmaf chain, at K = 4096.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// C[32x32] = A[32xK] * B[32xK]^T, fp16 operands, one wave.
__global__ void k_f16_32(const _Float16* A, const _Float16* B, float* C, int K) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  float16v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 a = *(const half8*)(A + r * K + k0 + 8 * h);
    half8 b = *(const half8*)(B + r * K + k0 + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = acc[i];
  }
}

// C[16x16] = A[16xK] * B[16xK]^T
__global__ void k_f16_16(const _Float16* A, const _Float16* B, float* C, int K) {
  int l = threadIdx.x, r = l & 15, q = l >> 4;
  float4v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    half8 a = *(const half8*)(A + r * K + k0 + 8 * q);
    half8 b = *(const half8*)(B + r * K + k0 + 8 * q);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = acc[i];
}

// fp32-input MFMA: C[32x32] = A[32xK] * B[32xK]^T
__global__ void k_f32_32(const float* A, const float* B, float* C, int K) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  float16v acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[r * K + k0 + h], acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = acc[i];
  }
}

// split-fp16, 3 MFMA passes, two accumulators (main + cross scaled by 2^-11).
// mode 0: separate accumulators; mode 1: single accumulator with unscaled lo parts.
__global__ void k_split(const _Float16* Ah, const _Float16* Al, const _Float16* Bh, const _Float16* Bl,
                        float* C, int K, int mode, float lo_scale_inv) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  float16v acc = {0}, acc2 = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 ah = *(const half8*)(Ah + r * K + k0 + 8 * h);
    half8 al = *(const half8*)(Al + r * K + k0 + 8 * h);
    half8 bh = *(const half8*)(Bh + r * K + k0 + 8 * h);
    half8 bl = *(const half8*)(Bl + r * K + k0 + 8 * h);
    if (mode == 0) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = (mode == 0) ? acc[i] + acc2[i] * lo_scale_inv : acc[i];
  }
}

template <class T> T* dcopy(const std::vector<T>& v) {
  T* d; CK(hipMalloc(&d, v.size() * sizeof(T)));
  CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static double frand() { return (double)rand() / RAND_MAX; }
static double grand() {  // gaussian
  double u = frand() + 1e-12, v = frand();
  return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  int fails = 0;
  // ---- 1. layout, f16 32x32x16, K=32 with integer data
  {
    int K = 32;
    std::vector<_Float16> A(32 * K), B(32 * K);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) {
      A[i * K + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
      B[i * K + k] = (_Float16)((i * 5 + k * 2 + 1) % 9 - 4);
    }
    auto dA = dcopy(A), dB = dcopy(B);
    float* dC; CK(hipMalloc(&dC, 32 * 32 * 4));
    k_f16_32<<<1, 64>>>(dA, dB, dC, K);
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[j * K + k];
      if (s != C[i * 32 + j]) ++bad;
    }
    printf("layout f16 32x32x16: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  {
    int K = 64;
    std::vector<_Float16> A(16 * K), B(16 * K);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) {
      A[i * K + k] = (_Float16)((i * 3 + k * 5) % 7 - 3);
      B[i * K + k] = (_Float16)((i * 5 + k * 2 + 1) % 9 - 4);
    }
    auto dA = dcopy(A), dB = dcopy(B);
    float* dC; CK(hipMalloc(&dC, 16 * 16 * 4));
    k_f16_16<<<1, 64>>>(dA, dB, dC, K);
    std::vector<float> C(16 * 16);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[j * K + k];
      if (s != C[i * 16 + j]) ++bad;
    }
    printf("layout f16 16x16x32: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  {
    int K = 8;
    std::vector<float> A(32 * K), B(32 * K);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) {
      A[i * K + k] = (float)((i * 3 + k * 5) % 7 - 3);
      B[i * K + k] = (float)((i * 5 + k * 2 + 1) % 9 - 4);
    }
    auto dA = dcopy(A), dB = dcopy(B);
    float* dC; CK(hipMalloc(&dC, 32 * 32 * 4));
    k_f32_32<<<1, 64>>>(dA, dB, dC, K);
    std::vector<float> C(32 * 32);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[j * K + k];
      if (s != C[i * 32 + j]) ++bad;
    }
    printf("layout f32 32x32x2 : %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad); fails += bad != 0;
  }
  // ---- 3. split-fp16 accuracy at K=4096, gaussian data std 0.6 (Z-like) and weights-like
  for (int scen = 0; scen < 3; ++scen) {
    int K = 4096;
    double sa = scen == 0 ? 0.6 : (scen == 1 ? 1.0 : 1.0), sb = scen == 0 ? 0.6 : (scen == 1 ? 0.02 : 0.02);
    float bscale = scen == 2 ? 32.f : 1.f;  // scen 2: weights pre-scaled by 2^5 (exact), unscaled after
    std::vector<float> A(32 * K), B(32 * K);
    for (auto& x : A) x = (float)(grand() * sa);
    for (auto& x : B) x = (float)(grand() * sb);
    const float LS = 2048.f;
    std::vector<_Float16> Ah(32 * K), Al(32 * K), Bh(32 * K), Bl(32 * K), Al0(32 * K), Bl0(32 * K);
    for (int i = 0; i < 32 * K; ++i) {
      Ah[i] = (_Float16)A[i]; Al[i] = (_Float16)((A[i] - (float)Ah[i]) * LS); Al0[i] = (_Float16)(A[i] - (float)Ah[i]);
      float b = B[i] * bscale;
      Bh[i] = (_Float16)b; Bl[i] = (_Float16)((b - (float)Bh[i]) * LS); Bl0[i] = (_Float16)(b - (float)Bh[i]);
    }
    auto dAh = dcopy(Ah), dAl = dcopy(Al), dBh = dcopy(Bh), dBl = dcopy(Bl), dAl0 = dcopy(Al0), dBl0 = dcopy(Bl0);
    auto dA = dcopy(A), dB = dcopy(B);
    float* dC; CK(hipMalloc(&dC, 32 * 32 * 4));
    std::vector<float> C0(1024), C1(1024), C2(1024), C3(1024);
    k_split<<<1, 64>>>(dAh, dAl, dBh, dBl, dC, K, 0, 1.f / LS);
    CK(hipMemcpy(C0.data(), dC, 4096, hipMemcpyDeviceToHost));
    k_split<<<1, 64>>>(dAh, dAl0, dBh, dBl0, dC, K, 1, 1.f);
    CK(hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost));
    k_f32_32<<<1, 64>>>(dA, dB, dC, K);
    CK(hipMemcpy(C2.data(), dC, 4096, hipMemcpyDeviceToHost));
    k_f16_32<<<1, 64>>>(dAh, dBh, dC, K);
    CK(hipMemcpy(C3.data(), dC, 4096, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0, ecpu = 0, mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; float f = 0; double sabs = 0;
      for (int k = 0; k < K; ++k) {
        s += (double)A[i * K + k] * (double)B[j * K + k];
        sabs += fabs((double)A[i * K + k] * (double)B[j * K + k]);
        f = fmaf(A[i * K + k], B[j * K + k], f);
      }
      e0 = fmax(e0, fabs(C0[i * 32 + j] / bscale - s));
      e1 = fmax(e1, fabs(C1[i * 32 + j] / bscale - s));
      e2 = fmax(e2, fabs(C2[i * 32 + j] - s));
      e3 = fmax(e3, fabs(C3[i * 32 + j] / bscale - s));
      ecpu = fmax(ecpu, fabs(f - s));
      mag = fmax(mag, sabs);
    }
    printf("scen %d (sa=%.2f sb=%.2f bscale=%g) K=%d  sum|ab|max=%.3g\n", scen, sa, sb, bscale, K, mag);
    printf("   max abs err vs fp64: split2acc=%.3e split1acc=%.3e f32mfma=%.3e f16_1pass=%.3e cpu_fmaf=%.3e\n",
           e0, e1, e2, e3, ecpu);
  }
  printf(fails ? "PROBE FAIL\n" : "PROBE OK\n");
  return fails;
}
