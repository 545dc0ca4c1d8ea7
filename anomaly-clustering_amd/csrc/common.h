// Shared host/device helpers for the gfx950 anomaly-clustering kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ac_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- error reporting (thread-local message, C ABI returns a code)
int ac_fail(int code, const char* fmt, ...);
int ac_check_launch(const char* what);

int ac_device_cus();                 // CU count of the calling thread's current device (cached per device)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device); done_mask is the
// instantiation's own static word (bit = device id).
int ac_set_max_lds(const void* kernel, int bytes, unsigned long long* done_mask);
int ac_gemm_policy();                // AC_GEMM_* (ac_set_gemm_policy)

#define AC_REQUIRE(cond, ...) do { if (!(cond)) return ac_fail(AC_EINVAL, __VA_ARGS__); } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace (256-byte aligned slices).
struct Arena {
  uint8_t* base; size_t size; size_t off;
  Arena(void* p, size_t n) : base(static_cast<uint8_t*>(p)), size(n), off(0) {}
  template <class T> T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (off + bytes > size) { off = size + 1; return nullptr; }
    T* r = reinterpret_cast<T*>(base + off); off += bytes; return r;
  }
  bool ok() const { return off <= size; }
};

// ---- hl32 addressing: element (row, k) of a [rows, K] matrix
//   hi at row*K*4 + (k/32)*128 + (k%32)*2 ; lo at +64.
__host__ __device__ inline size_t hl32_row_bytes(int K) { return (size_t)K * 4; }

// hi = fp16(x), lo = fp16(x - hi) for TWO values, packed as the MFMA operands want them (x0 in the low half): three
// vector instructions - one v_cvt_pk_f16_f32 and two v_fma_mix{lo,hi}_f16, which read the fp16 hi half in place
// (op_sel_hi), form hi * (-1) + x in fp32 and round ONCE to fp16.  x - float(hi) is exact in fp32 (hi is x rounded to 11
// bits), so the bits equal `(_Float16)(x - (float)hi)`: that form compiles to 2 x v_cvt_f32_f16 + v_pk_add_f32 +
// v_cvt_pk_f16_f32 per pair, 2.5 instructions per element against 1.5 here - and the split sits in every epilogue that
// writes hl32, in LayerNorm, in the pack kernel and, per score, in the attention loop, whose vector issue is as busy as
// its MFMA pipe (round 5; hipcc folds fma(h, -1, x) back into x - h, hence the two lines of assembly).
__device__ inline void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  const half2_t h = {(_Float16)x0, (_Float16)x1};
  hi = __builtin_bit_cast(uint32_t, h);
#ifdef AC_OLD_SPLIT                    // same-box A/B builds: the plain form (scratch/ab_vit.py)
  const half2_t l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
  lo = __builtin_bit_cast(uint32_t, l);
  return;
#endif
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
}

// Split 8 consecutive floats (k % 8 == 0) into the hi / lo fp16 vectors of hl32.
__device__ inline void split8(const float* v, half8& hi, half8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t a, b;
    split2(v[2 * j], v[2 * j + 1], a, b);
    h[j] = a;
    l[j] = b;
  }
  hi = __builtin_bit_cast(half8, h);
  lo = __builtin_bit_cast(half8, l);
}

// Store 8 consecutive columns [k, k+8) of one hl32 row.
__device__ inline void store_hl32_8(uint8_t* row_base, int k, const float* v) {
  half8 hi, lo;
  split8(v, hi, lo);
  uint8_t* p = row_base + (size_t)(k >> 5) * 128 + (size_t)(k & 31) * 2;
  *reinterpret_cast<half8*>(p) = hi;
  *reinterpret_cast<half8*>(p + 64) = lo;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
