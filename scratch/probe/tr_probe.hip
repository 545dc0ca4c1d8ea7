// Probe: what does ds_read_b64_tr_b16 deliver for the address pattern of the attention kernel's V reads?
// LDS image: V[key][d] (32 keys x 64 features, hl32 row = 256 B, hi plane only filled), value = key*64 + d,
// 16-byte slots XOR ((row&3)<<2 | (row>>2)&3).  Every lane reports the 4 halves of each of the 8 reads
// (mt, s, j2) of the hi plane: expected key 16s + 8(lane>>5) + 4 j2 + e, feature 32 mt + (lane & 31).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  __shared__ __attribute__((aligned(256))) uint8_t lds[16384];
  const int lane = threadIdx.x;
  for (int i = lane; i < 32 * 64; i += 64) {
    int key = i / 64, d = i % 64;
    int byte = (d < 32 ? d * 2 : 128 + (d - 32) * 2), ch = byte >> 4, w = byte & 15;
    int x = ((key & 3) << 2) | ((key >> 2) & 3);
    *reinterpret_cast<_Float16*>(lds + 8192 + key * 256 + ((ch ^ x) << 4) + w) = (_Float16)(float)(key * 64 + d);
  }
  __syncthreads();
  const int trq = (lane & 15) >> 2, trp = lane & 3, trG = lane >> 4, trh = trG >> 1;
  unsigned vaddr[2];
  for (int j2 = 0; j2 < 2; ++j2) {
    const int row = 8 * trh + 4 * j2 + trq, ch = 2 * (trG & 1) + (trp >> 1), x = (trq << 2) | ((2 * trh + j2) & 3);
    vaddr[j2] = (unsigned)(8192 + row * 256 + ((ch ^ x) << 4) + 8 * (trp & 1));
  }
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  int n = 0;
  for (int mt = 0; mt < 2; ++mt)
    for (int s = 0; s < 2; ++s)
      for (int j2 = 0; j2 < 2; ++j2) {
        unsigned a = (base + vaddr[j2]) ^ (unsigned)(mt * 128);
        a += s * 4096;
        u32x2_t r;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        out[(lane * 8 + n) * 2 + 0] = r[0];
        out[(lane * 8 + n) * 2 + 1] = r[1];
        ++n;
      }
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 2 * 4);
  k<<<1, 64>>>(d);
  unsigned h[64 * 8 * 2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int n = 0;
    for (int mt = 0; mt < 2; ++mt) for (int s = 0; s < 2; ++s) for (int j2 = 0; j2 < 2; ++j2, ++n) {
      for (int e = 0; e < 4; ++e) {
        unsigned w = h[(lane * 8 + n) * 2 + (e >> 1)];
        unsigned short bits = (e & 1) ? (w >> 16) : (w & 0xffff);
        _Float16 f; memcpy(&f, &bits, 2);
        int got = (int)(float)f, key = got / 64, dd = got % 64;
        int ekey = 16 * s + 8 * (lane >> 5) + 4 * j2 + e, ed = 32 * mt + (lane & 31);
        if (key != ekey || dd != ed) { if (bad < 24) printf("lane %2d mt %d s %d j2 %d e %d: got key %2d d %2d, expected key %2d d %2d\n", lane, mt, s, j2, e, key, dd, ekey, ed); ++bad; }
      }
    }
  }
  printf("mismatches: %d of %d\n", bad, 64 * 8 * 4);
  return 0;
}
