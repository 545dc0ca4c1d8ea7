// What does gfx950's buffer range check look at?  (round 6: gemm_hl7.h moves the burst's row offset of its LDS-DMA into
// the SCALAR offset of a raw buffer load; whether the rows behind a tile's edge are dropped then depends on whether
// soffset takes part in the check.)  Each case: one wave, lane l loads a dword at base + voffset(l) + soffset (+ imm)
// through a raw descriptor (stride 0) with num_records = R bytes, once to a register and once straight into LDS;
// the buffer holds the dword's own byte offset + 1, so 0 means "dropped by the range check".
//   hipcc --offload-arch=gfx950 -O2 scratch/probe/buffer_oob_probe.hip -o scratch/probe/buffer_oob_probe && scratch/probe/buffer_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void probe(const unsigned* buf, unsigned records, int vstride, int soff, unsigned* out_reg, unsigned* out_lds) {
  __shared__ unsigned lds[64];
  const int lane = threadIdx.x;
  lds[lane] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, records, 0x00020000);
  const int voff = lane * vstride;
  out_reg[lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 4, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out_lds[lane] = lds[lane];
}

int main() {
  const size_t n = 1 << 20;                       // 4 MB of dwords, all of it mapped: a check that fails to fire reads real data
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 4 + 1);
  unsigned *d, *o1, *o2;
  if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&o1, 256) != hipSuccess || hipMalloc(&o2, 256) != hipSuccess) return 1;
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  struct Case { unsigned records; int vstride, soff; const char* what; };
  const Case cases[] = {
      {128, 4, 0, "voffset crosses num_records (lanes 32.. out)"},
      {128, 4, 64, "soffset 64 < num_records 128, voffset < 128 for lanes < 32: lanes 16..31 are in range only if soffset is NOT checked"},
      {128, 4, 4096, "soffset 4096 > num_records 128: everything out if soffset is checked; all of lanes < 32 in if not (or if the check wraps)"},
      {1024, 4, 1024, "soffset == num_records"},
      {1024, 4, 1020, "soffset + voffset: only lane 0 inside if checked"},
  };
  for (const Case& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c.records, c.vstride, c.soff, o1, o2);
    unsigned r[64], l[64];
    (void)hipMemcpy(r, o1, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(l, o2, 256, hipMemcpyDeviceToHost);
    int in_reg = 0, in_lds = 0, bad = 0, lds_untouched = 0;
    for (int i = 0; i < 64; ++i) {
      const unsigned want = (unsigned)(c.soff + i * c.vstride + 1);
      if (r[i] == want) ++in_reg; else if (r[i] != 0) ++bad;
      if (l[i] == want) ++in_lds; else if (l[i] == 0xdeadbeefu) ++lds_untouched; else if (l[i] != 0) ++bad;
    }
    printf("records %5u  voffset 4*lane  soffset %5d : %2d lanes fetched to registers, %2d to LDS (%2d LDS words left untouched, %d unexpected)  [%s]\n",
           c.records, c.soff, in_reg, in_lds, lds_untouched, bad, c.what);
  }
  return 0;
}
