// Host-side check of the tile enumeration of the all-pairs kernels (csrc/gemm_hl7.h: upper triangle on 256 x 192 tiles,
// with the group cache; csrc/gemm_hl2.h: 256 x 128 tiles): for every problem size given on the command line, every linear
// id of the launch maps to a tile inside the triangle, no tile twice, every tile that holds an element on or above the
// diagonal exactly once - in launch order (the XCD interleave of the block ids included) and in a scrambled order (the
// cache must not matter).  Built and run by tests/test_tile_order_cpu.py; prints "ok <rows> <tiles>" per size.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <utility>
#include <vector>
#include "gemm_hl7.h"

int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    const int M = atoi(argv[a]);
    for (int G : {1, 5, 6}) {
      const int Tm = (M + gemm7::BM - 1) / gemm7::BM, Tn = (M + gemm7::BN - 1) / gemm7::BN;
      const int total = gemm7::tri_rows_before(Tm, Tn);
      std::set<std::pair<int, int>> need, seen;
      for (int p = 0; p < Tm; ++p)
        for (int c = 0; c < Tn; ++c)
          if (gemm7::BN * c + gemm7::BN - 1 >= gemm7::BM * p) need.insert({p, c});
      if ((int)need.size() != total) { printf("FAIL count M=%d: %d tiles needed, launch has %d\n", M, (int)need.size(), total); return 1; }
      gemm7::TriCache tc;
      for (int id = 0; id < total; ++id) {                       // block ids in launch order
        int tm = -1, tn = -1;
        gemm7::tri_tile(gemm2::xcd_chunk_lin(id, total), Tm, Tn, G, tm, tn, tc);   // the kernels' own id -> position map
        if (!need.count({tm, tn}) || !seen.insert({tm, tn}).second) { printf("FAIL M=%d G=%d id=%d -> (%d, %d)\n", M, G, id, tm, tn); return 1; }
      }
      gemm7::TriCache t2;
      unsigned x = 12345u;
      for (int k = 0; k < 20000 && total > 0; ++k) {             // scrambled: every call may change the cached group
        x = x * 1664525u + 1013904223u;
        const int lin = (int)(x % (unsigned)total);
        int tm, tn, um, un;
        gemm7::TriCache fresh;
        gemm7::tri_tile(lin, Tm, Tn, G, tm, tn, t2);
        gemm7::tri_tile(lin, Tm, Tn, G, um, un, fresh);
        if (tm != um || tn != un) { printf("FAIL cache M=%d G=%d lin=%d\n", M, G, lin); return 1; }
      }
    }
    // the 256 x 128 kernel's triangle (gemm_hl2.h)
    {
      const int Tm = (M + 255) / 256, Tn = (M + 127) / 128, G = 6;
      const long long total = gemm2::tri_rows_before(Tm, Tn);
      std::set<std::pair<int, int>> seen;
      for (int id = 0; id < total; ++id) {
        int tm = -1, tn = -1;
        gemm2::tri_tile(gemm2::xcd_chunk_lin(id, total), Tm, Tn, G, tm, tn);
        if (tm < 0 || tm >= Tm || tn < 2 * tm || tn >= Tn || !seen.insert({tm, tn}).second) { printf("FAIL hl2 M=%d id=%d\n", M, id); return 1; }
      }
    }
    printf("ok %d %d\n", M, gemm7::tri_rows_before((M + 255) / 256, (M + 191) / 192));
  }
  return 0;
}
