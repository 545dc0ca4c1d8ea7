// Patch aggregation on gfx950: tapped feature maps -> per-patch embedding matrix Z.
//
// Reference arithmetic (paths relative to Anomaly-Clustering/):
//   models/patchcore/patchcore.py:375-385  CLS drop, [B,T,C] -> [B,C,H,W], fresh LayerNorm([C,H,W]) (eps 1e-5)
//   models/patchcore/patchcore.py:439-465  Unfold(k, stride 1, pad (k-1)/2): patch vector index e = c*k*k + ky*k + kx
//   models/patchcore/patchcore.py:398-421  bilinear (align_corners=False) resize of the patch GRID to layer 0's grid
//   models/patchcore/common.py:163-170     adaptive_avg_pool1d(C*k*k -> pre_dim) per layer
//   models/patchcore/common.py:173-183     adaptive_avg_pool1d(n_layers*pre_dim -> tgt_dim)
//
// The reference materialises the unfolded tensor (9x the feature map) in HBM and copies it
// several times; here one workgroup owns one output row of Z, builds that patch's unfolded,
// standardised, grid-aligned vector in LDS (e-order, so every pooling window is a contiguous
// LDS range) and writes the 16 KB row of Z once.  HBM-bound on the Z write; the feature maps
// are re-read from L2 (each token is needed by k*k neighbouring patches).
#include "common.h"

// ---- pass 1: per (layer, image) sum and sum of squares over C*H*W values, fp64.  Deterministic: every
// workgroup writes its partial pair to its own slot, agg_stats_finish_kernel adds the slots in a fixed order
// (mean and rstd scale every value of Z, so a run-to-run last-bit difference here would show everywhere).
constexpr int kStatChunks = 64;
__global__ __launch_bounds__(256) void agg_stats_kernel(const float* __restrict__ f, int kind, int C, int HW,
                                                        double* __restrict__ partial /* [B][kStatChunks][2] */) {
  const int b = blockIdx.y;
  // both kinds: the image's values are one contiguous range (tokens: after the CLS row)
  const size_t n = (size_t)C * HW;
  const float* base = f + (kind == AC_FEAT_TOKENS ? ((size_t)b * (HW + 1) + 1) * C : (size_t)b * n);
  double s = 0.0, ss = 0.0;
  if ((n & 3) == 0 && ((uintptr_t)base & 15) == 0) {           // 16-byte loads (every ViT / CNN shape on the path)
    const f32x4* b4 = reinterpret_cast<const f32x4*>(base);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
      const f32x4 q = b4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const double v = q[j]; s += v; ss += v * v; }
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      double v = base[i];
      s += v;
      ss += v * v;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  __shared__ double red[2][4];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = s; red[1][wv] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((size_t)b * kStatChunks + blockIdx.x) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// one wave per image: lane c holds chunk c's partial, fixed-order butterfly
// ... and turns them into the image's mean and 1/std ONCE (fp64, rounded to fp32 as LayerNorm's fp32 arithmetic
// sees them) - the pool kernels used to redo this fp64 division + square root in every thread of every workgroup.
__global__ __launch_bounds__(64) void agg_stats_finish_kernel(const double* __restrict__ partial, int chunks, double n,
                                                              float eps, float* __restrict__ norm /* [B][2] */) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const double* p = partial + ((size_t)b * kStatChunks + lane) * 2;
  double s = lane < chunks ? p[0] : 0.0, ss = lane < chunks ? p[1] : 0.0;
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane == 0) {
    const double mean_d = s / n;
    const double var_d = fmax(ss / n - mean_d * mean_d, 0.0);
    norm[b * 2] = (float)mean_d;
    norm[b * 2 + 1] = (float)(1.0 / sqrt(var_d + (double)eps));
  }
}

// Workgroups are dealt round-robin to the 8 XCDs, and neighbouring patches share 6 of their 9 tokens:
// give every XCD a contiguous run of patches so those re-reads hit its own L2 (FETCH_SIZE showed the
// plain mapping fetching every token ~6 times from beyond L2).  Bijective for any grid size.
__device__ inline int xcd_local_patch(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

struct AggLayer {
  const float* f;
  const float* norm;     // [B][2]: mean, 1/sqrt(var + eps) of the image's C*H*W values
  int kind, C, H, W;
};
struct AggParams {
  AggLayer layer[4];
  int n_layers, k, pre, tgt, H0, W0;
  float eps;
  float* Z;
  uint8_t* Zhl;           // register path only (ac_patch_aggregate_hl32): the rows leave as hl32 instead of fp32
};

// torch's area_pixel_compute_source_index for bilinear, align_corners=False
__device__ inline void bilinear_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l0, float& l1) {
  if (in_size == out_size) { i0 = i1 = dst; l0 = 1.f; l1 = 0.f; return; }
  float scale = (float)in_size / (float)out_size;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// ---- pass 2: one workgroup per (reference-grid patch, image)
__global__ __launch_bounds__(256) void agg_pool_kernel(AggParams prm) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int patch = xcd_local_patch(blockIdx.x, gridDim.x), b = blockIdx.y;
  const int y = patch / prm.W0, x = patch - y * prm.W0;
  const int k = prm.k, k2 = k * k, pad = (k - 1) / 2;
  int maxL = 0;
  for (int l = 0; l < prm.n_layers; ++l) maxL = max(maxL, prm.layer[l].C * k2);
  float* U = lds;                 // [maxL] unfolded, standardised, aligned patch vector of the current layer
  float* V = lds + maxL;          // [n_layers*pre] pooled layers (only when the final pool is not the identity)
  const bool identity = (prm.n_layers * prm.pre == prm.tgt);
  float* zrow = prm.Z + ((size_t)b * prm.H0 * prm.W0 + patch) * prm.tgt;

  for (int l = 0; l < prm.n_layers; ++l) {
    const AggLayer L = prm.layer[l];
    const int HW = L.H * L.W, Lk = L.C * k2;
    const float mean = L.norm[b * 2], rstd = L.norm[b * 2 + 1];
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    bilinear_src(y, L.H, prm.H0, y0, y1, ly0, ly1);
    bilinear_src(x, L.W, prm.W0, x0, x1, lx0, lx1);
    const bool same_grid = (L.H == prm.H0 && L.W == prm.W0);

    // standardised value of channel c at own-grid position (yy, xx); 0 outside (Unfold's zero padding
    // is applied AFTER the LayerNorm)
    auto val = [&](int c, int yy, int xx) -> float {
      if (yy < 0 || yy >= L.H || xx < 0 || xx >= L.W) return 0.f;
      size_t idx = (L.kind == AC_FEAT_TOKENS) ? ((size_t)b * (HW + 1) + 1 + yy * L.W + xx) * L.C + c
                   : (L.kind == AC_FEAT_NHWC) ? ((size_t)b * HW + yy * L.W + xx) * L.C + c
                                              : (((size_t)b * L.C + c) * L.H + yy) * L.W + xx;
      return (L.f[idx] - mean) * rstd;
    };
    if (L.kind != AC_FEAT_NCHW) {
      // channel on the thread: global reads contiguous in c; LDS writes at stride k2 floats
      for (int kk = 0; kk < k2; ++kk) {
        const int dy = kk / k - pad, dx = kk - (kk / k) * k - pad;
        for (int c = tid; c < L.C; c += 256) {
          float v;
          if (same_grid) v = val(c, y + dy, x + dx);
          else
            v = ly0 * (lx0 * val(c, y0 + dy, x0 + dx) + lx1 * val(c, y0 + dy, x1 + dx)) +
                ly1 * (lx0 * val(c, y1 + dy, x0 + dx) + lx1 * val(c, y1 + dy, x1 + dx));
          U[c * k2 + kk] = v;
        }
      }
    } else {
      for (int e = tid; e < Lk; e += 256) {
        const int c = e / k2, kk = e - c * k2;
        const int dy = kk / k - pad, dx = kk - (kk / k) * k - pad;
        float v;
        if (same_grid) v = val(c, y + dy, x + dx);
        else
          v = ly0 * (lx0 * val(c, y0 + dy, x0 + dx) + lx1 * val(c, y0 + dy, x1 + dx)) +
              ly1 * (lx0 * val(c, y1 + dy, x0 + dx) + lx1 * val(c, y1 + dy, x1 + dx));
        U[e] = v;
      }
    }
    __syncthreads();
    // adaptive_avg_pool1d(Lk -> pre): window [floor(o*Lk/pre), ceil((o+1)*Lk/pre)), sequential fp32 sum / length
    for (int o = tid; o < prm.pre; o += 256) {
      const int s = (int)(((unsigned)o * (unsigned)Lk) / (unsigned)prm.pre);      // 32-bit: pre*Lk < 2^31 (host check)
      const int e = (int)(((unsigned)(o + 1) * (unsigned)Lk + (unsigned)prm.pre - 1u) / (unsigned)prm.pre);
      float acc = 0.f;
      for (int i = s; i < e; ++i) acc += U[i];
      const float r = acc / (float)(e - s);
      if (identity) zrow[l * prm.pre + o] = r;
      else V[l * prm.pre + o] = r;
    }
    __syncthreads();
  }
  if (!identity) {
    const int Lv = prm.n_layers * prm.pre;
    for (int t = tid; t < prm.tgt; t += 256) {
      const int s = (int)(((unsigned)t * (unsigned)Lv) / (unsigned)prm.tgt);
      const int e = (int)(((unsigned)(t + 1) * (unsigned)Lv + (unsigned)prm.tgt - 1u) / (unsigned)prm.tgt);
      float acc = 0.f;
      for (int i = s; i < e; ++i) acc += V[i];
      zrow[t] = acc / (float)(e - s);
    }
  }
}

// ---- pass 2, fast path: every layer is a token map on the reference grid (the ViT case), C % 4 == 0,
// patch size K in {1, 3}.  The generic kernel above issues one dependent 4-byte global load per LDS
// element (a run-time trip count keeps hipcc from batching them): ~27 serial L2 round trips per layer,
// which made the whole kernel latency-bound at 8 % of the HBM roofline.  Here each thread fetches 4
// channels of all K*K neighbour tokens with K*K independent 16-byte loads in flight, and the LDS image is
// token-major ([kk][C + 4], 16-byte stores); the pooling loop walks (c, kk) incrementally.
template <int K>
__global__ __launch_bounds__(256) void agg_pool_tokens_kernel(AggParams prm) {
  extern __shared__ float lds[];
  constexpr int K2 = K * K, PAD = (K - 1) / 2;
  const int tid = threadIdx.x;
  const int patch = xcd_local_patch(blockIdx.x, gridDim.x), b = blockIdx.y;
  const int y = patch / prm.W0, x = patch - y * prm.W0;
  const int C0 = prm.layer[0].C;
  int maxC = 0;
  for (int l = 0; l < prm.n_layers; ++l) maxC = max(maxC, prm.layer[l].C);
  const int Cp = maxC + 4;
  float* U = lds;                       // [K2][Cp]
  float* V = lds + K2 * Cp;             // [n_layers*pre] when the final pool is not the identity
  const bool identity = (prm.n_layers * prm.pre == prm.tgt);
  float* zrow = prm.Z + ((size_t)b * prm.H0 * prm.W0 + patch) * prm.tgt;
  (void)C0;
  for (int l = 0; l < prm.n_layers; ++l) {
    const AggLayer L = prm.layer[l];
    const int HW = L.H * L.W, Lk = L.C * K2;
    const float mean = L.norm[b * 2], rstd = L.norm[b * 2 + 1];
    const float* img = L.f + ((size_t)b * (HW + 1) + 1) * L.C;
    for (int c4 = tid * 4; c4 < L.C; c4 += 1024) {
      f32x4 ld[K2];
      bool inb[K2];
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) {
        const int yy = y + kk / K - PAD, xx = x + kk % K - PAD;
        inb[kk] = (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W);
        const int yc = min(max(yy, 0), L.H - 1), xc = min(max(xx, 0), L.W - 1);
        ld[kk] = *reinterpret_cast<const f32x4*>(img + (size_t)(yc * L.W + xc) * L.C + c4);   // unconditional
      }
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = inb[kk] ? (ld[kk][j] - mean) * rstd : 0.f;
        *reinterpret_cast<f32x4*>(U + kk * Cp + c4) = v;
      }
    }
    __syncthreads();
    for (int o = tid; o < prm.pre; o += 256) {
      const int s = (int)(((unsigned)o * (unsigned)Lk) / (unsigned)prm.pre);
      const int e = (int)(((unsigned)(o + 1) * (unsigned)Lk + (unsigned)prm.pre - 1u) / (unsigned)prm.pre);
      int c = s / K2, kk = s - c * K2;
      float acc = 0.f;
      for (int i = s; i < e; ++i) {
        acc += U[kk * Cp + c];
        if (++kk == K2) { kk = 0; ++c; }
      }
      const float r = acc / (float)(e - s);
      if (identity) zrow[l * prm.pre + o] = r;
      else V[l * prm.pre + o] = r;
    }
    __syncthreads();
  }
  if (!identity) {
    const int Lv = prm.n_layers * prm.pre;
    for (int t = tid; t < prm.tgt; t += 256) {
      const int s = (int)(((unsigned)t * (unsigned)Lv) / (unsigned)prm.tgt);
      const int e = (int)(((unsigned)(t + 1) * (unsigned)Lv + (unsigned)prm.tgt - 1u) / (unsigned)prm.tgt);
      float acc = 0.f;
      for (int i = s; i < e; ++i) acc += V[i];
      zrow[t] = acc / (float)(e - s);
    }
  }
}

// ---- pass 2, register path for the DINO ViT shape: token maps on one grid, K = 3, 3 * pre == 8 * C per layer
// (C = 768 -> 6912 unfolded values pooled to 2048: every 8 outputs read exactly 27 = 3 channels x 9 neighbours
// inputs), identity final pool.  Thread t owns channels 3t..3t+2: its 27 inputs are nine 12-byte loads (one per
// neighbour token, contiguous across the wave), all in flight at once; its 8 outputs are window sums over those registers, in the same order as the kernels
// above (bit-identical results), written as two 16-byte stores.  One (patch, layer) per workgroup.  No LDS, no
// barrier, no bank conflicts: what is left is the 16 KB row write per patch (HBM) and the neighbour re-reads
// from L2.
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };

// HL: the 8 outputs of a thread leave as hl32 (16 bytes of hi halves + 16 of lo halves into the row's 128-byte group: the split
// pack_hl32_kernel would make of the same fp32 values, so the same bytes) - round 6: the shards a rank RECEIVES over the ring
// are only ever multiplied, their fp32 Z (12.85 MB per image written, then read back by the pack kernel) need not exist.
template <bool HL>
__global__ __launch_bounds__(256) void agg_pool_tokens_27_8_kernel(AggParams prm) {
  const int t = threadIdx.x;
  const int patch = xcd_local_patch(blockIdx.x, gridDim.x), b = blockIdx.y, l = blockIdx.z;   // one layer per workgroup:
  const int y = patch / prm.W0, x = patch - y * prm.W0;                                       // 48 registers, 8 waves per SIMD
  if (t * 8 >= prm.pre) return;
  const AggLayer L = prm.layer[l];
  const int HW = L.H * L.W;
  const float* img = L.f + ((size_t)b * (HW + 1) + 1) * L.C + 3 * t;
  float v[27];
#pragma unroll
  for (int kk = 0; kk < 9; ++kk) {
    const int yy = y + kk / 3 - 1, xx = x + kk % 3 - 1;
    const int yc = min(max(yy, 0), L.H - 1), xc = min(max(xx, 0), L.W - 1);
    const F3 f = *reinterpret_cast<const F3*>(img + (size_t)(yc * L.W + xc) * L.C);      // unconditional
    v[kk] = f.a; v[9 + kk] = f.b; v[18 + kk] = f.c;                                        // e-order: c * 9 + kk
  }
  const float mean = L.norm[b * 2], rstd = L.norm[b * 2 + 1];
#pragma unroll
  for (int kk = 0; kk < 9; ++kk) {
    const int yy = y + kk / 3 - 1, xx = x + kk % 3 - 1;
    const bool inb = (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W);          // Unfold pads with zeros AFTER the LayerNorm
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c * 9 + kk] = inb ? (v[c * 9 + kk] - mean) * rstd : 0.f;
  }
  float out[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int s = (q * 27) / 8, e = ((q + 1) * 27 + 7) / 8;                 // adaptive_avg_pool1d window, local
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 27; ++i)
      if (i >= s && i < e) acc += v[i];
    out[q] = (e - s) == 4 ? acc * 0.25f : acc / (float)(e - s);            // x / 4 == x * 0.25 exactly
  }
  const size_t row = (size_t)b * prm.H0 * prm.W0 + patch;
  if constexpr (HL) {
    store_hl32_8(prm.Zhl + row * prm.tgt * 4, l * prm.pre + 8 * t, out);
  } else {
    float* o = prm.Z + row * prm.tgt + l * prm.pre + 8 * t;
    *reinterpret_cast<f32x4*>(o) = f32x4{out[0], out[1], out[2], out[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{out[4], out[5], out[6], out[7]};
  }
}

// ---- the same, one workgroup per ROW of patches (round 4).  Adjacent patches share six of their nine neighbour tokens:
// the workgroup slides a 3 x 3 window of standardised tokens along the row, loading the three tokens of the new column per
// step (3 x 12 bytes per thread) instead of nine - a third of the L2 reads and 1 / W0 of the workgroups (one per (patch,
// layer) was 130 000 launches of ~20 instructions of useful work at 83 images).  Same window sums in the same order:
// bit-identical.  Columns are kept in four named register sets rotated by a 4-step unrolled loop (no copies); the loads
// of column x + 2 fly while patch x is summed.  Small batches (a rank's 11 images) keep one workgroup per patch: a row
// per workgroup would leave the chip a third full.
// Round 6: the workgroups of ONE (image, layer) - its H0 patch rows, each of which reads the token rows y - 1, y, y + 1 - run on ONE XCD
// (workgroups go to the XCDs round robin by their linear id: id & 7 = XCD, id >> 3 = its slot there; XCD x works the units x, x + 8, ...
// row by row), so a token row is fetched over the fabric once and met in that XCD's L2 by its two neighbours.  The plain (y, b, l) grid
// dealt the rows of an image to all eight XCDs: every token row crossed the fabric three times (1171 MB fetched for 400 MB of taps at 83
// images, profiles/r05_bench_pmc.json).  Same arithmetic per workgroup: bit-identical.
template <bool HL>
__global__ __launch_bounds__(256) void agg_pool_tokens_27_8_row_kernel(AggParams prm, int B) {
  const int t = threadIdx.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / prm.H0) * 8 + xcd, y = slot - (slot / prm.H0) * prm.H0;       // unit = image + B * layer
  if (unit >= B * prm.n_layers) return;                                                  // (the grid is padded to whole rounds of 8 units)
  const int b = unit % B, l = unit / B;
  if (t * 8 >= prm.pre) return;
  const AggLayer L = prm.layer[l];
  const int HW = L.H * L.W;
  const float* img = L.f + ((size_t)b * (HW + 1) + 1) * L.C + 3 * t;
  const float mean = L.norm[b * 2], rstd = L.norm[b * 2 + 1];
  // one column of the window: the standardised tokens (y-1, xx), (y, xx), (y+1, xx) - zeros outside the map (Unfold pads
  // with zeros AFTER the LayerNorm) - as [row][channel]
  auto load_col = [&](int xx, float (&c)[3][3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = y + r - 1;
      const bool inb = (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W);
      const int yc = min(max(yy, 0), L.H - 1), xc = min(max(xx, 0), L.W - 1);
      const F3 f = *reinterpret_cast<const F3*>(img + (size_t)(yc * L.W + xc) * L.C);      // unconditional
      c[r][0] = inb ? (f.a - mean) * rstd : 0.f;
      c[r][1] = inb ? (f.b - mean) * rstd : 0.f;
      c[r][2] = inb ? (f.c - mean) * rstd : 0.f;
    }
  };
  // patch x from the columns x-1 (c0), x (c1), x+1 (c2): v[c * 9 + kk], kk = row * 3 + column (the e-order of the unfold)
  auto emit = [&](int x, const float (&c0)[3][3], const float (&c1)[3][3], const float (&c2)[3][3]) {
    float v[27];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        v[ch * 9 + r * 3 + 0] = c0[r][ch];
        v[ch * 9 + r * 3 + 1] = c1[r][ch];
        v[ch * 9 + r * 3 + 2] = c2[r][ch];
      }
    float out[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int s = (q * 27) / 8, e = ((q + 1) * 27 + 7) / 8;                 // adaptive_avg_pool1d window, local
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 27; ++i)
        if (i >= s && i < e) acc += v[i];
      out[q] = (e - s) == 4 ? acc * 0.25f : acc / (float)(e - s);            // x / 4 == x * 0.25 exactly
    }
    const size_t row = (size_t)b * prm.H0 * prm.W0 + (size_t)y * prm.W0 + x;
    if constexpr (HL) {
      store_hl32_8(prm.Zhl + row * prm.tgt * 4, l * prm.pre + 8 * t, out);
    } else {
      float* o = prm.Z + row * prm.tgt + l * prm.pre + 8 * t;
      *reinterpret_cast<f32x4*>(o) = f32x4{out[0], out[1], out[2], out[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{out[4], out[5], out[6], out[7]};
    }
  };
  // four column sets: the loads of column x + 2 are in flight while patch x is summed and stored
  float ca[3][3], cb[3][3], cc[3][3], cd[3][3];
  load_col(-1, ca);
  load_col(0, cb);
  load_col(1, cc);
  int x = 0;
  for (; x + 3 < prm.W0; x += 4) {                     // four patches per trip: the column sets rotate by name
    load_col(x + 2, cd); emit(x, ca, cb, cc);
    load_col(x + 3, ca); emit(x + 1, cb, cc, cd);
    load_col(x + 4, cb); emit(x + 2, cc, cd, ca);
    load_col(x + 5, cc); emit(x + 3, cd, ca, cb);
  }
  if (x < prm.W0) { load_col(x + 2, cd); emit(x, ca, cb, cc); ++x; }
  if (x < prm.W0) { load_col(x + 2, ca); emit(x, cb, cc, cd); ++x; }
  if (x < prm.W0) { emit(x, cc, cd, ca); }
}

extern "C" size_t ac_patch_aggregate_workspace_bytes(int n_layers, const int* C, const int* H, const int* W, int B,
                                                     int patchsize, int pre_dim, int tgt_dim) {
  (void)C; (void)H; (void)W; (void)patchsize; (void)pre_dim; (void)tgt_dim;
  if (n_layers <= 0 || B <= 0) return 256;
  return align_up((size_t)n_layers * B * 2 * sizeof(double), 256) +
         align_up((size_t)n_layers * B * kStatChunks * 2 * sizeof(double), 256) + 256;
}

// The register path's shape test (both entry points): every layer a token map on the reference grid, K = 3,
// 3 * pre == 8 * C per layer, identity final pool.
static bool agg_is_reg27(const int* kind, const int* C, const int* H, const int* W, int n_layers, int patchsize, int pre_dim,
                         int tgt_dim) {
  bool ok = n_layers >= 1 && n_layers <= 4 && patchsize == 3 && (long long)n_layers * pre_dim == tgt_dim && pre_dim % 8 == 0 &&
            pre_dim / 8 <= 256 && tgt_dim % 4 == 0;
  for (int l = 0; l < n_layers && ok; ++l)
    ok = kind[l] == AC_FEAT_TOKENS && H[l] == H[0] && W[l] == W[0] && C[l] % 4 == 0 && (3LL * pre_dim == 8LL * C[l]) && C[l] % 3 == 0;
  return ok;
}

extern "C" int ac_patch_aggregate_hl32_supported(const int* kind, const int* C, const int* H, const int* W, int n_layers,
                                                 int patchsize, int pre_dim, int tgt_dim) {
  if (!kind || !C || !H || !W) return 0;
  return agg_is_reg27(kind, C, H, W, n_layers, patchsize, pre_dim, tgt_dim) && tgt_dim % 32 == 0 ? 1 : 0;
}

static int patch_aggregate_impl(const float* const* feats, const int* kind, const int* C, const int* H, const int* W,
                                int n_layers, int B, int patchsize, int pre_dim, int tgt_dim, float eps, float* Z,
                                uint8_t* Zhl, void* ws, size_t ws_bytes, hipStream_t stream);

extern "C" int ac_patch_aggregate(const float* const* feats, const int* kind, const int* C, const int* H, const int* W,
                                  int n_layers, int B, int patchsize, int pre_dim, int tgt_dim, float eps, float* Z,
                                  void* ws, size_t ws_bytes, void* stream_) {
  AC_REQUIRE(Z, "ac_patch_aggregate: null pointer");
  return patch_aggregate_impl(feats, kind, C, H, W, n_layers, B, patchsize, pre_dim, tgt_dim, eps, Z, nullptr, ws, ws_bytes,
                              as_stream(stream_));
}

extern "C" int ac_patch_aggregate_hl32(const float* const* feats, const int* kind, const int* C, const int* H, const int* W,
                                       int n_layers, int B, int patchsize, int pre_dim, int tgt_dim, float eps, void* Z_hl32,
                                       void* ws, size_t ws_bytes, void* stream_) {
  AC_REQUIRE(Z_hl32 && ((uintptr_t)Z_hl32 & 15) == 0, "ac_patch_aggregate_hl32: Z_hl32 must be a 16-byte aligned pointer");
  AC_REQUIRE(ac_patch_aggregate_hl32_supported(kind, C, H, W, n_layers, patchsize, pre_dim, tgt_dim),
             "ac_patch_aggregate_hl32: only the token-map shape of the register path (K = 3, 3 * pre_dim == 8 * C, identity final "
             "pool, tgt_dim %% 32 == 0) writes hl32 directly - ask ac_patch_aggregate_hl32_supported first");
  return patch_aggregate_impl(feats, kind, C, H, W, n_layers, B, patchsize, pre_dim, tgt_dim, eps, nullptr,
                              static_cast<uint8_t*>(Z_hl32), ws, ws_bytes, as_stream(stream_));
}

static int patch_aggregate_impl(const float* const* feats, const int* kind, const int* C, const int* H, const int* W,
                                int n_layers, int B, int patchsize, int pre_dim, int tgt_dim, float eps, float* Z,
                                uint8_t* Zhl, void* ws, size_t ws_bytes, hipStream_t stream) {
  AC_REQUIRE(feats && kind && C && H && W && (Z || Zhl) && ws, "ac_patch_aggregate: null pointer");
  AC_REQUIRE(n_layers >= 1 && n_layers <= 4, "ac_patch_aggregate: n_layers=%d must be 1..4", n_layers);
  AC_REQUIRE(patchsize >= 1 && (patchsize & 1), "ac_patch_aggregate: patchsize=%d must be odd (stride-1 unfold keeps the grid)",
             patchsize);
  AC_REQUIRE(pre_dim > 0 && tgt_dim > 0, "ac_patch_aggregate: bad dims");
  AC_REQUIRE(B > 0 && B <= 65535, "ac_patch_aggregate: B=%d out of range", B);
  if (ws_bytes < ac_patch_aggregate_workspace_bytes(n_layers, C, H, W, B, patchsize, pre_dim, tgt_dim))
    return ac_fail(AC_EWORKSPACE, "ac_patch_aggregate: workspace too small");
  AggParams prm;
  prm.n_layers = n_layers; prm.k = patchsize; prm.pre = pre_dim; prm.tgt = tgt_dim;
  prm.H0 = H[0]; prm.W0 = W[0]; prm.eps = eps; prm.Z = Z; prm.Zhl = Zhl;
  float* norm = static_cast<float*>(ws);
  double* partial = reinterpret_cast<double*>(static_cast<uint8_t*>(ws) + align_up((size_t)n_layers * B * 2 * sizeof(double), 256));
  int maxL = 0;
  for (int l = 0; l < n_layers; ++l) {
    AC_REQUIRE(feats[l] && C[l] > 0 && H[l] > 0 && W[l] > 0, "ac_patch_aggregate: bad layer %d", l);
    AC_REQUIRE(kind[l] == AC_FEAT_TOKENS || kind[l] == AC_FEAT_NCHW || kind[l] == AC_FEAT_NHWC,
               "ac_patch_aggregate: bad kind for layer %d", l);
    prm.layer[l] = AggLayer{feats[l], norm + (size_t)l * B * 2, kind[l], C[l], H[l], W[l]};
    maxL = maxL > C[l] * patchsize * patchsize ? maxL : C[l] * patchsize * patchsize;
    size_t n = (size_t)C[l] * H[l] * W[l];
    int chunks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (chunks > kStatChunks) chunks = kStatChunks;
    double* part_l = partial + (size_t)l * B * kStatChunks * 2;
    hipLaunchKernelGGL(agg_stats_kernel, dim3(chunks, B), dim3(256), 0, stream, feats[l], kind[l], C[l], H[l] * W[l],
                       part_l);
    hipLaunchKernelGGL(agg_stats_finish_kernel, dim3(B), dim3(64), 0, stream, part_l, chunks, (double)n, eps,
                       norm + (size_t)l * B * 2);
  }
  AC_REQUIRE((long long)(pre_dim + 1) * maxL < (1LL << 31) && (long long)(tgt_dim + 1) * n_layers * pre_dim < (1LL << 31),
             "ac_patch_aggregate: pooling index arithmetic would overflow 32 bits");
  // fast path: all layers are token maps on the reference grid, channels a multiple of 4, K in {1, 3}
  bool fast = (patchsize == 1 || patchsize == 3);
  int maxC = 0;
  for (int l = 0; l < n_layers; ++l) {
    fast = fast && kind[l] == AC_FEAT_TOKENS && H[l] == H[0] && W[l] == W[0] && C[l] % 4 == 0;
    maxC = maxC > C[l] ? maxC : C[l];
  }
  // register path: the DINO ViT shape (see agg_pool_tokens_27_8_kernel)
  const bool reg27 = agg_is_reg27(kind, C, H, W, n_layers, patchsize, pre_dim, tgt_dim);
  if (reg27) {
#ifdef AC_AGG_PER_PATCH          // (same-box A/B: round 2's one workgroup per (patch, layer) at every size)
    const bool per_row = false;
#else
    const bool per_row = (long long)H[0] * B * n_layers >= 8LL * ac_device_cus();   // >= 8 row-workgroups per CU
#endif
    const unsigned row_grid = 8u * (unsigned)H[0] * (unsigned)((B * n_layers + 7) / 8);      // 8 XCDs x H0 rows x units per XCD
    if (Zhl) {
      if (per_row) hipLaunchKernelGGL(agg_pool_tokens_27_8_row_kernel<true>, dim3(row_grid), dim3(256), 0, stream, prm, B);
      else hipLaunchKernelGGL(agg_pool_tokens_27_8_kernel<true>, dim3(H[0] * W[0], B, n_layers), dim3(256), 0, stream, prm);
    } else {
      if (per_row) hipLaunchKernelGGL(agg_pool_tokens_27_8_row_kernel<false>, dim3(row_grid), dim3(256), 0, stream, prm, B);
      else hipLaunchKernelGGL(agg_pool_tokens_27_8_kernel<false>, dim3(H[0] * W[0], B, n_layers), dim3(256), 0, stream, prm);
    }
    return ac_check_launch("ac_patch_aggregate(tokens, registers)");
  }
  if (Zhl) return ac_fail(AC_EINVAL, "ac_patch_aggregate_hl32: shape not on the register path");
  if (fast) {
    const bool identity = (n_layers * pre_dim == tgt_dim);
    size_t flds = ((size_t)patchsize * patchsize * (maxC + 4) + (identity ? 0 : (size_t)n_layers * pre_dim)) * sizeof(float);
    if (flds <= 160 * 1024) {
      if (patchsize == 3) {
        if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&agg_pool_tokens_kernel<3>), (int)flds, nullptr)) return rc;
        hipLaunchKernelGGL(agg_pool_tokens_kernel<3>, dim3(H[0] * W[0], B), dim3(256), flds, stream, prm);
      } else {
        if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&agg_pool_tokens_kernel<1>), (int)flds, nullptr)) return rc;
        hipLaunchKernelGGL(agg_pool_tokens_kernel<1>, dim3(H[0] * W[0], B), dim3(256), flds, stream, prm);
      }
      return ac_check_launch("ac_patch_aggregate(tokens)");
    }
  }
  size_t lds = ((size_t)maxL + (size_t)n_layers * pre_dim) * sizeof(float);
  AC_REQUIRE(lds <= 160 * 1024, "ac_patch_aggregate: C*k*k + layers*pre_dim = %zu floats exceeds LDS", lds / 4);
  if (int rc = ac_set_max_lds(reinterpret_cast<const void*>(&agg_pool_kernel), (int)lds, nullptr)) return rc;
  hipLaunchKernelGGL(agg_pool_kernel, dim3(H[0] * W[0], B), dim3(256), lds, stream, prm);
  return ac_check_launch("ac_patch_aggregate");
}

// ------------------------------------------------ stand-alone unfold / pool
__global__ void unfold_kernel(const float* __restrict__ x, int B, int C, int H, int W, int k, float* __restrict__ out) {
  const int k2 = k * k, pad = (k - 1) / 2;
  size_t total = (size_t)B * H * W * C * k2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int kk = (int)(i % k2);
    size_t t = i / k2;
    int c = (int)(t % C); t /= C;
    int p = (int)(t % (H * W));
    int b = (int)(t / (H * W));
    int yy = p / W + kk / k - pad, xx = p % W + kk % k - pad;
    out[i] = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.f : x[(((size_t)b * C + c) * H + yy) * W + xx];
  }
}

extern "C" int ac_unfold_patches(const float* x, int B, int C, int H, int W, int patchsize, float* out, void* stream) {
  AC_REQUIRE(x && out, "ac_unfold_patches: null pointer");
  AC_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && patchsize >= 1 && (patchsize & 1), "ac_unfold_patches: bad shape");
  size_t total = (size_t)B * H * W * C * patchsize * patchsize;
  unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(unfold_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, B, C, H, W, patchsize, out);
  return ac_check_launch("ac_unfold_patches");
}

__global__ void pool_rows_kernel(const float* __restrict__ x, int64_t rows, int L, int O, float* __restrict__ y) {
  const int64_t row = blockIdx.y;
  const float* xr = x + row * L;
  for (int o = blockIdx.x * 256 + threadIdx.x; o < O; o += gridDim.x * 256) {
    const int s = (int)(((long long)o * L) / O);
    const int e = (int)((((long long)(o + 1)) * L + O - 1) / O);
    float acc = 0.f;
    for (int i = s; i < e; ++i) acc += xr[i];
    y[row * O + o] = acc / (float)(e - s);
  }
}

extern "C" int ac_adaptive_avg_pool_rows(const float* x, int64_t rows, int L, int O, float* y, void* stream) {
  AC_REQUIRE(x && y, "ac_adaptive_avg_pool_rows: null pointer");
  AC_REQUIRE(L > 0 && O > 0 && rows >= 0 && rows <= 65535LL * 65535LL, "ac_adaptive_avg_pool_rows: bad shape");
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    int64_t n = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(pool_rows_kernel, dim3((O + 255) / 256, (unsigned)n), dim3(256), 0, as_stream(stream),
                       x + r0 * L, n, L, O, y + r0 * O);
  }
  return ac_check_launch("ac_adaptive_avg_pool_rows");
}
