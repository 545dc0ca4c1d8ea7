/* ac_hip.h - C ABI of the MI355X (gfx950) anomaly-clustering hot path.
 *
 * The reference (KevinWangHP/Anomaly-Clustering) has no FFI: its boundary is the
 * Python call surface used by examples/main.py and examples/test.py.  This
 * library sits *under* that surface: every entry point below replaces the
 * arithmetic of one reference function (cited per function, paths relative to
 * Anomaly-Clustering/), and the `patchcore` package under anomaly-clustering_amd/models/ keeps
 * the reference's Python signatures on top of it (INTEGRATION.md shows the
 * ctypes stubs a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C, no torch types: device pointers + sizes + an opaque HIP stream
 *     (`void* stream` is a hipStream_t; NULL = the null stream).
 *   - no allocation, no synchronisation, no global state inside: the caller owns
 *     every buffer, including the scratch `ws` whose size the matching
 *     *_workspace_bytes() returns.  All pointers are DEVICE pointers unless a
 *     parameter says "host".
 *   - every device pointer must be 16-byte aligned (the kernels use 16-byte loads
 *     and stores); violations are reported as AC_EINVAL, not as a GPU fault.
 *   - returns AC_OK (0) or an AC_E* code; ac_last_error() gives the message of
 *     the calling thread's last failure.
 *   - "hl32" = the split-fp16 operand format of the MFMA kernels: a [rows, K]
 *     fp32 matrix stored as rows of K*4 bytes, each 128-byte group holding 32
 *     high halves (fp16(x)) followed by the 32 low halves (fp16(x - hi)) of the
 *     same 32 columns.  hi+lo carries 22 significand bits, and the 3-pass
 *     product hi*hi + hi*lo + lo*hi accumulated in fp32 by
 *     v_mfma_f32_16x16x32_f16 is at least as accurate as an fp32 FMA chain
 *     (measured: profiles/r01_mfma_probe.txt).
 *   - devices: every call runs on the device that is current on the calling thread
 *     (hipSetDevice) and on the stream it is given; what the library caches (CU
 *     count, kernel attributes) is keyed by that device, so a process may drive
 *     several GPUs.  The ONE piece of process-wide state is the diagnostic GEMM
 *     policy (ac_set_gemm_policy), which never changes results.
 */
#ifndef AC_HIP_H
#define AC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_OK 0
#define AC_EINVAL 1     /* bad shape / alignment / null pointer */
#define AC_EWORKSPACE 2 /* workspace too small */
#define AC_ELAUNCH 3    /* HIP reported a launch error */

/* ABI revision: changes whenever an entry point changes meaning or a new one is added; a binding checks it at load
 * time (models/acnative does: a stale library under a newer binding - or the reverse - fails there, not in a kernel).
 * 500 = round 5: ac_alpha_softmax is the 7-argument form again, the fp32 copy moved to ac_alpha_softmax_f32.
 * 600 = round 6: ac_patch_aggregate_hl32 / ac_patch_aggregate_hl32_supported added. */
#define AC_ABI_VERSION 600
int ac_version(void);
const char* ac_last_error(void);

/* Diagnostic knob, process-wide: which of the two (bit-identical) linear-layer GEMM kernels the ViT
 * entry points launch.  AUTO picks by problem size.  Exists so that tests can prove the two agree
 * and so that same-box A/B timings need no rebuild. */
#define AC_GEMM_AUTO 0
#define AC_GEMM_256x128 1 /* csrc/gemm_hl2.h */
#define AC_GEMM_256x256 2 /* csrc/gemm_hl5.h */
int ac_set_gemm_policy(int policy);

/* ------------------------------------------------------------------ operands */

/* fp32 [rows, K] (row stride `src_ld` floats) -> hl32 [rows, K]; K % 32 == 0.
 * row_sqnorm (nullable): fp32 [rows] = sum_k src[r][k]^2 (fp64 accumulation, rounded once).
 * Used for: weight matrices at model load; Z inside ac_pair_min_dist. */
int ac_pack_hl32(const float* src, int64_t rows, int K, int64_t src_ld, void* dst_hl32,
                 float* row_sqnorm, void* stream);

/* ------------------------------------------------------------- input pipeline */

/* The image transform of the MVTec dataset (models/patchcore/datasets/mvtec.py:74-81), batched:
 *   Resize(resize) (short edge, Pillow's antialiased bilinear, uint8 after each pass) -> CenterCrop(crop) ->
 *   ToTensor (/255) -> Normalize(mean, std)
 * images: DEVICE uint8 [B, H, W, 3] (decoded RGB, all of one size); mean, std: HOST float[3];
 * out: fp32 [B, 3, crop, crop] (NCHW), what ac_vit_prepare_tokens reads.  Bit-identical to the host chain
 * (integer resampling; fp32 division / subtraction / division, IEEE-rounded). */
size_t ac_preprocess_workspace_bytes(int B, int H, int W, int resize, int crop);
int ac_preprocess_images(const uint8_t* images, int B, int H, int W, int resize, int crop, const float* mean,
                         const float* stddev, float* out, void* ws, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------- ViT */

/* Weights of one transformer block, reference models/patchcore/vision_transformer.py:95-113.
 * Matrices are hl32 copies of the torch [out, in] weights. */
typedef struct ac_block_weights {
  const float* ln1_w; const float* ln1_b;          /* [C] */
  const void* qkv_w; const float* qkv_b;           /* hl32 [3C, C], [3C] */
  const void* proj_w; const float* proj_b;         /* hl32 [C, C], [C] */
  const float* ln2_w; const float* ln2_b;          /* [C] */
  const void* fc1_w; const float* fc1_b;           /* hl32 [Hd, C], [Hd] */
  const void* fc2_w; const float* fc2_b;           /* hl32 [C, Hd], [C] */
} ac_block_weights;

/* Scratch for ac_vit_prepare_tokens / ac_vit_block at batch B. */
size_t ac_vit_workspace_bytes(int B, int T, int C, int hidden, int heads, int patch_dim);

/* VisionTransformer.prepare_tokens (vision_transformer.py:196-207) + PatchEmbed (:128-131):
 * images fp32 [B, in_ch, img, img] (NCHW) -> x fp32 [B, T, C], T = (img/patch)^2 + 1.
 * proj_w: hl32 [C, in_ch*patch*patch] (the conv weight flattened (c,ky,kx)); cls [C]; pos [T, C]. */
int ac_vit_prepare_tokens(const float* images, int B, int in_ch, int img, int patch, int C,
                          const void* proj_w, const float* proj_b, const float* cls, const float* pos,
                          float* x, void* ws, size_t ws_bytes, void* stream);

/* Block.forward (vision_transformer.py:107-113): x_out = x + attn(ln1(x)); x_out += mlp(ln2(x_out)).
 * x_in, x_out: fp32 [B, T, C]; may alias.  head_dim must be 64; C % 64 == 0; hidden % 128 == 0. */
int ac_vit_block(const ac_block_weights* w /* host struct of device pointers */, const float* x_in,
                 float* x_out, int B, int T, int C, int heads, int hidden, float eps, void* ws,
                 size_t ws_bytes, void* stream);

/* nn.LayerNorm over the last dim of fp32 [rows, C] (final `norm`, vision_transformer.py:213). */
int ac_layernorm_rows(const float* x, int64_t rows, int C, const float* w, const float* b, float eps,
                      float* y, void* stream);

/* ------------------------------------------------------- convolutional layers */

/* The layers of torchvision's wide_resnet50_2 up to layer3 - the reference's `wideresnet50` backbone
 * (models/patchcore/backbones.py:27: models.wide_resnet50_2(pretrained=True), tapped at layer2 / layer3 through
 * models/patchcore/common.py:211-292) - as GEMMs on the split-fp16 MFMA engine.  Activations: NHWC fp32.
 *
 *   ac_im2col_hl32: x fp32 with ELEMENT strides (sb, sc, sy, sx) - NHWC activations (sc = 1) or the NCHW image -
 *       -> hl32 [B*Ho*Wo, Kpad], column k = (ky*kw + kx)*C + c, zero beyond kh*kw*C and outside the image
 *       (nn.Conv2d's zero padding `pad`); Ho = (H + 2 pad - kh) / stride + 1.  A 1x1 stride-1 convolution's
 *       operand is simply ac_pack_hl32 of the activation.
 *   ac_conv_gemm_bn: out[m][n] = relu?( (sum_k A[m][k] W[n][k]) * alpha[n] + beta[n] (+ res[m][n]) ), fp32 [M, N]:
 *       nn.Conv2d(bias=False) + eval-mode nn.BatchNorm2d as torch's CPU kernel evaluates it
 *       (alpha = weight / sqrt(running_var + eps), beta = bias - running_mean * alpha) + the Bottleneck's residual
 *       and ReLU.  W: hl32 [N, K] = the conv weight permuted to [Cout, kh, kw, Cin] and flattened (zero padded to K).
 *       N % 8 == 0, K % 32 == 0.  res may be NULL; out may alias res.
 *   ac_maxpool_nhwc: nn.MaxPool2d(k, stride, pad) on NHWC fp32, C % 4 == 0. */
int ac_im2col_hl32(const float* x, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int kh,
                   int kw, int stride, int pad, int Kpad, void* out_hl32, void* stream);
int ac_conv_gemm_bn(const void* A_hl32, const void* W_hl32, int64_t M, int N, int K, const float* alpha,
                    const float* beta, const float* res, int relu, float* out, void* stream);
int ac_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, float* out, void* stream);

/* --------------------------------------------------------- patch aggregation */

#define AC_FEAT_TOKENS 0 /* [B, 1+H*W, C] token map, CLS first (ViT) */
#define AC_FEAT_NCHW 1   /* [B, C, H, W] (CNN backbones) */
#define AC_FEAT_NHWC 2   /* [B, H, W, C]: the same map in channels-last memory (what the HIP CNN layers write) */

size_t ac_patch_aggregate_workspace_bytes(int n_layers, const int* C, const int* H, const int* W, int B,
                                          int patchsize, int pre_dim, int tgt_dim);

/* AnomalyClusteringCore._embed after the backbone (models/patchcore/patchcore.py:375-427),
 * PatchMaker.patchify (:439-465), Preprocessing/MeanMapper (models/patchcore/common.py:145-170),
 * Aggregator (common.py:173-183):
 *   per layer: drop CLS -> standardise over (C,H,W) per image (eps) -> zero-padded k x k unfold
 *   (stride 1) -> bilinear resize of the patch grid to layer 0's grid -> adaptive_avg_pool1d
 *   (C*k*k -> pre_dim); concat layers -> adaptive_avg_pool1d (n_layers*pre_dim -> tgt_dim).
 * feats: host array of n_layers device pointers; kind/C/H/W: host int arrays.
 * Z: fp32 [B * H0*W0, tgt_dim]. */
int ac_patch_aggregate(const float* const* feats, const int* kind, const int* C, const int* H,
                       const int* W, int n_layers, int B, int patchsize, int pre_dim, int tgt_dim,
                       float eps, float* Z, void* ws, size_t ws_bytes, void* stream);

/* The same chain with the rows written as hl32 [B * H0*W0, tgt_dim * 4 bytes] - the operand format of the distance
 * kernels (ac_pack_hl32 of the fp32 Z above, byte for byte) - for blocks that are only ever multiplied: the shards a rank
 * receives over the ring (their squared norms travel with them).  Only the token-map shape of the register path (K = 3,
 * 3 * pre_dim == 8 * C per layer, identity final pool, tgt_dim % 32 == 0: the DINO ViT-B/8 configuration);
 * ac_patch_aggregate_hl32_supported says whether a shape qualifies (1 / 0), the operator refuses the others. */
int ac_patch_aggregate_hl32_supported(const int* kind, const int* C, const int* H, const int* W, int n_layers,
                                      int patchsize, int pre_dim, int tgt_dim);
int ac_patch_aggregate_hl32(const float* const* feats, const int* kind, const int* C, const int* H,
                            const int* W, int n_layers, int B, int patchsize, int pre_dim, int tgt_dim,
                            float eps, void* Z_hl32, void* ws, size_t ws_bytes, void* stream);

/* The pieces of the chain above as separate operators, for callers that use the reference's
 * modules one by one (PatchMaker.patchify, MeanMapper.forward, Aggregator.forward): */

/* torch.nn.Unfold(k, stride 1, pad (k-1)/2) + reshape/permute of patchcore.py:447-461:
 * x fp32 [B, C, H, W] -> out fp32 [B, H*W, C, k, k] (zero padded). k odd. */
int ac_unfold_patches(const float* x, int B, int C, int H, int W, int patchsize, float* out, void* stream);

/* F.adaptive_avg_pool1d over the last axis (common.py:168-170, 181-182):
 * x fp32 [rows, L] -> y fp32 [rows, O]; window [floor(i*L/O), ceil((i+1)*L/O)), fp32 sum / length. */
int ac_adaptive_avg_pool_rows(const float* x, int64_t rows, int L, int O, float* y, void* stream);

/* --------------------------------------------------------------------- alpha */

#define AC_REDUCE_MEAN_OTHERS 0 /* unsupervised: mean over key images j != self (utils.py:222-227) */
#define AC_REDUCE_MIN_ALL 1     /* supervised:   min over all key images     (utils.py:230-237) */

size_t ac_pair_min_dist_workspace_bytes(int64_t nq, int64_t nk, int P, int D);

/* Weight_Distance_{Unsupervised,Supervised} for a block of query images (models/patchcore/utils.py:222-237):
 *   w[i][p] = reduce_j  min_q || Zq[i][p] - Zk[j][q] ||_2
 * Zq fp32 [nq, P, D], Zk fp32 [nk, P, D] (may be the same buffer), D % 32 == 0.
 * MEAN_OTHERS skips key image j == i + self_offset (the query's own image in the key list) and
 * divides by the number of remaining images.  w: fp32 [nq, P]. */
int ac_pair_min_dist(const float* Zq, int64_t nq, const float* Zk, int64_t nk, int P, int D, int reduce,
                     int64_t self_offset, float* w, void* ws, size_t ws_bytes, void* stream);

/* The same computation in stages, for callers that keep packed operands resident, split a category into
 * blocks that fit a memory budget, shard it over GPUs (models/patchcore/pipeline.py), or time the MFMA
 * kernel alone.  The reference's per-pair loop being replaced is utils.py:222-237.
 *
 *   ac_pack_hl32(Z, rows = n*P, K = D, ..., row_sqnorm)            -> Z_hl, sq
 *
 *   ac_pair_min_sq_packed: ONE launch of the MFMA GEMM with the fused min, on the rectangle
 *       keys (nk images) x queries (nq images):
 *         minbuf[j][i*P + p]   = min(minbuf[..],   min_q d2(Zq[i][p], Zk[j][q]))     uint32 [nk][nq*P]
 *         minbuf_t[i][j*P + q] = min(minbuf_t[..], min_p d2(Zq[i][p], Zk[j][q]))     uint32 [nq][nk*P]
 *       as the IEEE bits of a non-negative float; the caller presets them to +inf (0x7f800000) with ac_fill_u32.
 *       - minbuf_t == NULL: one direction only (supervised: test patches against the normal bank).
 *       - minbuf_t != NULL ("two-way"): the block between two DIFFERENT image sets of one unsupervised
 *         category is reduced in both directions from a single pass over its tiles, so a category costs
 *         N^2/2 image pairs of MFMA work however it is blocked.
 *       - Zq and Zk the same packed buffer (nq == nk, minbuf_t NULL): the symmetric case; only the upper
 *         triangle of tiles runs, each reduced both ways into minbuf.  part / nparts (0 / 1 = everything):
 *         only the part-th of nparts equal slices of that tile grid - several GPUs each take a slice of one
 *         small symmetric problem into their own minbuf and combine with an element-wise MIN all-reduce.
 *
 *   ac_max_f32 / ac_pair_fixed_shift: the fixed-point scale of the unsupervised sum.  The mean over key
 *       images is accumulated as 64-bit integers (each sqrt scaled by 2^shift and rounded) so that the
 *       result does not depend on how the key images are blocked, ordered or spread over ranks.
 *       max_sq (device, preset to 0): running max of the squared norms of every row of the category
 *       (all-reduce MAX across ranks); shift (device int32) = min(40, 62 - ceil(log2(n_keys*2*sqrt(max_sq)))).
 *
 *   ac_pair_accumulate: fold a minbuf block (nk rows of stride ld, ncols query columns) into acc:
 *       MEAN_OTHERS: acc int64 [ncols] += sum_{j != self_first + col/P} round(sqrt(minbuf[j][col]) * 2^shift)
 *                    (self_first = index, within this block's key rows, of the key image that IS the first
 *                    query image; any value outside [-(ncols/P), nk) skips nothing);  preset acc to 0.
 *       MIN_ALL:     acc uint32 [ncols] = min(acc, min_j minbuf[j][col]);               preset acc to +inf bits.
 *   ac_pair_finalize: w = (float)(acc * 2^-shift / count)  or  sqrt(acc as float).  count = number of key
 *       images that entered the mean (N - 1). */
int ac_pair_min_sq_packed(const void* Zk_hl32, const float* k_sqnorm, int64_t nk, const void* Zq_hl32,
                          const float* q_sqnorm, int64_t nq, int P, int D, uint32_t* minbuf, uint32_t* minbuf_t,
                          int part, int nparts, void* stream);
int ac_fill_u32(uint32_t* dst, int64_t count, uint32_t value, void* stream);   /* presets: +inf bits 0x7f800000, zeros */
int ac_fill_u64(uint64_t* dst, int64_t count, uint64_t value, void* stream);   /* int64 sums (0), float64 +inf bits */
int ac_max_f32(const float* x, int64_t n, float* max_out, void* stream);
int ac_pair_fixed_shift(const float* max_sq, int64_t n_keys, int32_t* shift, void* stream);
int ac_pair_accumulate(const uint32_t* minbuf, int64_t nk, int64_t ld, int64_t ncols, int P, int reduce,
                       int64_t self_first, const int32_t* shift, void* acc, void* stream);
int ac_pair_finalize(const void* acc, int64_t ncols, int reduce, int64_t count, const int32_t* shift, float* w,
                     void* stream);

/* Supervised weights as an EXACT pruned nearest-neighbour search (utils.py:230-237: the min runs over every patch of
 * the normal-image bank, so only the nearest bank patch of each query patch needs an exact distance):
 *   ac_h16_from_hl32: the hi plane of an hl32 image as plain fp16 rows [rows][K] (K % 64 == 0).
 *   ac_nn_slack:      slack (device float [3]): the candidate window of a query a is slack[0] * |a| + slack[1] = 2 delta(a),
 *                     delta(a) = the bound on |d~2 - d2| of pass 1 for that query, from the largest squared norms of
 *                     the two sets (ac_max_f32).  slack[2] != 0: a norm is outside the range in which the fp16 high
 *                     halves are finite (|row| >= 65504, inf or NaN) - pass 1 proves nothing there, and ac_nn_refine
 *                     then treats EVERY query as overflowed (below), so the exhaustive kernel decides.
 *   ac_nn_candidates: pass 1, one MFMA launch on the fp16 hi planes (a third of the 3-pass work, half the bytes):
 *                     keeps the running minimum approx_min[q] of d~2 over all launches (uint32 float bits, preset
 *                     +inf) and appends to cand[q][.] every key row with d~2 <= running minimum + slack; cand_count
 *                     [q] counts them (the caller zeroes it per key set; rows beyond `cap` are counted, not stored).
 *   ac_nn_refine:     pass 2: best_d2[q] = min(best_d2[q], sum_k (q_k - key_k)^2) in fp64 on the split values of the
 *                     hl32 images, over the stored candidates.  A query whose list was longer than cap
 *                     (cand_count[q] > cap) is SKIPPED, overflow_flag (int32 [1]) set and overflow_rows[q] (int32
 *                     [nq_rows], may be NULL; sticky across key sets, the caller zeroes it once per search) set: the
 *                     caller measures those queries against the WHOLE bank with the exhaustive kernel
 *                     (ac_pair_min_sq_packed, one-way) - nothing is read back per key set; one read of overflow_flag
 *                     at the end of the search says whether any of this is needed.  slack: ac_nn_slack's vector (may
 *                     be NULL = no range check).
 *   ac_nn_overflow_compact: idx (int32 [<= n]) = the q with overflow_rows[q] != 0 in ascending order, n_out[0] their
 *                     number - on the device (one workgroup).
 *   ac_gather_rows:   dst row r = src row idx[min(r, n_idx - 1)] for r < n_rows (row_bytes % 4 == 0): the compact
 *                     copy of the overflowed queries (hl32 rows and squared norms), padded to whole images.
 *   ac_nn_scatter_exhaustive: best_d2[idx[i]] = (double) of the exhaustive kernel's min d2 (uint32 float bits) of
 *                     compact query i: those queries equal the un-pruned path bit for bit.
 *   ac_nn_finalize:   w = (float)sqrt(best_d2).
 * The true nearest neighbour always survives pass 1 (proof in csrc/alpha.hip), whatever the order of the tiles. */
int ac_h16_from_hl32(const void* hl32, int64_t rows, int K, void* dst_h16, void* stream);
int ac_nn_slack(const float* max_sq_keys, const float* max_sq_queries, float* slack, void* stream);
int ac_nn_candidates(const void* Zk_h16, const float* k_sq, int64_t nk_rows, const void* Zq_h16, const float* q_sq,
                     int64_t nq_rows, int D, const float* slack, uint32_t* approx_min, int32_t* cand, int32_t* cand_count,
                     int cap, int64_t key_row_offset, void* stream);
int ac_nn_refine(const void* Zk_hl32, int64_t nk_rows, const void* Zq_hl32, int64_t nq_rows, int D, const int32_t* cand,
                 const int32_t* cand_count, int cap, int64_t key_row_offset, double* best_d2, int32_t* overflow_flag,
                 int32_t* overflow_rows, const float* slack, void* stream);
int ac_nn_overflow_compact(const int32_t* overflow_rows, int64_t n, int32_t* idx, int32_t* n_out, void* stream);
int ac_gather_rows(const void* src, int64_t row_bytes, const int32_t* idx, int64_t n_idx, int64_t n_rows, void* dst,
                   void* stream);
int ac_nn_scatter_exhaustive(const uint32_t* acc_bits, const int32_t* idx, int64_t n, double* best_d2, void* stream);
int ac_nn_finalize(const double* best_d2, int64_t n, float* w, void* stream);

/* The softmax half of Matrix_Alpha_* (utils.py:246-254 / 266-274), float64:
 *   tau == 0 (what math.isclose(tau, 0) accepts): alpha = [w == max w] / #ties;
 *   else alpha = k*exp(w/tau) / sum, computed max-shifted (same value, no overflow).
 * w fp32 [N, P] -> alpha fp64 [N, P].
 * ac_alpha_softmax_f32 also writes alpha_f32 (fp32 [N, P], may be NULL): the same values rounded once - what
 * examples/main.py:294 obtains with `matrix_alpha.float()` before the bmm.  (Round 4 had inserted that argument into
 * ac_alpha_softmax itself; a caller built against the 7-argument form would then have passed its stream as a pointer
 * to write through.  The old symbol keeps the old signature; new functionality gets a new name.) */
int ac_alpha_softmax(const float* w, int64_t N, int P, double tau, double k, double* alpha, void* stream);
int ac_alpha_softmax_f32(const float* w, int64_t N, int P, double tau, double k, double* alpha, float* alpha_f32,
                         void* stream);

/* examples/main.py:294-296: X[i] = alpha[i] (fp32 [P]) . Z[i] (fp32 [P, D]) -> fp32 [N, D]. */
int ac_weighted_avg(const float* alpha, const float* Z, int64_t N, int P, int D, float* X, void* stream);

/* scipy pdist(euclidean) as used by Ward linkage (examples/test.py:193-195 via sklearn/scipy):
 * X fp32 [N, D] -> condensed fp64 [N*(N-1)/2], each distance accumulated in fp64 from fp64 differences. */
int ac_pdist(const float* X, int64_t N, int D, double* out, void* stream);

/* ---------------------------------------------------------------- clustering */

/* Ward linkage by the nearest-neighbour chain on condensed float64 distances (what ac_pdist writes):
 * scipy.cluster.hierarchy.linkage(y, method="ward") as reached from examples/test.py:193-195 through sklearn's
 * AgglomerativeClustering -> ward_tree.  Z: float64 [n-1, 4] = (cluster a, cluster b, distance, size), rows
 * sorted by distance (stable), ids relabelled as scipy does - bit-identical to scipy's Z on the same input.
 * `condensed` is not modified.  n <= 46000; workspace ~ 8 n^2 bytes.  bad_input_flag: int32 [1] (device), set to 1
 * when a distance is NaN or infinite (scipy raises ValueError there), to 2 when the Lance-Williams update itself
 * produced NaN and a cluster was left without a comparable neighbour (numerically degenerate input); Z is then left
 * unwritten - the caller checks the flag after synchronising. */
size_t ac_ward_workspace_bytes(int64_t n);
int ac_ward_linkage(const double* condensed, int64_t n, double* Z, int32_t* bad_input_flag, void* ws, size_t ws_bytes,
                    void* stream);

/* Contingency matrix of two integer labelings (values in [0, k_true) / [0, k_pred)) and, from it, sklearn's
 * normalized_mutual_info_score and adjusted_rand_score (examples/test.py:218-219).
 * contingency: uint64 [k_true, k_pred] (out), k <= 4096; scores: float64 [2] = {NMI, ARI}; bad_label_flag: int32 [1],
 * set to 1 when a label lies outside its range (the caller checks it after synchronising). */
int ac_cluster_scores(const int32_t* labels_true, const int32_t* labels_pred, int64_t n, int k_true, int k_pred,
                      uint64_t* contingency, double* scores, int32_t* bad_label_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AC_HIP_H */
