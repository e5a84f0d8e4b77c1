/*
 * librectools_hip.so — C ABI of the MI355X (gfx950) engine behind RecTools'
 * SASRecModel / BERT4RecModel / HSTUModel fit() + recommend() hot path.
 *
 * Conventions:
 *   - every entry point returns an int status (RT_OK == 0); no exception crosses this boundary;
 *   - all pointers are DEVICE pointers unless a parameter says "host"; tensors are row-major fp32,
 *     ids are int64 (the reference's LongTensor ids), CSR column indices are int32;
 *   - the caller owns every buffer (inputs, outputs, workspace); kernels never allocate or free;
 *   - kernels are enqueued on the given hipStream_t, are asynchronous and re-entrant, and never call
 *     hipDeviceSynchronize; the caller synchronises when it reads results;
 *   - `rt_*_workspace_bytes` functions are pure host arithmetic.
 *
 * Each function cites the reference call site (file:line, RecTools v0.17.0) it replaces.  The Python
 * binding a maintainer would add on the reference side is shown in INTEGRATION.md.
 */
#ifndef RECTOOLS_HIP_H
#define RECTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* rt_stream_t; /* == hipStream_t */

enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARG = 1, /* -> ValueError on the Python side */
  RT_ERR_WORKSPACE = 2,   /* workspace missing / too small */
  RT_ERR_LAUNCH = 3,      /* HIP launch error -> RuntimeError */
  RT_ERR_UNSUPPORTED = 4  /* -> NotImplementedError */
};

enum rt_distance { RT_DIST_DOT = 0, RT_DIST_COSINE = 1, RT_DIST_EUCLIDEAN = 2 };

/* library / device introspection (host) */
int rt_version(void);
int rt_device_cu_count(void);
/* text of the last HIP failure recorded by this library on the calling thread ("" if none) */
const char* rt_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * K12  exact full-catalog top-k scoring
 * Replaces the body of TorchRanker.rank — rectools/models/rank/rank_torch.py:119-155
 * (`objects_factors[whitelist]`, `_dot_score/_cosine_score/_euclid_score` :194-208, the dense
 * `filter_pairs_csr.toarray()[:, whitelist]` mask :138-144 and `torch.topk(sorted=True)` :146-152).
 *
 *  users        [*, d] fp32, row stride `user_stride` floats; batch row i is
 *               users[user_rows ? user_rows[i] : i]                  (== subjects_factors[subject_ids])
 *  items        [*, d] fp32, row stride `item_stride`; candidate position p (0 <= p < n_candidates)
 *               is items[whitelist ? whitelist[p] : p]               (== objects_factors[whitelist]);
 *               with whitelist == NULL its item id is p + candidate_id_offset (a contiguous whitelist
 *               [lo, lo+n) is passed as items + lo*item_stride, candidate_id_offset = lo)
 *  filt_indptr  nullable [n_users+1] int64, filt_indices int32 ascending per row, in the id space of
 *               `whitelist` values (full item ids): pairs that must not be recommended
 *  filt_hash    nullable: per-user hash sets over the same CSR (rt_filter_hash_build) — membership in 1-2 loads
 *               instead of a binary search of the row; worth building when many users meet a small catalog
 *  k            1 <= k <= n_candidates (caller clamps, as rank_torch.py:148 does)
 *  out_ids      [n_users, k] int64 item ids (whitelist-mapped), best first; out_scores [n_users, k];
 *               out_counts [n_users] number of valid leading entries (< k only when the filter leaves
 *               fewer than k candidates — the reference drops -inf rows, rank_torch.py:167-171)
 *  Ordering: score descending (ascending distance for EUCLIDEAN); exact ties -> lower position first.
 *  users_per_pass: 32, 64 or 128 users share one pass over the catalog (0 = library default).
 *  d, strides must be multiples of 4 floats and base pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
size_t rt_topk_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t users_per_pass);
size_t rt_filter_hash_bytes(int32_t n_users, int64_t nnz);
int rt_filter_hash_build(const int64_t* filt_indptr, const int32_t* filt_indices, int32_t n_users, int64_t nnz, int32_t* hash,
                         rt_stream_t stream);

int rt_topk_score(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users,
                  const float* items, int64_t item_stride, const int64_t* whitelist, int64_t n_candidates,
                  int64_t candidate_id_offset, int32_t d, int32_t distance, int32_t k,
                  const int64_t* filt_indptr, const int32_t* filt_indices, const int32_t* filt_hash,
                  int64_t* out_ids, float* out_scores, int32_t* out_counts,
                  void* workspace, size_t workspace_bytes, int32_t users_per_pass, rt_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * K12c  Two-stage exact top-k for dot products and cosine similarity (distance 0 / 1) — the regime where rt_topk_score is bound by the f32-input matrix instruction
 * (recommend(): thousands of users per catalog pass).  Same result contract as rt_topk_score — the ids / order / fp32 scores
 * TorchRanker.rank produces (rank_torch.py:77-223) — with the f32-input instruction spent only on candidates:
 *   1. rt_to_hm_rows: "hm image" of fp32 rows — every value x becomes the word (h << 16) | m, h = bf16 truncation of x, m = bf16
 *      truncation of x - h (|x - h - m| < 2^-15 |x|) — plus the rows' fp32 L2 norms.  Same row geometry as the source (dst_stride in
 *      32-bit words >= d); `rows` optional gather (NULL = 0..n-1); normalize = 1: the image of the L2-normalised row (cosine: stage 1
 *      ranks dot products of unit rows, stage 2 evaluates the exact cosine of the fp32 rows).  The catalog's image is built once per
 *      ranker, the users' per call.  normalize bit 1 (values 2 / 3): the H-ONLY image — one round-to-nearest bf16 per value, rows of d / 2
 *      words (dst_stride >= d / 2): half the bytes, for passes bound by HBM (h_only = 1 in rt_topk_score_two_stage: coarse error
 *      2^-7 |u| |v|, items_hm rows item_stride / 2 words apart, d % 64 == 0).
 *   2. rt_topk_score_two_stage: stage 1 streams the images through rt_topk_score's selection machinery (viewed filter / whitelist as
 *      there) with two v_mfma_f32_32x32x16_bf16 per four k — (h + m)(h' + m'), a quarter of the matrix-pipe time — and hands the k_cand
 *      (32 or 64) best COARSE candidates per user to stage 2, which scores them again in the exact arithmetic of rt_topk_score's 32-wide
 *      engine (same instruction, same k order: bit-identical scores) and orders them (score desc, position asc).
 *      |coarse - exact| <= (2^-14 + 5 d 2^-24) |u| |v|; every pair stage 1 dropped had a coarse score <= tau_u (the shared bound at the
 *      end of the pass, and the worst candidate's coarse score when the candidate set is full), so out_unproven[u] = 0 — the k-th exact
 *      score clears tau_u by more than that error — PROVES the outputs of user u are exactly rt_topk_score's; out_unproven[u] = 1 (ties
 *      or near-ties at the k-th place): rank those users with rt_topk_score.  users_hm [n_users, d] dense in call order; items_hm strided
 *      and offset like `items`; d % 32 == 0, k <= 16.  Workspace: rt_topk_two_stage_workspace_bytes.
 * ------------------------------------------------------------------------------------------------ */
int rt_to_hm_rows(const float* src, int64_t src_stride, const int64_t* rows, int64_t n_rows, int32_t d, int32_t normalize, uint32_t* dst,
                  int64_t dst_stride, float* norms, rt_stream_t stream);
/* One-plane image (rt_to_hm_rows mode 2 / 3, rows of src_stride_words words) -> fragment-major: the 16-byte unit (row r, slot 2 s + half)
 * at unit ((r / 32)(d / 16) + s) 64 + 32 half + r % 32, rows n_rows .. rows_pad zero (rows_pad % 128 == 0, d % 16 == 0).  With both images
 * in this form rt_topk_score_two_stage(h_only = 2) loads an item fragment — the A operand of one v_mfma_f32_32x32x16_bf16 — with one
 * coalesced 1 KB read and keeps the user tile in the LDS (no whitelist, d % 128 == 0, items_hm offset by whole 128-row blocks only). */
int rt_one_plane_to_fragments(const uint32_t* src, int64_t src_stride_words, int64_t n_rows, int32_t d, uint32_t* dst, int64_t rows_pad,
                              rt_stream_t stream);
size_t rt_topk_two_stage_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t k_cand, int32_t users_per_pass);
int rt_topk_score_two_stage(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users, const float* items,
                            int64_t item_stride, const uint32_t* users_hm, const uint32_t* items_hm, int32_t h_only, const float* user_norms,
                            float max_item_norm, const int64_t* whitelist, int64_t n_candidates, int64_t candidate_id_offset, int32_t d,
                            int32_t distance, int32_t k, int32_t k_cand, const int64_t* filt_indptr, const int32_t* filt_indices,
                            const int32_t* filt_hash, int64_t* out_ids, float* out_scores, int32_t* out_counts, int32_t* out_unproven,
                            void* workspace, size_t workspace_bytes, int32_t users_per_pass, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K7  dense fp32 GEMM (f32-input MFMA):  C[M,N] = A . B^T (+ bias[n]) (+ R[m,n]) (relu)
 * Replaces nn.Linear / MultiheadAttention in_proj & out_proj / `normed_x @ uvqk_proj` / `session_embs @ item_embs.T`
 * (net_blocks.py:63-64,108-109; sasrec.py:191; ligr.py:56,99-105; hstu.py:258,293; similarity.py:85) and their
 * autograd products.  a_kc / b_kc = 1: operand is [rows, K] row-major with row stride ld ("k-contiguous");
 * = 0: element (r, k) lives at r + k*ld (a transposed view).  split_k > 1 splits the reduction over the grid:
 * every slice writes an [M,N] slab into `workspace` (rt_gemm_workspace_bytes) and a second kernel sums the
 * slabs in a fixed order into C (deterministic, no float atomics; R / relu then not allowed).
 * a_rowsum (optional, [M]; row-contiguous A only): also returns sum_k A(m,k) — the bias gradient db = colsum(dy)
 * of a wgrad product dW = dy^T x, taken from the A tiles already staged in LDS (no second pass over dy).
 * Arithmetic: fp32 in, fp32 accumulate, fp32 out.  On exact tile grids (M, N multiples of 128, K of 32, 16-byte aligned
 * operands) every fp32 product is formed as six bf16 matrix-pipe products of an EXACT 3-way bf16 split of both operands
 * (all terms down to 2^-16 relative; what is dropped is below one fp32 rounding of the product) — error against an fp64
 * product at or below that of the f32-input MFMA (v_mfma_f32_32x32x2_f32), which ragged shapes use and which the
 * environment variable RT_GEMM_SPLIT=exact selects everywhere.
 * ------------------------------------------------------------------------------------------------ */
size_t rt_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t split_k);
int rt_gemm(const float* A, int64_t lda, int32_t a_kc, const float* B, int64_t ldb, int32_t b_kc,
            float* C, int64_t ldc, const float* bias, const float* R, int64_t ldr, float* a_rowsum,
            int32_t M, int32_t N, int32_t K, int32_t relu, int32_t split_k, void* workspace, size_t workspace_bytes,
            rt_stream_t stream);
/* Up to 6 WEIGHT GRADIENTS over the same rows in two launches (all products split-K in one grid + one combine): dW_i [n_out_i, n_in_i]
 * (contiguous) = dy_i^T in_i, db_i [n_out_i] = column sums of dy_i (nullable) — the five weight gradients of a transformer block
 * (lightning.py:296-321: what loss.backward() produces for sasrec.py:191-194's Linear layers) instead of a product and a combine each.
 * splits: upper bound of the split-K factor.  Exact-tile shapes only (n_out, n_in % 128, rows % 32, 16-byte aligned), otherwise
 * RT_ERR_UNSUPPORTED (the caller issues rt_gemm per product).  Same slices and fixed-order combine as rt_gemm(split_k). */
typedef struct rt_wgrad_problem {
  const float* dy; int64_t ldy; const float* in; int64_t ldin; float* dw; float* db; int32_t n_out, n_in;
} rt_wgrad_problem;
size_t rt_wgrad_grouped_workspace_bytes(const rt_wgrad_problem* problems, int32_t n, int32_t rows, int32_t splits);
int rt_wgrad_grouped(const rt_wgrad_problem* problems, int32_t n, int32_t rows, int32_t splits, void* workspace, size_t workspace_bytes,
                     rt_stream_t stream);
/* K7w  the same products with a PRE-SPLIT weight operand (csrc/rt_gemm_wp.hip): rt_split_planes writes the exact three-way bf16 split of
 * a contiguous fp32 range (every weight of a layer stack in one launch; plane p at planes + p * plane_stride, n % 4 == 0, plane_stride % 8
 * == 0), rt_gemm_wp computes up to 4 products C[M,N] = A[M,K] . W' (+ bias) (+ R) (relu) in one launch from the planes: w_tr = 0:
 * W'(n,k) = W[n*ldw + k] (y = x W^T, nn.Linear forward); w_tr = 1: W'(n,k) = W[k*ldw + n] (dx = dy W).  Only the activation operand is
 * split in registers: half the split arithmetic of rt_gemm's loop, same six bf16 products per fp32 product.  Exact tile grids only
 * (M, N % 128 == 0, K % 32 == 0, 16-byte aligned): otherwise RT_ERR_UNSUPPORTED and the caller takes rt_gemm. */
int rt_split_planes(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, rt_stream_t stream);
typedef struct rt_gemm_wp_problem {
  const float* A; int64_t lda; const uint16_t* W; int64_t plane_stride, ldw; float* C; int64_t ldc;
  const float* bias; const float* R; int64_t ldr; int32_t M, N, K, relu;
} rt_gemm_wp_problem;
int rt_gemm_wp(const rt_gemm_wp_problem* problems, int32_t n, int32_t w_tr, rt_stream_t stream);
/* K7f  the feed-forward half of a SASRec block as ONE launch per direction (csrc/rt_ffn.hip; replaces sasrec.py:225-229 +
 * net_blocks.py:63-64 = LayerNorm, Linear, ReLU, Dropout, Linear, Dropout, skip).  A workgroup owns 64 rows from the LayerNorm input to
 * the block output; both products run the K7w loop on the pre-split planes of W1 [dff, d] / W2 [d, dff], masks are applied in the
 * epilogues (the same (seed, stream, float4-group) hash as rt_act_dropout_*), the intermediate goes through L2 only.
 *   fwd: f = LN(y) (+ mean, rstd [M]); hdrop [M, dff] = drop(relu(f W1^T + b1)); out [M, d] = f + drop(hdrop W2^T + b2)
 *   bwd: g_o [M, d] = drop'(g_out) (p > 0 only; else g_o is not written and g_out stands for it); g_h [M, dff] = [hdrop != 0] / (1 - p)
 *        * (g_o W2); g_f [M, d] = g_h W1 + g_out.  Weight gradients (g_o^T hdrop, g_h^T f) and the LayerNorm backward stay with the caller.
 * Shapes: M % 64 == 0, d % 128 == 0 (<= 1024), dff % 128 == 0, contiguous 16-byte aligned arrays; otherwise RT_ERR_UNSUPPORTED
 * (rt_ffn_fused_supported answers without launching). */
int rt_ffn_fused_supported(int32_t M, int32_t d, int32_t dff);
int rt_ffn_fused_fwd(const float* y, const float* ln_w, const float* ln_b, float eps, float* f, float* mean, float* rstd,
                     const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, const float* b1, const float* b2,
                     float* hdrop, float* out, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_h, uint64_t sid_h,
                     uint64_t seed_o, uint64_t sid_o, rt_stream_t stream);
int rt_ffn_fused_bwd(const float* g_out, const float* hdrop, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride,
                     float* g_o, float* g_h, float* g_f, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o, uint64_t sid_o,
                     rt_stream_t stream);
/* The whole tail of a packed SASRec block behind its attention as ONE launch per direction (same kernel family, three products):
 *   fwd: y = q + attn Wo^T + bo; f = LN2(y); hdrop = drop(relu(f W1^T + b1)); out = f + drop(hdrop W2^T + b2)   (sasrec.py:224-229)
 *        training != 0 writes y, f, mean, rstd, hdrop (the backward's inputs); training == 0 (p must be 0; y, mean, rstd, hdrop may be
 *        NULL) writes `out` and the scratch rows `f` only — a row crosses the memory pipe 3 times instead of 12 (recommend()).
 *   bwd: g_o, g_h as rt_ffn_fused_bwd; g_y [M, d] = LN2'(g_h W1 + g_out; y, mean, rstd, ln_w) with the rows still on chip; g_A [M, d] =
 *        g_y Wo.  ln_partial [rt_block_tail_partial_floats(M, d)]: per-workgroup shares of d ln_w / d ln_b, summed by
 *        rt_layernorm_bwd_reduce(ln_partial, M / 64, d, dw, db).  The three weight gradients stay with the caller.
 * wo_planes: planes of the out-projection weight [d, d], same plane_stride as w1_planes / w2_planes.  Shapes as rt_ffn_fused_*. */
int rt_block_tail_fwd(const float* attn, const float* q, const uint16_t* wo_planes, const float* bo, const float* ln_w, const float* ln_b,
                      float eps, float* y, float* f, float* mean, float* rstd, const uint16_t* w1_planes, const uint16_t* w2_planes,
                      int64_t plane_stride, const float* b1, const float* b2, float* hdrop, float* out, int32_t M, int32_t d, int32_t dff,
                      float p, uint64_t seed_h, uint64_t sid_h, uint64_t seed_o, uint64_t sid_o, int32_t training, rt_stream_t stream);
int rt_block_tail_bwd(const float* g_out, const float* hdrop, const float* y, const float* mean, const float* rstd, const float* ln_w,
                      const uint16_t* wo_planes, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, float* g_o,
                      float* g_h, float* g_y, float* g_A, float* ln_partial, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o,
                      uint64_t sid_o, rt_stream_t stream);
size_t rt_block_tail_partial_floats(int32_t M, int32_t d);
int rt_layernorm_bwd_reduce(const float* partial, int32_t blocks, int32_t d, float* dw, float* db, rt_stream_t stream);
/* Up to 4 independent products of the same operand layouts in ONE launch (tile ranges back to back: the tail of one product is
 * filled by the head of the next — the q and k/v projections of a block, sasrec.py:221-224, or two data-gradient products).
 * Problems off the exact-tile path are executed as consecutive rt_gemm calls; results are identical either way. */
typedef struct rt_gemm_problem {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc;
  const float* bias; const float* R; int64_t ldr; int32_t M, N, K, relu;
} rt_gemm_problem;
int rt_gemm_grouped(const rt_gemm_problem* problems, int32_t n, int32_t a_kc, int32_t b_kc, rt_stream_t stream);
/* out[n] += sum_m X[m,n]  (bias gradients; caller zero-fills out) */
int rt_colsum(const float* X, int64_t ld, int32_t M, int32_t N, float* out, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1  batch collation on the device.  The session store of the reference's SequenceDataset
 * (data_preparator.py:39-99) as CSR: offsets [n_sessions+1], items / weights / unix_ts flat, ordered by time inside a
 * session.  One launch cuts a batch `idx` [B] out of it, bit-identical to the reference's collate functions:
 *   mode 0 SASRec train (sasrec.py:86-104): x = tail[:-1], y = tail[1:], yw = weights[1:] of the last L+1 items,
 *          left padded; ts_out [B,L+1] optional (left pad repeats the first kept timestamp)
 *   mode 1 SASRec recommend (sasrec.py:149-166): last L items;  mode 2: with timestamps (last L+1, final row = context)
 *   mode 3 BERT4Rec train (bert4rec.py:109-153): last L items, probs [B,L] uniform draws and rand_ids [B,L] random item
 *          ids decide MASK / random / keep;  mode 4 BERT4Rec recommend (bert4rec.py:182-193): last L-1 items + MASK
 * ------------------------------------------------------------------------------------------------ */
int rt_collate(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* unix_ts, const int64_t* idx,
               int32_t B, int32_t L, int32_t mode, const float* probs, const int64_t* rand_ids, float mask_prob,
               int64_t mask_id, int64_t* x, int64_t* y, float* yw, int64_t* ts_out, rt_stream_t stream);

/* Packed (padding-free) SASRec batch, DESIGN.md §9.0 — the rows of the batch are the REAL positions only (the reference builds the
 * left-padded [B, L] window, sasrec.py:86-104,149-166; 28 % / 45 % of its rows are padding at ML-20M scale).  Session idx[b] owns
 * rows cu_seqlens[b] .. cu_seqlens[b+1]-1, oldest first; the caller cuts cu_seqlens [B+1] from the store's offsets on the host
 * (cu[b+1] - cu[b] = min(length - train, window)) — the row count sizes every buffer of the step, so the host knows it without a
 * device round trip.  rows >= cu[B]: the outputs' row count (tail rows: id 0, target 0, weight 0, dist 0).  train = 1: x = kept
 * tail[:-1], y = tail[1:], yw = weights of y; train = 0: x = the last items.  dist [rows] = distance of a row from its session's
 * end = the index of its positional row (net_blocks.py:388-399). */
int rt_collate_packed(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* idx, const int64_t* cu_seqlens,
                      int32_t B, int32_t rows, int32_t train, int64_t* x, int64_t* y, float* yw, int64_t* dist, rt_stream_t stream);
/* ... the timestamps of a packed SASRec-style batch (sasrec.py:96-104 with `add_unix_ts`): session b gets its kept tail's
 * cu[b+1] - cu[b] + 1 timestamps at ts_out[cu[b] + b ..] (the rows' items and the target of the last row:
 * what the padded [B, L+1] batch holds behind its left pad).  ctx [B] != NULL (recommend with a context, sasrec.py:149-166): the last
 * cu[b+1] - cu[b] items' timestamps followed by ctx[b], the time of the request.  n_out = cu[B] + B entries (the caller knows cu[B]). */
int rt_collate_packed_ts(const int64_t* offsets, const int64_t* unix_ts, const int64_t* idx, const int64_t* cu_seqlens, const int64_t* ctx,
                         int32_t B, int64_t n_out, int64_t* ts_out, rt_stream_t stream);
/* ... the BERT4Rec batch on packed rows (bert4rec.py:109-153, 182-193).  train = 1: cu[b+1] - cu[b] = min(length, window) rows; probs /
 * rand_ids [B, window] are the draws of rt_collate mode 3, read at the row's padded position (b, window - n + j) — the packed batch
 * masks what the padded one masks; y = the item where the position was picked, else 0.  draw_rows [B] or NULL: the row of the draws
 * session b reads (NULL = b; a loop that re-orders a batch keeps every session on the draws of its original slot).  train = 0:
 * min(length, window - 1) + 1 rows, the last one the MASK token. */
int rt_collate_packed_bert(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* idx, const int64_t* cu_seqlens,
                           int32_t B, int32_t rows, int32_t window, int32_t train, const float* probs, const int64_t* rand_ids,
                           const int64_t* draw_rows, float mask_prob, int64_t mask_id, int64_t* x, int64_t* y, float* yw, int64_t* dist,
                           rt_stream_t stream);

/* a11  uniform negatives on the device — CatalogUniformSampler.get_negatives (negative_sampler.py:58-73):
 * out[e] uniform over item ids [low, high), e < n (the [B, L | 1, N] tensor, flat), no rejection of positives.
 * Philox4x32-10 keyed by (seed, offset): the same pair always yields the same batch; pass the step counter as `offset`.
 * `out` 16-byte aligned. */
int rt_sample_negatives(int64_t low, int64_t high, int64_t n, uint64_t seed, uint64_t offset, int64_t* out, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1b item-row producer for feature-aware item nets.
 *   table[i,:] = ids_emb[i,:] + dropout( sum_j cat_emb[emb_bag_inputs[offsets[i] + j], :] ),  j < input_lengths[i]
 * Replaces SumOfEmbeddingsConstructor.forward / get_all_embeddings (item_net.py:361-368,463-482) over
 * IdEmbeddingsItemNet (item_net.py:266-281; ids_emb may be NULL for a features-only net) and CatFeaturesItemNet
 * (item_net.py:101-132: nn.EmbeddingBag(mode="sum") then nn.Dropout).  `emb_bag_inputs`, `offsets` [V] and
 * `input_lengths` [V] are the reference module's own int64 buffers (item_net.py:96-98).  ids_emb / out [V,d],
 * cat_emb [F,d]; d % 4 == 0.
 * Backward: d ids_emb = d table (identity, not a kernel); d cat_emb [F,d] is reduced over the TRANSPOSED structure,
 * which is static and prepared once by the host: `t_items` [nnz] = item ids grouped by feature value (ascending
 * item id inside a group), cut into `n_chunks` chunks that never straddle a group, chunk c = t_items[chunk_ptr[c] :
 * chunk_ptr[c+1]]; feature value f owns chunks feat_chunk_ptr[f] .. feat_chunk_ptr[f+1] (none => zero gradient).
 * Every d_cat row is written exactly once, sums run in a fixed order (no float atomics).  The dropout mask is
 * regenerated from (seed, stream_id).  Workspace: rt_bag_sum_bwd_workspace_bytes = n_chunks * d * 4.
 * ------------------------------------------------------------------------------------------------ */
int rt_bag_sum_fwd(const float* ids_emb, const float* cat_emb, const int64_t* emb_bag_inputs, const int64_t* offsets,
                   const int64_t* input_lengths, int32_t V, int32_t d, float p, uint64_t seed, uint64_t stream_id,
                   float* out, rt_stream_t stream);
size_t rt_bag_sum_bwd_workspace_bytes(int64_t n_chunks, int32_t d);
int rt_bag_sum_bwd(const float* d_out, const int64_t* t_items, const int64_t* chunk_ptr, int64_t n_chunks,
                   const int64_t* feat_chunk_ptr, int32_t F, int32_t d, float p, uint64_t seed, uint64_t stream_id,
                   float* d_cat, void* workspace, size_t workspace_bytes, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K2  embedding gather + inverse positional encoding + dropout
 * out[m,:] = dropout(table[ids[m]] * scale + pos[L-1-(m mod L)])     (pos may be NULL)
 * Replaces `item_embs[sessions]` (torch_backbone.py:245), LearnableInversePositionalEncoding.forward
 * (net_blocks.py:388-399) and emb_dropout (torch_backbone.py:247); the full-table copy of
 * get_all_embeddings (item_net.py:361-368) is not needed.  Backward: positions are counting-sorted by item id
 * and every gtable row [V,d] / gpos row [L,d] is written exactly once (no float atomics; row 0 = padding_idx
 * stays zero, item_net.py:260-264); workspace from rt_embed_bwd_workspace_bytes.  Dropout masks are regenerated
 * from (seed, stream_id).
 * ------------------------------------------------------------------------------------------------ */
int rt_embed_fwd(const int64_t* ids, const float* table, const float* pos, float scale, int32_t M, int32_t L,
                 int32_t d, float p, uint64_t seed, uint64_t stream_id, float* out, rt_stream_t stream);
size_t rt_embed_bwd_workspace_bytes(int32_t M, int32_t V, int32_t d);
/* accumulate = 1: gtable already holds another gradient of the same table (the loss's, lightning.py:311-321 — autograd
 * would add the two with a [V,d] kernel): rows occurring in `ids` are added to in place, no other row is touched. */
int rt_embed_bwd(const int64_t* ids, const float* gout, float scale, int32_t M, int32_t L, int32_t d, int32_t V, float p,
                 uint64_t seed, uint64_t stream_id, float* gtable, int32_t accumulate, float* gpos, void* workspace,
                 size_t workspace_bytes, int32_t prepared, rt_stream_t stream);
/* The counting sort of the rows by id depends on `ids` alone: run ahead of the backward pass (on another stream it keeps eight small
 * launches out of the tail of a training step) it fills `workspace`; rt_embed_bwd / rt_embed_packed_bwd called with prepared = 1 on the
 * SAME workspace and ids then start at the row reductions (prepared = 0: they sort themselves). */
int rt_embed_bwd_prepare(const int64_t* ids, int32_t M, int32_t d, int32_t V, void* workspace, size_t workspace_bytes, rt_stream_t stream);

/* K2 on packed rows: out[m,:] = dropout(table[ids[m]] * scale + pos[dist[m]]) (dist from rt_collate_packed; pos may be NULL).
 * Backward: gtable as rt_embed_bwd; gpos [L,d] (optional, fully overwritten): gpos[t] = sum over the sessions longer than t of
 * the gradient row at distance t from the session's end (cu_seqlens [B+1]). */
int rt_embed_packed_fwd(const int64_t* ids, const int64_t* dist, const float* table, const float* pos, float scale, int32_t M,
                        int32_t d, float p, uint64_t seed, uint64_t stream_id, float* out, rt_stream_t stream);
int rt_embed_packed_bwd(const int64_t* ids, const int64_t* cu_seqlens, int32_t B, const float* gout, float scale, int32_t M, int32_t L,
                        int32_t d, int32_t V, float p, uint64_t seed, uint64_t stream_id, float* gtable, int32_t accumulate,
                        float* gpos, void* workspace, size_t workspace_bytes, int32_t prepared, rt_stream_t stream);

/* K3  LayerNorm over rows of [M,d] (nn.LayerNorm call sites: sasrec.py:221,226,303; net_blocks.py:247,257;
 * ligr.py:90,102; hstu.py:256,291).  mean/rstd [M] are saved for the backward; dx/dw/db are overwritten
 * (per-block partial sums in `workspace`, then a fixed-order reduction: deterministic, no float atomics). */
int rt_layernorm_fwd(const float* x, const float* w, const float* b, float eps, int32_t M, int32_t d, float* y,
                     float* mean, float* rstd, rt_stream_t stream);
/* y = LN(x * (ids != 0)): the timeline mask in front of a LayerNorm applied on the fly (sasrec.py:300,313); x0 (nullable)
 * receives the masked rows (the LayerNorm input the backward reads). */
int rt_layernorm_fwd_masked(const float* x, const int64_t* ids, const float* w, const float* b, float eps, int32_t M, int32_t d,
                            float* x0, float* y, float* mean, float* rstd, rt_stream_t stream);
size_t rt_layernorm_bwd_workspace_bytes(int32_t M, int32_t d);
int rt_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, int32_t M,
                     int32_t d, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes,
                     rt_stream_t stream);
/* The same backward with the passes around a LayerNorm of a transformer block fused (the reference leaves them to autograd:
 * `seqs *= timeline_mask`, the skip connection's gradient add; sasrec.py:300, net_blocks.py:244-259, hstu.py:256,291):
 * mask_dy: dy rows with ids[row] == 0 read as zero;  res (nullable): dx += res;  mask_dx: dx rows with ids[row] == 0
 * written as zero.  ids [M] int64, required when a mask flag is set. */
int rt_layernorm_bwd_fused(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d,
                           float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, rt_stream_t stream);
/* The two halves of rt_layernorm_bwd_fused as calls of their own (same workspace): `rows` writes dx and the per-block partial sums of
 * dw / db, `combine` reduces them into dw [d], db [d] in a fixed order.  dx is what the backward pass waits for; dw / db are read by the
 * optimiser only, so `combine` may be issued on another stream behind an event (the block executors and the Python layer put it on
 * the weight-gradient side stream: lightning.py:214-218 is its only reader). */
int rt_layernorm_bwd_rows(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, const float* res,
                          const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, float* dx, void* workspace,
                          size_t workspace_bytes, rt_stream_t stream);
/* rt_layernorm_bwd_rows over dy * (gscale * upstream[0] / norm[0]) — a sampled loss's unit gradient (d_sess_unit of
 * rt_sampled_loss_fwd_train) taken as it is: the bits rt_sampled_loss_bwd's session half would write, without that launch. */
int rt_layernorm_bwd_rows_scaled(const float* dy, const float* norm, float gscale, const float* upstream, const float* x, const float* w,
                                 const float* mean, const float* rstd, int32_t M, int32_t d, float* dx, void* workspace,
                                 size_t workspace_bytes, rt_stream_t stream);
int rt_layernorm_bwd_combine(const void* workspace, size_t workspace_bytes, int32_t M, int32_t d, float* dw, float* db,
                             rt_stream_t stream);

/* The same LayerNorm over rows that carry ZERO COLUMNS: column c of a row exists iff (c % grp) < grp_real (grp % 4 == 0, d % grp == 0).
 * A model width or head size the kernels cannot tile (n_factors = 50; heads of 25 — the reference accepts any n_factors % n_heads == 0,
 * hstu.py:606-607) runs on rows padded with zero columns, head by head (rectools_amd.nn.DimPlan): statistics, outputs and gradients are
 * those of nn.LayerNorm over the d / grp * grp_real real columns; the other columns of y / dx / dw / db are written as zeros.
 * ids / x0 (both nullable) as rt_layernorm_fwd_masked; res / ids / mask_dy / mask_dx as rt_layernorm_bwd_fused (same workspace). */
int rt_layernorm_fwd_cols(const float* x, const int64_t* ids, const float* w, const float* b, float eps, int32_t M, int32_t d, int32_t grp,
                          int32_t grp_real, float* x0, float* y, float* mean, float* rstd, rt_stream_t stream);
int rt_layernorm_bwd_cols(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, const float* res,
                          const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, int32_t grp, int32_t grp_real,
                          float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, rt_stream_t stream);

/* element-wise streams (n = number of floats, multiple of 4).  kind: 0 none, 1 relu, 2 gelu(erf), 3 silu, 4 sigmoid.
 * y = dropout(act(z)) [+ residual] and its backward (net_blocks.py:63-64; hstu.py:257; dropouts at sasrec.py:228,
 * net_blocks.py:258-260); `residual` (nullable) fuses the skip connection that follows the dropout */
int rt_act_dropout_fwd(const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                       const float* residual, float* y, rt_stream_t stream);
int rt_act_dropout_bwd(const float* dy, const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id,
                       int64_t n, float* dz, rt_stream_t stream);
/* y = dropout(silu(a) * b)   (SwigluFeedForward, net_blocks.py:108) */
int rt_swiglu_fwd(const float* a, const float* b, float p, uint64_t seed, uint64_t stream_id, int64_t n, float* y,
                  rt_stream_t stream);
int rt_swiglu_bwd(const float* dy, const float* a, const float* b, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                  float* da, float* db, rt_stream_t stream);
/* y = x + sigmoid(gz) * dropout(a)   (LiGR gated residual, ligr.py:99-100,104-105); backward gives dgz, da (dx = dy) */
int rt_gate_fwd(const float* x, const float* gz, const float* a, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                float* y, rt_stream_t stream);
int rt_gate_bwd(const float* dy, const float* gz, const float* a, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                float* dgz, float* da, rt_stream_t stream);
/* y = a * alpha + b (b may be NULL) */
int rt_axpy(const float* a, float alpha, const float* b, int64_t n, float* y, rt_stream_t stream);
/* y = a * b * (ids[row] != 0); b, ids optional  (`seqs *= timeline_mask`, sasrec.py:300; hstu.py:256,291) */
int rt_mul_mask(const float* a, const float* b, const int64_t* ids, int32_t d, int64_t n, float* y, rt_stream_t stream);
/* the same over `rows` rows of width d with row strides (floats): operands / results may be column slices of a packed
 * projection buffer (HSTU's u, v, q, k = uvqk.split(...), hstu.py:259-262) */
int rt_mul_mask_ld(const float* a, int64_t lda, const float* b, int64_t ldb, const int64_t* ids, int64_t rows, int32_t d,
                   float* y, int64_t ldy, rt_stream_t stream);

/* K13 one dense Adam step over flat fp32 buffers (torch.optim.Adam semantics, lightning.py:214-218);
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int rt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr, float beta1, float beta2,
                 float eps, float grad_scale, rt_stream_t stream);
/* Same update over n_seg segments of the flat p/m/v buffers, every segment reading its gradient from its own device
 * pointer grads[i] (NULL = parameter without gradient: skipped like torch.optim.Adam does).  offsets/lens/grads are
 * HOST arrays (offsets in floats, multiples of 4).  Lets autograd hand over its gradient tensors as they are. */
int rt_adam_step_segments(float* p, float* m, float* v, int32_t n_seg, const int64_t* offsets, const int64_t* lens,
                          const float* const* grads, int32_t step, float lr, float beta1, float beta2, float eps,
                          float grad_scale, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K4  softmax multi-head attention, forward / backward (torch.nn.MultiheadAttention as called at sasrec.py:222-224,
 * net_blocks.py:248-255, ligr.py:91-98).  q/k/v/o: [B*L, ld] fp32, head h at columns [h*hd, (h+1)*hd); hd % 8 == 0,
 * hd <= 128.  Masks are derived from ids [B,L] (0 = PAD): causal (torch_backbone.py:249-252), keypad (:254), both =
 * merged mask with unmasked diagonal (:172-218).  Dropout (p_drop, seed) acts on the probabilities.
 * lse: [B,H,L] saved log-sum-exp; delta: [B,H,L] workspace.
 * ------------------------------------------------------------------------------------------------ */
int rt_mha_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
               const int64_t* ids, int32_t B, int32_t H, int32_t L, int32_t hd, int32_t causal, int32_t keypad,
               float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse, rt_stream_t stream);
int rt_mha_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
               const float* o, int64_t ldo, const float* dout, int64_t lddo, const float* lse, const int64_t* ids,
               int32_t B, int32_t H, int32_t L, int32_t hd, int32_t causal, int32_t keypad, float p_drop, uint64_t seed,
               float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta,
               rt_stream_t stream);
/* Inference shortcut of the same attention: ONLY the last query (position L-1) of every session — recommend() reads
 * `session_embs[:, -1, :]` (lightning.py:393-397), so the final block needs one query row per session.  q [B, ldq] = one
 * projected query row per session, k / v as above, o [B, ldo].  No dropout; masks as rt_mha_fwd applies them to query L-1. */
int rt_mha_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* ids,
                    int32_t B, int32_t H, int32_t L, int32_t hd, int32_t causal, int32_t keypad, float* o, int64_t ldo,
                    rt_stream_t stream);

/* The three entry points above with the logit scale given by the caller (scale <= 0: 1 / sqrt(hd), torch.nn.MultiheadAttention's): a head
 * padded with zero columns up to a size the kernels tile (hd % 8 == 0) keeps the scale of its REAL size. */
int rt_mha_fwd_scaled(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      const int64_t* ids, int32_t B, int32_t H, int32_t L, int32_t hd, float scale, int32_t causal, int32_t keypad,
                      float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse, rt_stream_t stream);
int rt_mha_bwd_scaled(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      const float* o, int64_t ldo, const float* dout, int64_t lddo, const float* lse, const int64_t* ids,
                      int32_t B, int32_t H, int32_t L, int32_t hd, float scale, int32_t causal, int32_t keypad, float p_drop, uint64_t seed,
                      float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta,
                      rt_stream_t stream);
int rt_mha_last_fwd_scaled(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* ids,
                           int32_t B, int32_t H, int32_t L, int32_t hd, float scale, int32_t causal, int32_t keypad, float* o, int64_t ldo,
                           rt_stream_t stream);

/* K4v  The same causal softmax attention over PACKED sessions (no padding rows; forward only — the recommend() encoder).  Session b
 * owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1 of q / k / v / o, oldest item first.  What the reference's left-padded window adds
 * (torch_backbone.py:245-260, sasrec.py:186-231: pad keys are visible to every real query of a causal SASRec block) is closed-form:
 * a pad key / value row equals the projection bias in every session and block (the block input is masked to 0), so ONE virtual
 * key per query — logit q.bk / sqrt(hd), value bv, multiplicity window - n_b — reproduces the padded softmax.  bk / bv [H*hd] =
 * in_proj_bias[d:2d] / [2d:3d]; pass NULL for both when pad keys are masked (key-padding masks).  max_len >= the longest session
 * (sizes the grid: ceil(max_len / 64) owner blocks per session and head; any length — the partner rows stream through 48 KB of LDS);
 * window = the reference's session_max_len.  hd in {32, 64, 128}, RT_ERR_UNSUPPORTED otherwise. */
int rt_mha_varlen_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                      int32_t max_len, int32_t window, float* o, int64_t ldo, rt_stream_t stream);
/* Training pair of the packed attention.  Forward: + attention dropout (counter-based masks keyed by (seed, session*H + head,
 * query, key pair), numbered inside the session; the window's pad keys are dropped one by one like real keys, numbered behind
 * them) and lse [N, H].  Backward: dq / dk / dv rows of the sessions fully overwritten; delta [N, H] workspace; dbv_part
 * [B, H*hd] (or NULL) receives per-session partials of the value-bias gradient contributed by the pad keys — sum over B and add
 * to the bias gradient of the real rows.  The key-bias gradient of the padded window is identically zero (b_k shifts every logit
 * of a query alike), so a caller on packed rows zeroes it instead of taking colsum(dk). */
int rt_mha_varlen_train_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            rt_stream_t stream);
int rt_mha_varlen_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                      const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, const float* bk, const float* bv,
                      int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* dq,
                      int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta, float* dbv_part,
                      rt_stream_t stream);
/* Bidirectional attention inside every packed session (no causal mask, no pad keys): BERT4Rec's key-padding-masked window
 * (torch_backbone.py:254, bert4rec.py:200) on packed rows.  lse != NULL: training forward (dropout, lse [N,H] kept); NULL: inference.
 * hd in {32, 64, 128}, any max_len; RT_ERR_UNSUPPORTED otherwise. */
int rt_mha_varlen_bidir_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                            int32_t B, int32_t H, int32_t hd, int32_t max_len, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            rt_stream_t stream);
int rt_mha_varlen_bidir_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                            const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, float p_drop, uint64_t seed, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                            int64_t lddv, float* delta, rt_stream_t stream);
/* K4v3p  Causal attention over packed sessions BEHIND A SHARED PAD PREFIX.  A stack that neither masks pad keys nor re-zeroes pad rows
 * between blocks (LiGR: ligr.py:161-191 — the reference's default eSASRec; Pre-LN without a key-padding mask: net_blocks.py:290-310)
 * gives the left pads of its [B, L] window a state that real queries read (torch_backbone.py:245-260: causal mask only).  That state
 * depends on the POSITION only — a pad row sees pad rows, and every session's pads start from the same rows (zero item row + positional
 * row) — so the packed batch carries it once: session number n_prefixed of cu_seqlens is the window's `window` positions as pads (B
 * counts it, and any sessions behind it, e.g. the unused tail of the row block).  Sessions 0 .. n_prefixed - 1 (n_b rows each) see the
 * prefix's first window - n_b rows as keys in front of their own; keys and queries are numbered by window position (dropout masks).
 * lse NULL: inference.  Backward: as rt_mha_varlen_bwd; the prefix rows' dk / dv receive the sum over the sessions behind them through
 * `workspace` (rt_mha_varlen_prefix_bwd_workspace_bytes: 16 groups x window x 2 H hd floats; fixed summation order, no atomics).
 * hd in {32, 64, 128}; max_len >= window. */
int rt_mha_varlen_prefix_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                             const int64_t* cu_seqlens, int32_t B, int32_t n_prefixed, int32_t H, int32_t hd, int32_t max_len,
                             int32_t window, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse, rt_stream_t stream);
size_t rt_mha_varlen_prefix_bwd_workspace_bytes(int32_t window, int32_t H, int32_t hd);
int rt_mha_varlen_prefix_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o,
                             int64_t ldo, const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, int32_t B,
                             int32_t n_prefixed, int32_t H, int32_t hd, int32_t max_len, int32_t window, float p_drop, uint64_t seed,
                             float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta, void* workspace,
                             size_t workspace_bytes, rt_stream_t stream);
/* ... for the LAST query of every session only: q [B, ldq] one projected query row per session, o [B, ldo] (cf. rt_mha_last_fwd) */
int rt_mha_varlen_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                           int32_t max_len, int32_t window, float* o, int64_t ldo, rt_stream_t stream);
/* ... the same query WITHOUT key / value rows (round 6; the final block of recommend(), sasrec.py:221-224 at the last position only):
 * q_h.(W_k,h x_j + b_k,h) = (W_k,h^T q_h).x_j + const and sum_j p_j (W_v,h x_j + b_v,h) = W_v,h (sum_j p_j x_j) + b_v,h, so the caller hands
 * qk [B, H, d] = W_k,h^T q_h (one small product per head), the kernel makes ONE pass over the block input x (packed rows [*, ldx]) and
 * returns xbar [B, H, d] = sum_j softmax_j(qk.x_j / sqrt(d / H)) x_j, to which the caller applies W_v,h (+ b_v,h).  pad_keys != 0: the
 * window's pad keys take part with logit 0 (x = 0).  prefix_row >= 0 (pad_keys = 0, max_len >= window): rows prefix_row .. + window - 1 of x
 * are the window's pad rows carried ONCE (the shared pad prefix of rt_mha_varlen_prefix_fwd): a session of n rows sees the first window - n
 * of them as keys in front of its own; -1: none.  d in {64, 128, 256, 512}; other widths: RT_ERR_UNSUPPORTED (use the form above). */
int rt_mha_varlen_last_x_fwd(const float* qk, const float* x, int64_t ldx, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t d,
                             int32_t max_len, int32_t window, int32_t pad_keys, int64_t prefix_row, float* xbar, rt_stream_t stream);
/* E [d, H d] = the head-expanded copy of a [d, d] projection weight (E[r, h d + c] = W[r, c] if r / (d / H) == h else 0): with it the per-head
 * products around rt_mha_varlen_last_x_fwd are ONE exact-tile rt_gemm each — qk = Q E(W_k) (E as [K = d, N = H d]) and
 * attention output = xbar E(W_v)^T + b_v (E as [N = d, K = H d]). */
int rt_mha_last_x_expand(const float* W, int32_t d, int32_t H, float* E, rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Native executor of one PACKED SASRec block (sasrec.py:186-231, :300; csrc/rt_block.hip): the whole launch sequence of the block's
 * training forward, its backward and its inference form behind ONE call each — the reference leaves the sequence to the autograd
 * engine and the Python interpreter; at C2 the interpreter needed longer to issue a step than the GPU to run it.
 * rows: row count of x / out (rows_real of them belong to sessions, the rest is the unused tail up to the 128-row GEMM tile);
 * cu [B+1]; window = session_max_len; pad_keys: no key-padding masks, the window's pad keys are the attention's virtual key.
 * Parameters are the reference's state_dict tensors of the block.  seed_* / sid_*: dropout streams.
 *   _fwd:   out [rows,d]; `saved` (rt_sasrec_block_saved_floats floats) keeps the activations for the backward.
 *   _bwd:   g_x [rows,d]; `grads` = flat parameter gradient in the order ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, w1, b1,
 *           w2, b2 at rt_sasrec_block_grad_offsets (13 entries, the last = total floats).  Weight gradients are issued on a
 *           library-owned side stream when use_side != 0: call rt_side_join(stream) before reading `grads`; x, saved, g_out,
 *           scratch (rt_sasrec_block_bwd_scratch_bytes) and grads must stay alive until then.
 *   _infer: eval mode; last_rows == NULL: out [rows,d]; last_rows [B]: the output at those rows only, out [B,d] (what
 *           recommend() keeps, lightning.py:393-397); scratch: rt_sasrec_block_infer_scratch_floats floats.  q_in / Q_in / kv_in
 *           (nullable, with last_rows == NULL; q_in and Q_in come together and need kv_in: x may then be NULL): LN1(x) [rows, d], the
 *           projected queries [rows, d] and the block's keys | values [rows, 2d] handed in (rt_embed_block1_fwd) — the FIRST block of recommend() reads embedding row +
 *           positional row, so W_kv (e + p) + b_kv = (W_kv e) + (W_kv p + b_kv) is a gather from two projected tables
 *           (rt_embed_packed_fwd over them) and only the query projection runs over the rows.  With last_rows the final block needs no
 *           key / value rows at all (rt_mha_varlen_last_x_fwd).
 * rt_timing_enable(1|2) brackets every internal launch with HIP events (2: weight gradients on the caller's stream);
 * rt_timing_collect synchronises and returns (id, ms, M N K) records: ids 0 gemm, 1 gemm_grouped, 2 layernorm_fwd, 3 layernorm_bwd,
 * 4 act_dropout_fwd, 5 act_dropout_bwd, 6 mha_varlen_train_fwd, 7 mha_varlen_bwd, 8 mha_varlen_last_fwd, 9 misc.
 * ------------------------------------------------------------------------------------------------ */
typedef struct rt_sasrec_block {
  int32_t rows, rows_real, B, H, d, dff, window, pad_keys;
  float p_drop, eps1, eps2;
  uint64_t seed_attn, seed_h, sid_h, seed_o, sid_o;
  const int64_t* cu;
  const float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
  /* optional bf16 planes of in_w / out_w / w1 / w2 (rt_split_planes; plane p at + p * wp_stride elements), NULL: rt_gemm everywhere */
  const uint16_t *in_wp, *out_wp, *w1_wp, *w2_wp;
  int64_t wp_stride;
} rt_sasrec_block;
size_t rt_sasrec_block_saved_floats(int32_t rows, int32_t d, int32_t dff, int32_t H, int32_t with_dropout);
size_t rt_sasrec_block_bwd_scratch_bytes(int32_t rows, int32_t B, int32_t d, int32_t dff, int32_t H, int32_t wgrad_splits);
void rt_sasrec_block_grad_offsets(int32_t d, int32_t dff, int64_t* offsets13);
int rt_sasrec_block_packed_fwd(const rt_sasrec_block* blk, const float* x, float* saved, float* out, rt_stream_t stream);
int rt_sasrec_block_packed_bwd(const rt_sasrec_block* blk, const float* x, const float* saved, const float* g_out, float* g_x, float* grads,
                               void* scratch, size_t scratch_bytes, int32_t wgrad_splits, int32_t use_side, rt_stream_t stream);
size_t rt_sasrec_block_infer_scratch_floats(int32_t rows, int32_t B, int32_t d, int32_t dff, int32_t last_only);
int rt_sasrec_block_packed_infer(const rt_sasrec_block* blk, const float* x, const float* q_in, const float* Q_in, const float* kv_in,
                                 const int64_t* last_rows, float* scratch, float* out, rt_stream_t stream);
/* The first block's inputs for that call, made WITHOUT a product over the rows: x = scale E[id] + P[dist] in registers -> LayerNorm statistics
 * -> q = LN1(x), Q = rstd (scale QE[id] + QP[dist] - mean wg) + wb, K | V = scale KVE[id] + KVP[dist], with the projected tables
 * QE = (E diag(g)) W_q^T, QP = (P diag(g)) W_q^T, wg = W_q g, wb = W_q beta + b_q, KVE = E W_kv^T, KVP = P W_kv^T + b_kv made once per
 * recommend() call (g / beta: LN1's weight / bias; sasrec.py:221-224 + net_blocks.py:388-399). */
int rt_embed_block1_fwd(const int64_t* ids, const int64_t* dist, const float* E, const float* P, float scale, const float* ln_w, const float* ln_b,
                        float eps, const float* QE, const float* QP, const float* wg, const float* wb, const float* KVE, const float* KVP, int32_t M,
                        int32_t d, float* q_out, float* Q_out, float* KV_out, rt_stream_t stream);
/* ... for a Pre-LN / LiGR first block (net_blocks.py:236-262, ligr.py:161-191: q, k AND v read LN1(x), the skip branch reads x): x [M, d]
 * and qkv [M, 3d] = rstd (scale QKVE[id] + QKVP[dist] - mean wg) + wb with QKVE = (E diag(g)) W^T [V, 3d], QKVP = (P diag(g)) W^T [L, 3d],
 * wg = W g, wb = W beta + b over the packed in_proj parameters W [3d, d], b [3d].  d <= 512. */
int rt_embed_block1_preln_fwd(const int64_t* ids, const int64_t* dist, const float* E, const float* P, float scale, float eps, const float* QKVE,
                              const float* QKVP, const float* wg, const float* wb, int32_t M, int32_t d, float* x_out, float* qkv_out,
                              rt_stream_t stream);

/* One packed SASRec TRAINING STEP (lightning.py:311-321 around sasrec.py:271-304, the sampled losses lightning.py:164-212 and
 * torch.optim.Adam, lightning.py:366-369): rt_embed_packed_fwd -> rt_sasrec_block_packed_fwd x n_blocks -> rt_layernorm_fwd ->
 * rt_sampled_loss_fwd_train -> rt_loss_reduce -> rt_sampled_loss_bwd (session half on `stream`, table half on the library's side stream)
 * -> rt_layernorm_bwd_rows / _combine -> rt_sasrec_block_packed_bwd x n_blocks -> rt_embed_packed_bwd (adds into the loss's table
 * gradient) -> rt_side_join -> rt_adam_step_segments: the entry points above in the order the autograd nodes of rectools_amd/ops.py issue
 * them, from compiled code (csrc/rt_step.hip).  No allocation: `arena` (rt_sasrec_step_arena_bytes) holds the parameter gradients
 * (fixed layout) and every activation / workspace of a step of `rows` rows; it may be reused by the next step on the same stream.
 *   rows: row count of the packed batch (multiple of 128); cu [B + 1]: the sessions' row offsets (the lookup's backward); cu_attn
 *   [B_attn + 1] / rows_real: what the blocks' attention sees (B_attn = B + 1 and rows_real = rows when the unused tail of the row block
 *   rides along as one more session, else cu / B / the sessions' row count); blocks [n_blocks]: parameter pointers, planes, eps1 / eps2
 *   and the dropout streams of this step (geometry fields are filled in by the call); loss: 0 BCE, 1 gBCE, 2 sampled softmax;
 *   upstream: device scalar d loss (1.0); loss_out: device [2] <- (loss, normaliser); pos may be NULL (pos_rows == window otherwise).
 *   seg_role [n_seg]: which gradient segment i of the flat parameter buffer reads: 0 table, 1 pos, 2 / 3 the last LayerNorm's weight /
 *   bias, 16 + 12 b + j = block b's parameter j in rt_sasrec_block_grad_offsets order, -1 none (the segment is skipped).
 * rt_sasrec_step_run(phase): 1 = forward + loss + backward, 2 = join + Adam, 3 = both (n_seg <= 1024). */
typedef struct rt_sasrec_step {
  int32_t n_blocks, rows, rows_real, B, B_attn, V, d, dff, H, window, pad_keys, n_neg, loss, cosine, wgrad_splits, pos_rows;
  float p_emb, p_blk, emb_scale, eps_last, logits_t;
  double gbce_beta;
  uint64_t seed_emb, sid_emb;
  const int64_t *ids, *dist, *y, *neg, *cu, *cu_attn;
  const float* yw;
  const float *table, *pos, *lnf_w, *lnf_b;
  const rt_sasrec_block* blocks;
  const float* planes_src; int64_t planes_n; uint16_t* planes; int64_t planes_stride;     /* rt_split_planes of the stack's weights, or NULL */
  const float* upstream;
  float* loss_out;
  void* arena; size_t arena_bytes;
  float *flat_p, *adam_m, *adam_v;
  int32_t n_seg; const int64_t *seg_offsets, *seg_lens; const int32_t* seg_role;
  int32_t adam_step; float lr, beta1, beta2, adam_eps;
} rt_sasrec_step;
size_t rt_sasrec_step_arena_bytes(const rt_sasrec_step* step);
int rt_sasrec_step_run(const rt_sasrec_step* step, int32_t phase, rt_stream_t stream);
/* out [n_seg]: where phase 1 leaves the gradient of every segment (device pointers into the arena; NULL: none) — what phase 2 reads. */
int rt_sasrec_step_grad_ptrs(const rt_sasrec_step* step, const float** out);

/* One packed Pre-LN block (net_blocks.py:223-262, BERT4Rec's stack) under key-padding masks — packed rows have no pad keys:
 *   h = LN1(x); qkv = h Win^T + bin; A = attention(qkv) (causal = 0: every query sees its whole session, rt_mha_varlen_bidir_*);
 *   x1 = x + drop(A Wo^T + bo); g = LN2(x1); a = drop(gelu(g W1^T + b1)); x2 = x1 + drop(a W2^T + b2); out = drop(x2).
 * Same conventions as rt_sasrec_block (rows / rows_real / cu / planes); the five dropout streams in forward order.  The flat parameter
 * gradient uses rt_sasrec_block_grad_offsets(d, dff) (in_w is [3d, d] in both).  Training forward / backward only. */
typedef struct rt_preln_block {
  int32_t rows, rows_real, B, H, d, dff, window, causal;
  float p_drop, eps1, eps2;
  uint64_t seed_attn, seed1, sid1, seed_h, sid_h, seed2, sid2, seed3, sid3;
  const int64_t* cu;
  const float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
  const uint16_t *in_wp, *out_wp, *w1_wp, *w2_wp;
  int64_t wp_stride;
} rt_preln_block;
size_t rt_preln_block_saved_floats(int32_t rows, int32_t d, int32_t dff, int32_t H);
size_t rt_preln_block_bwd_scratch_bytes(int32_t rows, int32_t d, int32_t dff, int32_t H, int32_t wgrad_splits);
int rt_preln_block_packed_fwd(const rt_preln_block* blk, const float* x, float* saved, float* out, rt_stream_t stream);
int rt_preln_block_packed_bwd(const rt_preln_block* blk, const float* x, const float* saved, const float* g_out, float* g_x, float* grads,
                              void* scratch, size_t scratch_bytes, int32_t wgrad_splits, int32_t use_side, rt_stream_t stream);
int rt_side_join(rt_stream_t stream);
/* `stream` waits for everything issued so far on the side stream; the side stream's bookkeeping is left alone (rt_side_join still joins):
 * a data-parallel step starts the exchange of the block weights' gradients from its own stream while the backward pass runs on
 * (the reference: DDP's bucketed all-reduce behind `Trainer.fit`, transformers/base.py:367-380). */
int rt_side_reach(rt_stream_t stream);
/* The library's side stream itself (NULL through *side_out when RT_SIDE_STREAM=0): a binding with side work of its own wraps this stream
 * instead of creating another (more streams than hardware queues = two of them share one). */
int rt_side_stream(void** side_out);
/* the side stream for the caller's own optimiser-only work: it waits for `stream`'s current position; *side_out = its handle, or
 * NULL when disabled (launch on `stream` then).  Joined by rt_side_join. */
int rt_side_fork(rt_stream_t stream, void** side_out);
/* A point on the side stream that a launch on another stream can wait for without waiting for what the side stream is given afterwards
 * (rt_side_join waits for everything issued so far): rt_side_mark records it, rt_side_wait_mark makes `stream` wait for the latest one
 * (no-op when none was recorded since the last join). */
int rt_side_mark(void);
int rt_side_wait_mark(rt_stream_t stream);
int rt_timing_enable(int32_t mode);
int rt_timing_collect(int32_t* ids, float* ms, int64_t* tags, int32_t max_records, int32_t* n_out);

/* K5/K6  HSTU pointwise attention with in-kernel relative time/position bias (hstu.py:84-128, 270-288).
 * ts [B,L+1] int64 (NULL: no time bias); time_w [n], n = num_buckets + 1; time_thr [148] = smallest |dt| of each of the 147 buckets an
 * int64 difference can fall in (unclamped), then n: a later bucket reads time_w[n - 1] — the reference's clamp — computed on
 * the host with the reference's float32 log(|dt|)/0.301 truncation; pos_w [2L-1] (NULL: no position bias).
 * Backward accumulates d_time_w [n] / d_pos_w [2L-1] (caller zero-fills). */
int rt_hstu_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                     const int64_t* ids, const int64_t* ts, const float* time_w, const int64_t* time_thr,
                     const float* pos_w, int32_t B, int32_t H, int32_t L, int32_t hd, float* o, int64_t ldo,
                     rt_stream_t stream);
int rt_hstu_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                     const float* dout, int64_t lddo, const int64_t* ids, const int64_t* ts, const float* time_w,
                     const int64_t* time_thr, const float* pos_w, int32_t B, int32_t H, int32_t L, int32_t hd,
                     float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* d_time_w,
                     float* d_pos_w, rt_stream_t stream);
/* ... over PACKED sessions (no pad rows): session b owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1 of q / k / v / o and
 * the cu[b+1] - cu[b] + 1 timestamps ts[cu[b] + b ..] (rt_collate_packed_ts); window = session_max_len (the 1 / L of hstu.py:284 and the
 * position table [2 window - 1]).  The reference zeroes pad rows before the bias-free projection (hstu.py:256-262): a pad key's v row is
 * silu(0) = 0, it contributes nothing, so the packed form equals the padded one on every real row.  Ring kernels only (hd 32 / 64,
 * 16-byte aligned rows), RT_ERR_UNSUPPORTED otherwise. */
int rt_hstu_attn_varlen_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const int64_t* cu_seqlens, const int64_t* ts, const float* time_w, const int64_t* time_thr,
                            const float* pos_w, int32_t B, int32_t H, int32_t window, int32_t hd, float* o, int64_t ldo,
                            rt_stream_t stream);
int rt_hstu_attn_varlen_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const float* dout, int64_t lddo, const int64_t* cu_seqlens, const int64_t* ts, const float* time_w,
                            const int64_t* time_thr, const float* pos_w, int32_t B, int32_t H, int32_t window, int32_t hd,
                            float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* d_time_w,
                            float* d_pos_w, rt_stream_t stream);
/* the same for the LAST query of every session only (inference, see rt_mha_last_fwd): q [B, ldq], o [B, ldo] */
int rt_hstu_attn_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* ids,
                          const int64_t* ts, const float* time_w, const int64_t* time_thr, const float* pos_w, int32_t B,
                          int32_t H, int32_t L, int32_t hd, float* o, int64_t ldo, rt_stream_t stream);
/* ... over PACKED sessions (round 6: the final STU block of recommend(), hstu.py:270-288 at the last position): k / v packed rows,
 * cu_seqlens [B+1], ts packed as rt_hstu_attn_varlen_fwd reads it, window = session_max_len, max_len >= the longest session */
int rt_hstu_attn_varlen_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                 const int64_t* cu_seqlens, const int64_t* ts, const float* time_w, const int64_t* time_thr,
                                 const float* pos_w, int32_t B, int32_t H, int32_t window, int32_t hd, int32_t max_len, float* o, int64_t ldo,
                                 rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K8/K9  negative-sampled losses without the [B,L,1+N,d] gather (similarity.py:88-95 + lightning.py:164-212).
 * loss: 0 BCE, 1 gBCE, 2 sampled_softmax.  sess [M,d]; table [V,d]; y [M] (0 = position ignored); neg [M,N]; w [M].
 * Forward writes logits [M,1+N] (already divided by logits_t) and the weighted per-position loss.
 * Training forward (fwd_train) additionally keeps, in `workspace`, the unit gradient of every logit and the
 * counting-sort ranks of the candidate ids, and writes d_sess_unit [M,d]: the candidate rows are gathered ONCE for
 * logits, loss and session gradient (the reference materialises the gather in forward and re-reads it in backward).
 * Backward scales by gscale / norm (norm = rt_loss_reduce's out+1) and overwrites d_sess [M,d] and d_table [V,d]
 * (index_put of the reference, done as a counting sort by item id + one segmented reduction per table row: no float
 * atomics).  One backward per training forward.
 * ------------------------------------------------------------------------------------------------ */
int rt_sampled_loss_fwd(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                        const float* w, int32_t M, int32_t N, int32_t d, int32_t loss, int32_t cosine, float logits_t,
                        double gbce_beta, float* logits, float* loss_pos, rt_stream_t stream);
size_t rt_sampled_loss_bwd_workspace_bytes(int32_t M, int32_t N, int32_t V, int32_t d);
int rt_sampled_loss_fwd_train(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                              const float* w, int32_t M, int32_t N, int32_t d, int32_t V, int32_t loss, int32_t cosine,
                              float logits_t, double gbce_beta, float* logits, float* loss_pos, float* d_sess_unit,
                              int64_t ld_du, void* workspace, size_t workspace_bytes, int32_t prepared, rt_stream_t stream);
/* The counting sort of the (position, candidate) pairs by candidate id depends on y / neg alone: run ahead of the forward pass (on
 * another stream: six small launches that would otherwise sit between the forward and the backward kernels) it leaves the ranks, the
 * segment offsets and the popular ids' chunks in `workspace`; fwd_train / bwd called with prepared = 1 on the SAME workspace skip their
 * share (the forward writes the pair records, the backward starts at the row reductions). */
int rt_sampled_loss_prepare(const int64_t* y, const int64_t* neg, int32_t M, int32_t N, int32_t d, int32_t V, void* workspace,
                            size_t workspace_bytes, rt_stream_t stream);
/* d_sess or d_table may be NULL: the two halves are independent (d_sess is a scaled copy of d_sess_unit; d_table consumes the
 * ranks in `workspace`, so ask for it exactly once per forward) and may be issued on different streams.  The gradients are scaled by
 * gscale * upstream[0] / norm[0]: `upstream` (nullable: 1) is dL/dloss ON THE DEVICE — autograd's root gradient is never read by the host. */
int rt_sampled_loss_bwd(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                        int32_t M, int32_t N, int32_t d, int32_t V, int32_t cosine, float logits_t, const float* logits,
                        const float* norm, float gscale, const float* upstream, const float* d_sess_unit, int64_t ld_du, float* d_sess,
                        int64_t ld_dsess, float* d_table, void* workspace, size_t workspace_bytes, int32_t prepared, rt_stream_t stream);
/* out[0] = sum(loss_pos)/normaliser, out[1] = normaliser; mode 0: count(loss_pos > 0) (lightning.py:159-161),
 * mode 1: count(y != 0) (lightning.py:197-198) */
int rt_loss_reduce(const float* loss_pos, const int64_t* y, int32_t M, int32_t mode, float* out, rt_stream_t stream);
/* K10 full-catalog softmax rows (lightning.py:145-162) on the logits of the R active positions produced by rt_gemm:
 * grad = 0: loss_pos[r] = (lse - z_y) * w, lse[r];  grad = 1: logits := (softmax - onehot) * w * gscale * upstream[0] / (norm * t)
 * (upstream: nullable device scalar, as in rt_sampled_loss_bwd) */
int rt_softmax_ce_rows(float* logits, int64_t ld, int32_t R, int32_t V, const int64_t* y_act, const float* w_act,
                       float logits_t, int32_t grad, const float* norm, float gscale, const float* upstream, float* loss_pos, float* lse,
                       rt_stream_t stream);
/* K11 L2 row normalisation with max(||x||, 1e-8) (similarity.py:97-100) and its backward */
int rt_l2norm_fwd(const float* x, int64_t ldx, int32_t M, int32_t d, float* y, int64_t ldy, float* nrm, rt_stream_t stream);
int rt_l2norm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t M, int32_t d, int32_t accumulate,
                  float* dx, int64_t lddx, rt_stream_t stream);
/* dst[r] = src[idx[r]] ; dst[idx[r]] = src[r] (idx unique) */
int rt_gather_rows(const float* src, int64_t ld_src, const int64_t* idx, int32_t R, int32_t d, float* dst, int64_t ld_dst,
                   rt_stream_t stream);
int rt_scatter_rows(const float* src, int64_t ld_src, const int64_t* idx, int32_t R, int32_t d, float* dst, int64_t ld_dst,
                    rt_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL (the reference delegates it to Lightning's DDP,
 * transformers/base.py:367-380: one all-reduce of the gradients per step).  One process per GPU; the current HIP device is
 * the rank's GPU.  rt_dp_unique_id: 128 opaque bytes made by ONE rank and handed to all ranks out of band;
 * rt_dp_init: collective, returns the communicator; rt_dp_allreduce: buf[0:n] <- sum over ranks (in place, fp32,
 * asynchronous on `stream`; DDP's 1/world is applied by rt_adam_step's grad_scale); rt_dp_broadcast: rank `root`'s buffer to
 * every rank (parameters and moments at the start of fit); rt_dp_finalize frees the communicator.  RCCL is resolved at run
 * time (dlopen: the copy torch already loaded, else RT_RCCL_LIB / librccl.so); RT_ERR_UNSUPPORTED if none is found.
 * ------------------------------------------------------------------------------------------------ */
int rt_dp_unique_id(void* out128);
int rt_dp_init(const void* uid128, int32_t rank, int32_t world, void** comm_out);
int rt_dp_allreduce(void* comm, float* buf, int64_t n, rt_stream_t stream);
int rt_dp_broadcast(void* comm, float* buf, int64_t n, int32_t root, rt_stream_t stream);
/* Sharded-optimiser exchange for table-dominated models (gradient of 1-10 GB per step): reduce-scatter the flat gradient (recv[0:n] <-
 * sum over ranks of send[rank*n : (rank+1)*n]), run rt_adam_step on the rank's slice of (p, m, v), all-gather the parameter slices
 * (recv[r*n : (r+1)*n] <- rank r's send[0:n]; send may alias the rank's own slice of recv).  n = padded total / world. */
int rt_dp_reduce_scatter(void* comm, const float* send, float* recv, int64_t n, rt_stream_t stream);
int rt_dp_allgather(void* comm, const float* send, float* recv, int64_t n, rt_stream_t stream);
int rt_dp_finalize(void* comm);
const char* rt_dp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RECTOOLS_HIP_H */
