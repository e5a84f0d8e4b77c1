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
 *  k            1 <= k <= n_candidates (caller clamps, as rank_torch.py:148 does)
 *  out_ids      [n_users, k] int64 item ids (whitelist-mapped), best first; out_scores [n_users, k];
 *               out_counts [n_users] number of valid leading entries (< k only when the filter leaves
 *               fewer than k candidates — the reference drops -inf rows, rank_torch.py:167-171)
 *  Ordering: score descending (ascending distance for EUCLIDEAN); exact ties -> lower position first.
 *  users_per_pass: 32, 64 or 128 users share one pass over the catalog (0 = library default).
 *  d, strides must be multiples of 4 floats and base pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
size_t rt_topk_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t users_per_pass);

int rt_topk_score(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users,
                  const float* items, int64_t item_stride, const int64_t* whitelist, int64_t n_candidates,
                  int64_t candidate_id_offset, int32_t d, int32_t distance, int32_t k,
                  const int64_t* filt_indptr, const int32_t* filt_indices,
                  int64_t* out_ids, float* out_scores, int32_t* out_counts,
                  void* workspace, size_t workspace_bytes, int32_t users_per_pass, rt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECTOOLS_HIP_H */
