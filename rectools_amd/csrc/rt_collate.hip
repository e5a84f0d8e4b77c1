// K1 — batch collation on the device: the session store (CSR offsets + flat item / weight / timestamp
// arrays, SequenceDataset of data_preparator.py:39-99) stays resident in HBM and every training / recommend batch is
// cut out of it by one kernel — no per-step host gather, no H2D copy.  Pure index arithmetic: results are bit-identical
// to the reference's collate functions
//   SASRec  train      x = tail[:-1], y = tail[1:], yw = weights[1:] of the last L+1 items, left padded   sasrec.py:86-104
//           recommend  last L items left padded (with timestamps: last L+1, final row is the context)   sasrec.py:149-166
//   BERT4Rec train     last L items; Bernoulli(mask_prob) positions become targets, 80 % -> MASK, 10 % -> random item,
//                      10 % unchanged                                                                      bert4rec.py:109-153
//           recommend  last L-1 items + MASK                                                               bert4rec.py:182-193
// The BERT4Rec random draws come in as two device arrays (uniform probabilities, random item ids), so the kernel is a
// deterministic function of its inputs and can be checked against the host collate fed with the same draws.
#include "rt_common.h"

namespace {

struct CollateArgs {
  const long long* offsets; const long long* items; const float* weights; const long long* unix_ts;
  const long long* idx;     // [B] session indices
  int B, L, mode;
  long long* x; long long* y; float* yw; long long* ts_out;   // [B,L], [B,L], [B,L], [B,L+1]
  const float* probs; const long long* rand_ids; float mask_prob; long long mask_id;
};

enum { MODE_SASREC_TRAIN = 0, MODE_SASREC_RECO = 1, MODE_SASREC_RECO_TS = 2, MODE_BERT_TRAIN = 3, MODE_BERT_RECO = 4 };

// one thread per (row b, column c), c in [0, L] (column L exists only for the timestamp output)
__global__ __launch_bounds__(256) void collate_kernel(CollateArgs a) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int W = a.L + 1;
  if (t >= (long long)a.B * W) return;
  const int b = (int)(t / W), c = (int)(t % W);
  const long long u = a.idx[b];
  const long long lo = a.offsets[u], hi = a.offsets[u + 1];
  const long long cnt = hi - lo;
  const int L = a.L;
  const long long o = (long long)b * L + c;
  if (a.mode == MODE_SASREC_TRAIN || a.mode == MODE_SASREC_RECO_TS) {
    const int n = (int)(cnt < L + 1 ? cnt : L + 1);      // kept tail
    const long long base = hi - n;
    if (c < L) {
      const int p = c - (L - (n - 1));                    // tail element shown at input column c
      const bool ok = p >= 0 && p < n - 1;
      a.x[o] = ok ? a.items[base + p] : 0;
      if (a.mode == MODE_SASREC_TRAIN) {
        a.y[o] = ok ? a.items[base + p + 1] : 0;
        a.yw[o] = ok ? a.weights[base + p + 1] : 0.f;
      }
    }
    if (a.ts_out != nullptr) {                            // [B, L+1]: left pad repeats the first kept timestamp
      const int p = c - (W - n);
      a.ts_out[(long long)b * W + c] = n > 0 ? a.unix_ts[base + (p >= 0 ? p : 0)] : 0;
    }
  } else if (c < L) {
    if (a.mode == MODE_SASREC_RECO) {
      const int n = (int)(cnt < L ? cnt : L);
      const int p = c - (L - n);
      a.x[o] = p >= 0 ? a.items[hi - n + p] : 0;
    } else if (a.mode == MODE_BERT_RECO) {
      const int n = (int)(cnt < L - 1 ? cnt : L - 1);
      const int p = c - ((L - 1) - n);
      a.x[o] = c == L - 1 ? a.mask_id : (p >= 0 ? a.items[hi - n + p] : 0);
    } else {  // MODE_BERT_TRAIN
      const int n = (int)(cnt < L ? cnt : L);
      const int p = c - (L - n);
      long long xi = 0, yi = 0; float w = 0.f;
      if (p >= 0) {
        const long long it = a.items[hi - n + p];
        w = a.weights[hi - n + p];
        const float pr = a.probs[o];
        xi = it; yi = 0;
        if (pr < a.mask_prob) {
          yi = it;
          const float pj = pr / a.mask_prob;
          if (pj < 0.8f) xi = a.mask_id;
          else if (pj < 0.9f) xi = a.rand_ids[o];
        }
      }
      a.x[o] = xi; a.y[o] = yi; a.yw[o] = w;
    }
  }
}

// Packed (padding-free) batches, DESIGN.md §9.0: the rows of a batch are the REAL positions only.  Session b of the batch owns rows
// cu[b] .. cu[b+1]-1 (oldest first); cu is cut on the host from the store's offsets (the row count sizes every buffer of the step, so
// the host has to know it anyway — computing it there keeps the step free of device -> host synchronisation).  One thread per row:
// binary search of the row's session in cu, then the same index arithmetic as the padded collate.  Rows behind cu[B] (the tail up
// to the 128-row GEMM tile) get id 0 / target 0 / weight 0 / dist 0.
struct PackedArgs {
  const long long* offsets; const long long* items; const float* weights; const long long* idx; const long long* cu;
  int B, rows, train;
  long long* x; long long* y; float* yw; long long* dist;
};

__global__ __launch_bounds__(256) void collate_packed_kernel(PackedArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.rows) return;
  long long xi = 0, yi = 0, di = 0; float w = 0.f;
  if (r < a.cu[a.B]) {
    int lo = 0, hi = a.B;                 // largest b with cu[b] <= r
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.cu[mid] <= r) lo = mid; else hi = mid;
    }
    const long long c0 = a.cu[lo], n = a.cu[lo + 1] - c0, j = r - c0;
    const long long end = a.offsets[a.idx[lo] + 1];
    // train: the kept tail is n + 1 items, x = all but the last, y = all but the first (sasrec.py:86-104); recommend: the last n items
    const long long src = end - n - (a.train ? 1 : 0) + j;
    xi = a.items[src];
    di = n - 1 - j;
    if (a.train) { yi = a.items[src + 1]; w = a.weights[src + 1]; }
  }
  a.x[r] = xi; a.dist[r] = di;
  if (a.train) { a.y[r] = yi; a.yw[r] = w; }
}

// timestamps of a packed training batch: one thread per output entry; session b owns entries cu[b] + b .. cu[b+1] + b (its n rows' items
// and the target of the last row = the last n + 1 timestamps of the session)
__global__ __launch_bounds__(256) void collate_packed_ts_kernel(const long long* __restrict__ offsets, const long long* __restrict__ unix_ts,
                                                                const long long* __restrict__ idx, const long long* __restrict__ cu,
                                                                const long long* __restrict__ ctx, int B, long long n_out,
                                                                long long* __restrict__ ts_out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_out || e >= cu[B] + B) return;
  int lo = 0, hi = B;                     // largest b with cu[b] + b <= e
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] + mid <= e) lo = mid; else hi = mid;
  }
  const long long n = cu[lo + 1] - cu[lo], j = e - cu[lo] - lo;      // j in [0, n]
  const long long end = offsets[idx[lo] + 1];
  // train: the session's last n + 1 timestamps; recommend (ctx): its last n items' and the request's own (sasrec.py:149-166 with the
  // context row data_preparator.py:384-394 appends behind the history)
  ts_out[e] = ctx == nullptr ? unix_ts[end - (n + 1) + j] : (j < n ? unix_ts[end - n + j] : ctx[lo]);
}

// BERT4Rec on packed rows (bert4rec.py:109-153 / 182-193): train — session b shows its last n = cu[b+1] - cu[b] items, the masking draws
// are read at the PADDED position of the row (b, window - n + j), so that the packed batch masks exactly what `rt_collate` mode 3 masks
// when it is fed the same [B, window] draws; recommend — n - 1 history items and the MASK token as the last row.
struct PackedBertArgs {
  PackedArgs p;
  int window; const float* probs; const long long* rand_ids; const long long* draw_rows; float mask_prob; long long mask_id;
};

__global__ __launch_bounds__(256) void collate_packed_bert_kernel(PackedBertArgs q) {
  const PackedArgs& a = q.p;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.rows) return;
  long long xi = 0, yi = 0, di = 0; float w = 0.f;
  if (r < a.cu[a.B]) {
    int lo = 0, hi = a.B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.cu[mid] <= r) lo = mid; else hi = mid;
    }
    const long long c0 = a.cu[lo], n = a.cu[lo + 1] - c0, j = r - c0;
    const long long end = a.offsets[a.idx[lo] + 1];
    di = n - 1 - j;
    if (a.train) {
      const long long src = end - n + j;
      const long long it = a.items[src];
      w = a.weights[src];
      const long long o = (q.draw_rows != nullptr ? q.draw_rows[lo] : (long long)lo) * q.window + (q.window - n + j);
      const float pr = q.probs[o];
      xi = it;
      if (pr < q.mask_prob) {
        yi = it;
        const float pj = pr / q.mask_prob;
        if (pj < 0.8f) xi = q.mask_id;
        else if (pj < 0.9f) xi = q.rand_ids[o];
      }
    } else {
      xi = j == n - 1 ? q.mask_id : a.items[end - (n - 1) + j];
    }
  }
  a.x[r] = xi; a.dist[r] = di;
  if (a.train) { a.y[r] = yi; a.yw[r] = w; }
}

// a11 — CatalogUniformSampler.get_negatives (negative_sampler.py:58-73): n ids uniform in [low, high), no rejection of
// positives.  Counter-based (Philox4x32-10): element e is word (e & 3) of philox(seed, subsequence = e >> 2, offset), so a
// batch is a pure function of (seed, offset) — reproducible whatever the launch geometry — and `offset` (the step counter)
// gives every batch a fresh stream.  A 32-bit word r maps to low + floor(r * range / 2^32) (bias <= range / 2^32, against
// the reference's `random() % range`, which is biased the same way).
__global__ __launch_bounds__(256) void sample_negatives_kernel(long long low, unsigned range, long long n4, long long n,
                                                                unsigned long long seed, unsigned long long offset,
                                                                long long* __restrict__ out) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    const uint4 r = philox4x32(seed, (unsigned long long)q, offset);
    long long v[4];
    v[0] = low + (long long)__umulhi(r.x, range); v[1] = low + (long long)__umulhi(r.y, range);
    v[2] = low + (long long)__umulhi(r.z, range); v[3] = low + (long long)__umulhi(r.w, range);
    const long long e = 4 * q;
    if (e + 3 < n) {   // out is 16-byte aligned (checked by the entry point) and e = 4q: both stores are aligned
      *reinterpret_cast<longlong2*>(out + e) = make_longlong2(v[0], v[1]);
      *reinterpret_cast<longlong2*>(out + e + 2) = make_longlong2(v[2], v[3]);
    } else {
      for (int j = 0; j < 4 && e + j < n; ++j) out[e + j] = v[j];
    }
  }
}

}  // namespace

extern "C" {

int rt_sample_negatives(int64_t low, int64_t high, int64_t n, uint64_t seed, uint64_t offset, int64_t* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (n < 0 || high <= low || high - low > 0xFFFFFFFFLL) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (out == nullptr || ((uintptr_t)out & 15) != 0) return RT_ERR_INVALID_ARG;
  const long long n4 = (n + 3) / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 16 * rt_num_cus()) blocks = 16 * rt_num_cus();
  sample_negatives_kernel<<<(int)blocks, 256, 0, stream>>>(low, (unsigned)(high - low), n4, n, seed, offset,
                                                            reinterpret_cast<long long*>(out));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Packed SASRec batch (no padding rows): cu_seqlens [B+1] (device; cu[b+1] - cu[b] = min(session length - train, window) rows of
// session idx[b], cut by the caller from the store's offsets), rows = the row count of the outputs (>= cu[B]; the tail is zero
// filled).  train = 1: x / y / yw as sasrec.py:86-104 (y, yw, weights required); train = 0: x = the last items (recommend).
// dist [rows] = distance of a row from its session's end (the index of its positional row, net_blocks.py:388-399).
int rt_collate_packed(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* idx, const int64_t* cu_seqlens,
                      int32_t B, int32_t rows, int32_t train, int64_t* x, int64_t* y, float* yw, int64_t* dist, hipStream_t stream) {
  (void)hipGetLastError();
  if (B < 0 || rows < 0) return RT_ERR_INVALID_ARG;
  if (rows == 0) return RT_OK;
  if (offsets == nullptr || items == nullptr || idx == nullptr || cu_seqlens == nullptr || x == nullptr || dist == nullptr)
    return RT_ERR_INVALID_ARG;
  if (train && (y == nullptr || yw == nullptr || weights == nullptr)) return RT_ERR_INVALID_ARG;
  PackedArgs a{};
  a.offsets = reinterpret_cast<const long long*>(offsets); a.items = reinterpret_cast<const long long*>(items); a.weights = weights;
  a.idx = reinterpret_cast<const long long*>(idx); a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.rows = rows;
  a.train = train ? 1 : 0; a.x = reinterpret_cast<long long*>(x); a.y = reinterpret_cast<long long*>(y); a.yw = yw;
  a.dist = reinterpret_cast<long long*>(dist);
  collate_packed_kernel<<<(rows + 255) / 256, 256, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Timestamps of a packed batch: n_out = cu[B] + B entries (the caller knows cu[B] on the host), session b's n + 1 at cu[b] + b.
// ctx == NULL (training): the session's last n + 1 timestamps (it holds them: n = min(length - 1, window)).  ctx [B] (recommend with a
// context): the last n items' timestamps followed by ctx[b], the time of the request.
int rt_collate_packed_ts(const int64_t* offsets, const int64_t* unix_ts, const int64_t* idx, const int64_t* cu_seqlens, const int64_t* ctx,
                         int32_t B, int64_t n_out, int64_t* ts_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (B < 0 || n_out < 0) return RT_ERR_INVALID_ARG;
  if (n_out == 0) return RT_OK;
  if (offsets == nullptr || unix_ts == nullptr || idx == nullptr || cu_seqlens == nullptr || ts_out == nullptr) return RT_ERR_INVALID_ARG;
  collate_packed_ts_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, stream>>>(
      reinterpret_cast<const long long*>(offsets), reinterpret_cast<const long long*>(unix_ts), reinterpret_cast<const long long*>(idx),
      reinterpret_cast<const long long*>(cu_seqlens), reinterpret_cast<const long long*>(ctx), B, n_out, reinterpret_cast<long long*>(ts_out));
  RT_CHECK_LAUNCH();
  return RT_OK;
}
// Packed BERT4Rec batch.  train = 1: cu[b+1] - cu[b] = min(session length, window) rows; probs / rand_ids [B, window] are the draws of
// `rt_collate` mode 3 (read at the row's padded position), y = the item where the position was picked, else 0.  draw_rows [B] or NULL:
// the row of probs / rand_ids session b reads (a loop that re-orders the sessions of a batch keeps every session on the draws of its
// original slot); NULL = b.  train = 0: min(session length, window - 1) + 1 rows, the last one the MASK token (bert4rec.py:182-193).
int rt_collate_packed_bert(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* idx, const int64_t* cu_seqlens,
                           int32_t B, int32_t rows, int32_t window, int32_t train, const float* probs, const int64_t* rand_ids,
                           const int64_t* draw_rows, float mask_prob, int64_t mask_id, int64_t* x, int64_t* y, float* yw, int64_t* dist, hipStream_t stream) {
  (void)hipGetLastError();
  if (B < 0 || rows < 0 || window <= 0) return RT_ERR_INVALID_ARG;
  if (rows == 0) return RT_OK;
  if (offsets == nullptr || items == nullptr || idx == nullptr || cu_seqlens == nullptr || x == nullptr || dist == nullptr)
    return RT_ERR_INVALID_ARG;
  if (train && (y == nullptr || yw == nullptr || weights == nullptr || probs == nullptr || rand_ids == nullptr)) return RT_ERR_INVALID_ARG;
  PackedBertArgs q{};
  PackedArgs& a = q.p;
  a.offsets = reinterpret_cast<const long long*>(offsets); a.items = reinterpret_cast<const long long*>(items); a.weights = weights;
  a.idx = reinterpret_cast<const long long*>(idx); a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.rows = rows;
  a.train = train ? 1 : 0; a.x = reinterpret_cast<long long*>(x); a.y = reinterpret_cast<long long*>(y); a.yw = yw;
  a.dist = reinterpret_cast<long long*>(dist);
  q.window = window; q.probs = probs; q.rand_ids = reinterpret_cast<const long long*>(rand_ids);
  q.draw_rows = reinterpret_cast<const long long*>(draw_rows); q.mask_prob = mask_prob; q.mask_id = mask_id;
  collate_packed_bert_kernel<<<(rows + 255) / 256, 256, 0, stream>>>(q);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// mode: 0 SASRec train, 1 SASRec recommend, 2 SASRec recommend with timestamps (last row = context), 3 BERT4Rec train,
// 4 BERT4Rec recommend.  Unused outputs / inputs may be NULL (y, yw outside the train modes; ts_out / unix_ts without
// timestamps; probs / rand_ids outside mode 3).  Sessions must be non-empty in modes 0 and 2.
int rt_collate(const int64_t* offsets, const int64_t* items, const float* weights, const int64_t* unix_ts, const int64_t* idx,
               int32_t B, int32_t L, int32_t mode, const float* probs, const int64_t* rand_ids, float mask_prob,
               int64_t mask_id, int64_t* x, int64_t* y, float* yw, int64_t* ts_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (B < 0 || L <= 0 || mode < MODE_SASREC_TRAIN || mode > MODE_BERT_RECO) return RT_ERR_INVALID_ARG;
  if (B == 0) return RT_OK;
  if (offsets == nullptr || items == nullptr || idx == nullptr || x == nullptr) return RT_ERR_INVALID_ARG;
  if ((mode == MODE_SASREC_TRAIN || mode == MODE_BERT_TRAIN) && (y == nullptr || yw == nullptr || weights == nullptr)) return RT_ERR_INVALID_ARG;
  if (mode == MODE_BERT_TRAIN && (probs == nullptr || rand_ids == nullptr)) return RT_ERR_INVALID_ARG;
  if (ts_out != nullptr && (unix_ts == nullptr || (mode != MODE_SASREC_TRAIN && mode != MODE_SASREC_RECO_TS))) return RT_ERR_INVALID_ARG;
  CollateArgs a{};
  a.offsets = reinterpret_cast<const long long*>(offsets); a.items = reinterpret_cast<const long long*>(items);
  a.weights = weights; a.unix_ts = reinterpret_cast<const long long*>(unix_ts); a.idx = reinterpret_cast<const long long*>(idx);
  a.B = B; a.L = L; a.mode = mode; a.x = reinterpret_cast<long long*>(x); a.y = reinterpret_cast<long long*>(y); a.yw = yw;
  a.ts_out = reinterpret_cast<long long*>(ts_out); a.probs = probs; a.rand_ids = reinterpret_cast<const long long*>(rand_ids);
  a.mask_prob = mask_prob; a.mask_id = mask_id;
  const long long n = (long long)B * (L + 1);
  collate_kernel<<<(int)((n + 255) / 256), 256, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
