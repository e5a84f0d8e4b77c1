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

}  // namespace

extern "C" {

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
