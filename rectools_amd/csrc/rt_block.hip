// Native executor of one PACKED SASRec block (sasrec.py:186-231, :300): the whole launch sequence of the block's forward, of its
// backward and of its inference form behind ONE C call each.
//
// Why: with the padding rows gone and the attention on the bf16 pipe, a C2 training step holds ~1.7 ms of GPU work in ~60 kernel
// launches — and Python needed 2.0-2.3 ms to issue them (ctypes call + torch.empty per buffer + stream / event objects for the
// weight-gradient side stream): `host_issue_ms_per_step` equalled `ms_per_step` in bench.py, the step was launch-bound on the HOST.
// Here the sequence is issued by compiled code (3-4 us per launch); Python hands over three buffers per block (saved activations,
// scratch, flat parameter gradient) and the parameter pointers.  Same kernels, same order, same arithmetic as
// `ops._SASRecLayerPacked` (kept as the cross-check: tests/test_packed_gpu.py::test_native_block_equals_the_python_block).
//
// Weight gradients leave the critical path on a library-owned side stream (fork: event on the caller's stream; join: rt_side_join,
// called once before the optimiser step).  The caller keeps the three buffers and the block input alive until the join.
// Optional HIP-event instrumentation (rt_timing_*) brackets every internal launch for bench.py's roofline pass.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rt_common.h"

extern "C" {
struct rt_gemm_problem {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc;
  const float* bias; const float* R; int64_t ldr; int32_t M, N, K, relu;
};
size_t rt_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t split_k);
int rt_gemm(const float* A, int64_t lda, int32_t a_kc, const float* B, int64_t ldb, int32_t b_kc, float* C, int64_t ldc, const float* bias,
            const float* R, int64_t ldr, float* a_rowsum, int32_t M, int32_t N, int32_t K, int32_t relu, int32_t split_k, void* workspace,
            size_t workspace_bytes, hipStream_t stream);
int rt_gemm_grouped(const rt_gemm_problem* problems, int32_t n, int32_t a_kc, int32_t b_kc, hipStream_t stream);
int rt_colsum(const float* X, int64_t ld, int32_t M, int32_t N, float* out, hipStream_t stream);
int rt_layernorm_fwd(const float* x, const float* w, const float* b, float eps, int32_t M, int32_t d, float* y, float* mean, float* rstd,
                     hipStream_t stream);
size_t rt_layernorm_bwd_workspace_bytes(int32_t M, int32_t d);
int rt_mha_varlen_last_x_fwd(const float* qk, const float* x, int64_t ldx, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t d,
                             int32_t max_len, int32_t window, int32_t pad_keys, int64_t prefix_row, float* xbar, hipStream_t stream);
int rt_mha_last_x_expand(const float* W, int32_t d, int32_t H, float* E, hipStream_t stream);
int rt_layernorm_bwd_rows(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, const float* res,
                          const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, float* dx, void* workspace,
                          size_t workspace_bytes, hipStream_t stream);
int rt_layernorm_bwd_combine(const void* workspace, size_t workspace_bytes, int32_t M, int32_t d, float* dw, float* db, hipStream_t stream);
int rt_layernorm_bwd_fused(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, const float* res,
                           const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, float* dx, float* dw, float* db,
                           void* workspace, size_t workspace_bytes, hipStream_t stream);
int rt_act_dropout_fwd(const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id, int64_t n, const float* residual,
                       float* y, hipStream_t stream);
int rt_act_dropout_bwd(const float* dy, const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id, int64_t n, float* dz,
                       hipStream_t stream);
int rt_mha_varlen_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                      const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float* o,
                      int64_t ldo, hipStream_t stream);
int rt_mha_varlen_train_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            hipStream_t stream);
int rt_mha_varlen_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                      const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, const float* bk, const float* bv,
                      int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* dq,
                      int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta, float* dbv_part, hipStream_t stream);
int rt_mha_varlen_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd, int32_t max_len,
                           int32_t window, float* o, int64_t ldo, hipStream_t stream);
int rt_mha_varlen_bidir_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                            int32_t B, int32_t H, int32_t hd, int32_t max_len, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            hipStream_t stream);
int rt_mha_varlen_bidir_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                            const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, float p_drop, uint64_t seed, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                            int64_t lddv, float* delta, hipStream_t stream);
int rt_gather_rows(const float* src, int64_t ld_src, const int64_t* idx, int32_t R, int32_t d, float* dst, int64_t ld_dst,
                   hipStream_t stream);
struct rt_gemm_wp_problem {
  const float* A; int64_t lda;
  const uint16_t* W; int64_t plane_stride, ldw;
  float* C; int64_t ldc;
  const float* bias; const float* R; int64_t ldr;
  int32_t M, N, K, relu;
};
int rt_gemm_wp(const rt_gemm_wp_problem* problems, int32_t n, int32_t w_tr, hipStream_t stream);
int rt_ffn_fused_supported(int32_t M, int32_t d, int32_t dff);
int rt_ffn_fused_fwd(const float* y, const float* ln_w, const float* ln_b, float eps, float* f, float* mean, float* rstd,
                     const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, const float* b1, const float* b2, float* hdrop,
                     float* out, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_h, uint64_t sid_h, uint64_t seed_o, uint64_t sid_o,
                     hipStream_t stream);
int rt_ffn_fused_bwd(const float* g_out, const float* hdrop, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride,
                     float* g_o, float* g_h, float* g_f, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o, uint64_t sid_o,
                     hipStream_t stream);
int rt_block_tail_fwd(const float* attn, const float* q, const uint16_t* wo_planes, const float* bo, const float* ln_w, const float* ln_b,
                      float eps, float* y, float* f, float* mean, float* rstd, const uint16_t* w1_planes, const uint16_t* w2_planes,
                      int64_t plane_stride, const float* b1, const float* b2, float* hdrop, float* out, int32_t M, int32_t d, int32_t dff, float p,
                      uint64_t seed_h, uint64_t sid_h, uint64_t seed_o, uint64_t sid_o, int32_t training, hipStream_t stream);
int rt_block_tail_bwd(const float* g_out, const float* hdrop, const float* y, const float* mean, const float* rstd, const float* ln_w,
                      const uint16_t* wo_planes, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, float* g_o,
                      float* g_h, float* g_y, float* g_A, float* ln_partial, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o,
                      uint64_t sid_o, hipStream_t stream);
int rt_layernorm_bwd_reduce(const float* partial, int32_t blocks, int32_t d, float* dw, float* db, hipStream_t stream);
struct rt_wgrad_problem { const float* dy; int64_t ldy; const float* in; int64_t ldin; float* dw; float* db; int32_t n_out, n_in; };
size_t rt_wgrad_grouped_workspace_bytes(const rt_wgrad_problem* problems, int32_t n, int32_t rows, int32_t splits);
int rt_wgrad_grouped(const rt_wgrad_problem* problems, int32_t n, int32_t rows, int32_t splits, void* workspace, size_t workspace_bytes,
                     hipStream_t stream);
}

namespace {

// ---- HIP-event instrumentation of the internal launches ---------------------------------------------------------------------------
struct TimeRec { int id; long long m, n, k; hipEvent_t e0, e1; };
bool g_timing = false;
bool g_single_stream = false;     // instrumentation mode 2: weight gradients on the caller's stream
std::vector<TimeRec> g_recs;

struct Timed {   // RAII bracket around one internal launch (no-op unless rt_timing_enable(1))
  hipStream_t s; bool on; TimeRec r;
  Timed(int id, long long m, long long n, long long k, hipStream_t stream) : s(stream), on(g_timing) {
    if (!on) return;
    r.id = id; r.m = m; r.n = n; r.k = k;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, s);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(r.e1, s);
    g_recs.push_back(r);
  }
};
enum { T_GEMM = 0, T_GEMM_GROUPED = 1, T_LN_FWD = 2, T_LN_BWD = 3, T_DROP_FWD = 4, T_DROP_BWD = 5, T_ATTN_FWD = 6, T_ATTN_BWD = 7,
       T_ATTN_LAST = 8, T_MISC = 9, T_ATTN_BIDIR_FWD = 10, T_ATTN_BIDIR_BWD = 11, T_FFN_FWD = 12, T_FFN_BWD = 13 };

// ---- the weight-gradient side stream (one per device, owned by the library) -------------------------------------------------------
struct Side { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr, mark = nullptr, reach = nullptr; bool dirty = false, marked = false; };
Side g_side[16];
bool side_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("RT_SIDE_STREAM"); on = (e != nullptr && e[0] == '0') ? 0 : 1; }
  return on == 1 && !g_single_stream;
}
Side* side_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  Side& s = g_side[dev];
  if (s.stream == nullptr) {
    // LOW priority: what runs here is needed by the optimiser only.  At equal priority the dispatcher keeps refilling the CUs' wave slots
    // with the side stream's small-footprint workgroups (the loss's row reductions: 8 waves per SIMD) and a main-stream kernel that needs
    // a whole CU's LDS waits for a slot: v2_bwd_dq 227 us instead of 67 beside sampled_bwd_rows (profiles/r4_timeline_train.txt).
    // RT_SIDE_PRIORITY=normal keeps the default priority.
    const char* e = getenv("RT_SIDE_PRIORITY");
    int least = 0, greatest = 0;
    const bool low = (e == nullptr || e[0] != 'n') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
    const hipError_t rc = low ? hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (rc != hipSuccess) return nullptr;
    (void)hipEventCreateWithFlags(&s.fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&s.join, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&s.reach, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&s.mark, hipEventDisableTiming);
  }
  return &s;
}

#define RT_TRY(call)                 \
  do {                               \
    const int rc__ = (call);         \
    if (rc__ != RT_OK) return rc__;  \
  } while (0)

inline size_t al(size_t floats) { return (floats + 63) & ~(size_t)63; }   // 256-byte aligned regions

}  // namespace

extern "C" {

// ---- instrumentation ---------------------------------------------------------------------------------------------------------------
// mode 0: off; 1: record an event pair around every internal launch of the block executor; 2: the same with the weight gradients on the
// caller's stream (undisturbed kernel durations).  rt_timing_collect synchronises the device, copies the records out
// (ids: 0 gemm, 1 gemm_grouped, 2 layernorm_fwd, 3 layernorm_bwd, 4 act_dropout_fwd, 5 act_dropout_bwd, 6 mha_varlen_fwd,
// 7 mha_varlen_bwd, 8 mha_varlen_last_fwd, 9 misc, 10 / 11 mha_varlen_bidir_fwd / _bwd, 12 / 13 ffn_fused_fwd / _bwd (tag = (M, 2 dff, d): both
// products); tags [n][3] = the GEMM's M, N, K or 0) and clears them.
int rt_timing_enable(int32_t mode) {
  g_timing = mode != 0;
  g_single_stream = mode == 2;
  return RT_OK;
}
int rt_timing_collect(int32_t* ids, float* ms, int64_t* tags, int32_t max_records, int32_t* n_out) {
  RT_CHECK_HIP(hipDeviceSynchronize());
  int n = 0;
  for (auto& r : g_recs) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    if (n < max_records && ids != nullptr) { ids[n] = r.id; ms[n] = t; tags[3 * n] = r.m; tags[3 * n + 1] = r.n; tags[3 * n + 2] = r.k; ++n; }
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  if (n_out != nullptr) *n_out = n;
  return RT_OK;
}

// The caller's stream waits for every weight-gradient product issued on the side stream since the last join (no-op when none).
int rt_side_join(hipStream_t stream) {
  Side* s = side_of_current_device();
  if (s == nullptr || !s->dirty) return RT_OK;
  RT_CHECK_HIP(hipEventRecord(s->join, s->stream));
  RT_CHECK_HIP(hipStreamWaitEvent(stream, s->join, 0));
  s->dirty = false; s->marked = false;
  return RT_OK;
}

// `stream` waits for everything issued so far on the side stream, and the side stream's bookkeeping stays as it is (the step's
// rt_side_join still joins and clears): the early gradient exchange of a data-parallel step (lightning.FlatAdam.begin_early_exchange)
// reads the block weights' gradients from its own stream while the caller's stream runs on.
int rt_side_reach(hipStream_t stream) {
  Side* s = side_of_current_device();
  if (s == nullptr || !s->dirty) return RT_OK;
  RT_CHECK_HIP(hipEventRecord(s->reach, s->stream));
  RT_CHECK_HIP(hipStreamWaitEvent(stream, s->reach, 0));
  return RT_OK;
}

// A point on the side stream that a later launch on another stream can wait for WITHOUT waiting for what the side stream is given
// afterwards (rt_side_join waits for everything issued so far): rt_side_mark records it, rt_side_wait_mark makes `stream` wait for the
// latest one (no-op when none was recorded since the last join).  The embedding backward uses the pair: it needs the loss's table
// gradient (side stream, early in the backward pass), not the weight gradients queued behind it.
int rt_side_mark(void) {
  Side* s = side_of_current_device();
  if (s == nullptr || s->stream == nullptr) return RT_OK;
  RT_CHECK_HIP(hipEventRecord(s->mark, s->stream));
  s->marked = true;
  return RT_OK;
}
int rt_side_wait_mark(hipStream_t stream) {
  Side* s = side_of_current_device();
  if (s == nullptr || !s->marked) return RT_OK;
  RT_CHECK_HIP(hipStreamWaitEvent(stream, s->mark, 0));
  return RT_OK;
}

// The library's side stream itself (created on first use; NULL through *side_out when RT_SIDE_STREAM=0): a host binding that has side work of
// its own (the next batch's collate, Python-issued weight gradients) wraps THIS stream instead of creating another — a process that owns
// more streams than the device has hardware queues finds two of them sharing one, and which two is decided at run time (round 6: the HSTU
// loop at 20.3 k seqs/s in one default bench line and 24.1 k in the next, its weight gradients serialised with the main stream in the first).
int rt_side_stream(void** side_out) {
  if (side_out == nullptr) return RT_ERR_INVALID_ARG;
  *side_out = nullptr;
  Side* s = side_enabled() ? side_of_current_device() : nullptr;
  if (s != nullptr) *side_out = s->stream;
  return RT_OK;
}

// The side stream as a service to the caller's own launches (the loss's table-gradient half, the embedding backward: results only the
// optimiser reads): rt_side_fork makes the side stream wait for everything issued so far on `stream` and returns its handle through
// *side_out (NULL when the side stream is disabled: launch on `stream` then); work issued on it is joined by rt_side_join.
int rt_side_fork(hipStream_t stream, void** side_out) {
  if (side_out == nullptr) return RT_ERR_INVALID_ARG;
  *side_out = nullptr;
  Side* s = side_enabled() ? side_of_current_device() : nullptr;
  if (s == nullptr) return RT_OK;
  RT_CHECK_HIP(hipEventRecord(s->fork, stream));
  RT_CHECK_HIP(hipStreamWaitEvent(s->stream, s->fork, 0));
  s->dirty = true;
  *side_out = s->stream;
  return RT_OK;
}

// One packed SASRec block.  rows: the row count of x / out (multiple of 128 for the exact-tile GEMM path; rows_real of them belong to
// sessions, the rest is the unused tail); cu_seqlens [B+1]; window = session_max_len; pad_keys: the block runs without key-padding
// masks, the window's pad keys are the virtual key of the attention.  Parameters as in the reference's state_dict (Appendix B).
struct rt_sasrec_block {
  int32_t rows, rows_real, B, H, d, dff, window, pad_keys;
  float p_drop, eps1, eps2;
  uint64_t seed_attn, seed_h, sid_h, seed_o, sid_o;     // dropout streams (ops.RNG)
  const int64_t* cu;
  const float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
  // optional: the bf16 planes of the four weight matrices (rt_split_planes over the stack's parameter range; same element offsets as
  // the fp32 weights, plane p at + p * wp_stride).  NULL: every product takes rt_gemm (both operands split in registers).
  const uint16_t *in_wp, *out_wp, *w1_wp, *w2_wp;
  int64_t wp_stride;
};

// floats of the activation record the forward keeps for the backward
size_t rt_sasrec_block_saved_floats(int32_t rows, int32_t d, int32_t dff, int32_t H, int32_t with_dropout) {
  const size_t M = (size_t)rows;
  return al(M * d) * 6 /* q Q A y f + spare */ + al(M * 2 * d) /* KV */ + al(M * dff) * (with_dropout ? 2 : 1) /* h, hdrop */ +
         al(M * H) /* lse */ + 4 * al(M) /* mean1 rstd1 mean2 rstd2 */;
}
// bytes of the backward's scratch (data gradients, attention workspace, LayerNorm partials, the split-K slabs of the side stream)
size_t rt_sasrec_block_bwd_scratch_bytes(int32_t rows, int32_t B, int32_t d, int32_t dff, int32_t H, int32_t wgrad_splits) {
  const size_t M = (size_t)rows;
  size_t fl = al(M * d) * 7 /* g_o g_f g_y g_A gQ g_q g_kv */ + al(M * dff) * 2 /* g_hd g_h */ + al(M * 2 * d) /* gKV */ + al(M * H) /* delta */ +
              al((size_t)B * d) /* part */;
  size_t by = fl * 4 + 2 * ((rt_layernorm_bwd_workspace_bytes(rows, d) + 255) & ~(size_t)255);
  size_t sk = 0;
  const int spl = wgrad_splits > 1 ? wgrad_splits : 1;
  const size_t cands[5] = {rt_gemm_workspace_bytes(d, dff, rows, spl), rt_gemm_workspace_bytes(dff, d, rows, spl), rt_gemm_workspace_bytes(d, d, rows, spl),
                           rt_gemm_workspace_bytes(2 * d, d, rows, spl), 0};
  for (size_t c : cands) sk = c > sk ? c : sk;
  {   // the grouped weight-gradient launches (rt_wgrad_grouped): {W2, W1, Wo} behind the block's tail, {Wq, Wkv} behind the attention
    const rt_wgrad_problem ga[3] = {{nullptr, 0, nullptr, 0, nullptr, nullptr, d, dff}, {nullptr, 0, nullptr, 0, nullptr, nullptr, dff, d},
                                    {nullptr, 0, nullptr, 0, nullptr, nullptr, d, d}};
    const rt_wgrad_problem gb[2] = {{nullptr, 0, nullptr, 0, nullptr, nullptr, d, d}, {nullptr, 0, nullptr, 0, nullptr, nullptr, 2 * d, d}};
    const size_t a = rt_wgrad_grouped_workspace_bytes(ga, 3, rows, spl), b2 = rt_wgrad_grouped_workspace_bytes(gb, 2, rows, spl);
    sk = a > sk ? a : sk;
    sk = b2 > sk ? b2 : sk;
  }
  return by + ((sk + 255) & ~(size_t)255) + 256;
}

namespace {
// y = A W'^T-or-W' (+ bias) (+ R) (relu) through the pre-split weight planes when the block carries them and the shape is an exact tile
// grid; RT_ERR_UNSUPPORTED -> the caller's rt_gemm call.  w_tr = 0: forward (W [N,K]); 1: data gradient (W [K,N], ldw = N).
int wp_one(const float* A, int lda, const uint16_t* W, int64_t stride, int ldw, int w_tr, float* C, int ldc, const float* bias, const float* R,
           int ldr, int M, int N, int K, int relu, hipStream_t s) {
  if (W == nullptr) return RT_ERR_UNSUPPORTED;
  rt_gemm_wp_problem pr{A, lda, W, stride, ldw, C, ldc, bias, R, ldr, M, N, K, relu};
  return rt_gemm_wp(&pr, 1, w_tr, s);
}
// The feed-forward half as one launch per direction (rt_ffn.hip) — a pure function of the block's shape and planes, so that the forward and
// the backward of a step always agree (the fused forward keeps hdrop only: the unfused backward would read an h nobody wrote).
// RT_FFN_FUSED=0 keeps the five-launch sequence (the cross-check of tests/test_packed_gpu.py).
// RT_FFN_FUSED: 0 = the separate launches; 1 = the feed-forward half fused (LN2 .. skip: two products); 2 (default) = the block's whole
// tail behind the attention fused (out-projection .. skip, and in the backward down to the attention's output gradient: three products).
int ffn_mode() {
  static const int m = [] { const char* e = getenv("RT_FFN_FUSED"); return e != nullptr && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 2; }();
  return m;
}
bool ffn_fused(const rt_sasrec_block& b) {
  return ffn_mode() >= 1 && b.w1_wp != nullptr && b.w2_wp != nullptr && rt_ffn_fused_supported(b.rows, b.d, b.dff) == 1;
}
bool tail_fused(const rt_sasrec_block& b) { return ffn_mode() == 2 && ffn_fused(b) && b.out_wp != nullptr; }
struct SavedView {
  float *q, *Q, *A, *y, *f, *KV, *h, *hdrop, *lse, *mean1, *rstd1, *mean2, *rstd2;
};
SavedView carve_saved(const rt_sasrec_block& b, float* base) {
  const size_t M = (size_t)b.rows;
  SavedView v;
  float* p = base;
  v.q = p; p += al(M * b.d); v.Q = p; p += al(M * b.d); v.A = p; p += al(M * b.d); v.y = p; p += al(M * b.d); v.f = p; p += al(M * b.d);
  p += al(M * b.d);   // spare
  v.KV = p; p += al(M * 2 * b.d);
  v.h = p; p += al(M * b.dff);
  if (b.p_drop > 0.f) { v.hdrop = p; p += al(M * b.dff); } else v.hdrop = v.h;
  v.lse = p; p += al(M * b.H);
  v.mean1 = p; p += al(M); v.rstd1 = p; p += al(M); v.mean2 = p; p += al(M); v.rstd2 = p; p += al(M);
  return v;
}
int zero_tail(float* base, const rt_sasrec_block& b, int cols, hipStream_t s) {   // rows behind the sessions must read as finite zeros
  if (b.rows_real >= b.rows) return RT_OK;
  RT_CHECK_HIP(hipMemsetAsync(base + (size_t)b.rows_real * cols, 0, (size_t)(b.rows - b.rows_real) * cols * sizeof(float), s));
  return RT_OK;
}
}  // namespace

// Training forward: out [rows, d] = block(x) (no masks: there are no pad rows); saved: rt_sasrec_block_saved_floats floats.
int rt_sasrec_block_packed_fwd(const rt_sasrec_block* blk, const float* x, float* saved, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (blk == nullptr || x == nullptr || saved == nullptr || out == nullptr) return RT_ERR_INVALID_ARG;
  const rt_sasrec_block& b = *blk;
  const int M = b.rows, d = b.d, dff = b.dff, hd = d / b.H;
  if (M <= 0 || d <= 0 || b.H <= 0 || d % b.H != 0 || b.cu == nullptr) return RT_ERR_INVALID_ARG;
  const SavedView v = carve_saved(b, saved);
  { Timed t(T_LN_FWD, 0, 0, 0, stream); RT_TRY(rt_layernorm_fwd(x, b.ln1_w, b.ln1_b, b.eps1, M, d, v.q, v.mean1, v.rstd1, stream)); }
  {
    Timed t(T_GEMM_GROUPED, (long long)M * d * d + (long long)M * 2 * d * d, 1, 1, stream);
    int rc = RT_ERR_UNSUPPORTED;
    if (b.in_wp != nullptr) {
      rt_gemm_wp_problem wp[2] = {{v.q, d, b.in_wp, b.wp_stride, d, v.Q, d, b.in_b, nullptr, 0, M, d, d, 0},
                                  {x, d, b.in_wp + (size_t)d * d, b.wp_stride, d, v.KV, 2 * d, b.in_b + d, nullptr, 0, M, 2 * d, d, 0}};
      rc = rt_gemm_wp(wp, 2, 0, stream);
    }
    if (rc == RT_ERR_UNSUPPORTED) {
      rt_gemm_problem pr[2] = {{v.q, d, b.in_w, d, v.Q, d, b.in_b, nullptr, 0, M, d, d, 0},                                // Q = LN1(x) Wq^T + bq
                               {x, d, b.in_w + (size_t)d * d, d, v.KV, 2 * d, b.in_b + d, nullptr, 0, M, 2 * d, d, 0}};    // K | V = x Wkv^T + bkv
      rc = rt_gemm_grouped(pr, 2, 1, 1, stream);
    }
    RT_TRY(rc);
  }
  RT_TRY(zero_tail(v.A, b, d, stream));
  const float* bk = b.pad_keys ? b.in_b + d : nullptr;
  const float* bv = b.pad_keys ? b.in_b + 2 * d : nullptr;
  { Timed t(T_ATTN_FWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_train_fwd(v.Q, d, v.KV, 2 * d, v.KV + d, 2 * d, b.cu, bk, bv, b.B, b.H, hd, b.window, b.window, b.p_drop, b.seed_attn,
                                   v.A, d, v.lse, stream)); }
  if (tail_fused(b)) {   // out-projection + skip -> LN2 -> W1 -> relu -> dropout -> W2 -> dropout -> + f in one launch
    Timed t(T_FFN_FWD, M, 2 * dff + d, d, stream);
    RT_TRY(rt_block_tail_fwd(v.A, v.q, b.out_wp, b.out_b, b.ln2_w, b.ln2_b, b.eps2, v.y, v.f, v.mean2, v.rstd2, b.w1_wp, b.w2_wp, b.wp_stride, b.b1,
                             b.b2, v.hdrop, out, M, d, dff, b.p_drop, b.seed_h, b.sid_h, b.seed_o, b.sid_o, 1, stream));
    return RT_OK;
  }
  { Timed t(T_GEMM, M, d, d, stream);                                                                                   // y = q + Wo A + bo
    int rc = wp_one(v.A, d, b.out_wp, b.wp_stride, d, 0, v.y, d, b.out_b, v.q, d, M, d, d, 0, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(v.A, d, 1, b.out_w, d, 1, v.y, d, b.out_b, v.q, d, nullptr, M, d, d, 0, 1, nullptr, 0, stream);
    RT_TRY(rc); }
  if (ffn_fused(b)) {   // LN2 -> W1 -> relu -> dropout -> W2 -> dropout -> + f in one launch
    Timed t(T_FFN_FWD, M, 2 * dff, d, stream);
    RT_TRY(rt_ffn_fused_fwd(v.y, b.ln2_w, b.ln2_b, b.eps2, v.f, v.mean2, v.rstd2, b.w1_wp, b.w2_wp, b.wp_stride, b.b1, b.b2, v.hdrop, out, M, d, dff,
                            b.p_drop, b.seed_h, b.sid_h, b.seed_o, b.sid_o, stream));
    return RT_OK;
  }
  { Timed t(T_LN_FWD, 0, 0, 0, stream); RT_TRY(rt_layernorm_fwd(v.y, b.ln2_w, b.ln2_b, b.eps2, M, d, v.f, v.mean2, v.rstd2, stream)); }
  { Timed t(T_GEMM, M, dff, d, stream);                                                                                 // h = relu(W1 f + b1)
    int rc = wp_one(v.f, d, b.w1_wp, b.wp_stride, d, 0, v.h, dff, b.b1, nullptr, 0, M, dff, d, 1, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(v.f, d, 1, b.w1, d, 1, v.h, dff, b.b1, nullptr, 0, nullptr, M, dff, d, 1, 1, nullptr, 0, stream);
    RT_TRY(rc); }
  if (b.p_drop > 0.f) {
    { Timed t(T_DROP_FWD, 0, 0, 0, stream);
      RT_TRY(rt_act_dropout_fwd(v.h, 0, b.p_drop, b.seed_h, b.sid_h, (int64_t)M * dff, nullptr, v.hdrop, stream)); }
    float* o = v.q + 5 * al((size_t)M * d);   // the spare region: o = W2 hdrop + b2 (dead after the next launch)
    { Timed t(T_GEMM, M, d, dff, stream);
      int rc = wp_one(v.hdrop, dff, b.w2_wp, b.wp_stride, dff, 0, o, d, b.b2, nullptr, 0, M, d, dff, 0, stream);
      if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(v.hdrop, dff, 1, b.w2, dff, 1, o, d, b.b2, nullptr, 0, nullptr, M, d, dff, 0, 1, nullptr, 0, stream);
      RT_TRY(rc); }
    { Timed t(T_DROP_FWD, 0, 0, 0, stream);
      RT_TRY(rt_act_dropout_fwd(o, 0, b.p_drop, b.seed_o, b.sid_o, (int64_t)M * d, v.f, out, stream)); }              // out = f + dropout(o)
  } else {
    Timed t(T_GEMM, M, d, dff, stream);
    int rc = wp_one(v.h, dff, b.w2_wp, b.wp_stride, dff, 0, out, d, b.b2, v.f, d, M, d, dff, 0, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(v.h, dff, 1, b.w2, dff, 1, out, d, b.b2, v.f, d, nullptr, M, d, dff, 0, 1, nullptr, 0, stream);
    RT_TRY(rc);
  }
  return RT_OK;
}

// Backward: g_x [rows, d] and the flat parameter gradient `grads` in the order ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b,
// w1, b1, w2, b2 (each segment starts at a multiple of 4 floats: rt_sasrec_block_grad_offsets).  Weight gradients are issued on the
// side stream (use_side != 0 and RT_SIDE_STREAM != 0): call rt_side_join before reading them; scratch / saved / x / g_out must stay
// alive until then.
void rt_sasrec_block_grad_offsets(int32_t d, int32_t dff, int64_t* offsets13) {
  const int64_t sizes[12] = {d, d, 3LL * d * d, 3LL * d, (int64_t)d * d, d, d, d, (int64_t)dff * d, dff, (int64_t)d * dff, d};
  int64_t o = 0;
  for (int i = 0; i < 12; ++i) { offsets13[i] = o; o += (sizes[i] + 3) & ~3LL; }
  offsets13[12] = o;
}

int rt_sasrec_block_packed_bwd(const rt_sasrec_block* blk, const float* x, const float* saved, const float* g_out, float* g_x, float* grads,
                               void* scratch, size_t scratch_bytes, int32_t wgrad_splits, int32_t use_side, hipStream_t stream) {
  (void)hipGetLastError();
  if (blk == nullptr || x == nullptr || saved == nullptr || g_out == nullptr || g_x == nullptr || grads == nullptr || scratch == nullptr)
    return RT_ERR_INVALID_ARG;
  const rt_sasrec_block& b = *blk;
  const int M = b.rows, d = b.d, dff = b.dff, hd = d / b.H;
  const int sp = wgrad_splits > 1 ? wgrad_splits : 1;
  if (scratch_bytes < rt_sasrec_block_bwd_scratch_bytes(M, b.B, d, dff, b.H, sp)) return RT_ERR_WORKSPACE;
  const SavedView v = carve_saved(b, const_cast<float*>(saved));
  int64_t go[13];
  rt_sasrec_block_grad_offsets(d, dff, go);
  float *d_ln1w = grads + go[0], *d_ln1b = grads + go[1], *d_in_w = grads + go[2], *d_in_b = grads + go[3], *d_wo = grads + go[4],
        *d_bo = grads + go[5], *d_ln2w = grads + go[6], *d_ln2b = grads + go[7], *d_w1 = grads + go[8], *d_b1 = grads + go[9],
        *d_w2 = grads + go[10], *d_b2 = grads + go[11];
  // scratch carve-up
  float* p = reinterpret_cast<float*>(scratch);
  const size_t Md = al((size_t)M * d), Mf = al((size_t)M * dff);
  float* g_o = p; p += Md; float* g_f = p; p += Md; float* g_y = p; p += Md; float* g_A = p; p += Md; float* gQ = p; p += Md;
  float* g_q = p; p += Md; float* g_kv = p; p += Md;
  float* g_hd = p; p += Mf; float* g_h = p; p += Mf;
  float* gKV = p; p += al((size_t)M * 2 * d);
  float* delta = p; p += al((size_t)M * b.H);
  float* part = p; p += al((size_t)b.B * d);
  unsigned char* bp = reinterpret_cast<unsigned char*>(p);
  const size_t lnws = (rt_layernorm_bwd_workspace_bytes(M, d) + 255) & ~(size_t)255;
  void* ln_ws1 = bp; bp += lnws; void* ln_ws2 = bp; bp += lnws;
  void* sk_ws = bp;
  const size_t sk_bytes = scratch_bytes - (size_t)(bp - reinterpret_cast<unsigned char*>(scratch));

  Side* side = (use_side && side_enabled()) ? side_of_current_device() : nullptr;
  hipStream_t ws = side != nullptr ? side->stream : stream;     // where the weight gradients go
  auto fork = [&]() -> int {                                     // the side stream sees everything issued so far on `stream`
    if (side == nullptr) return RT_OK;
    RT_CHECK_HIP(hipEventRecord(side->fork, stream));
    RT_CHECK_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
    side->dirty = true;
    return RT_OK;
  };
  auto wgrad = [&](const float* dy, int ldy, const float* in, int ldin, float* dw, int n_out, int n_in, float* db) -> int {
    Timed t(T_GEMM, n_out, n_in, M, ws);
    return rt_gemm(dy, ldy, 0, in, ldin, 0, dw, n_in, nullptr, nullptr, 0, db, n_out, n_in, M, 0, sp, sp > 1 ? sk_ws : nullptr,
                   sp > 1 ? sk_bytes : 0, ws);                   // dW = dy^T in, db = colsum(dy) from the staged dy^T tiles
  };

  // several weight gradients of the same rows as ONE split-K launch + one combine (else: a product and a combine each)
  constexpr int grouped_on = 1;
  auto wgrads = [&](const rt_wgrad_problem* pr, int n) -> int {
    int rc = RT_ERR_UNSUPPORTED;
    if (grouped_on && (d % 128) == 0 && (dff % 128) == 0) {
      long long fl = 0;
      for (int i = 0; i < n; ++i) fl += (long long)pr[i].n_out * pr[i].n_in;
      Timed t(T_GEMM_GROUPED, fl * M, 1, 1, ws);
      rc = rt_wgrad_grouped(pr, n, M, sp, sk_ws, sk_bytes, ws);
    }
    if (rc == RT_ERR_UNSUPPORTED) {
      for (int i = 0; i < n; ++i) RT_TRY(wgrad(pr[i].dy, (int)pr[i].ldy, pr[i].in, (int)pr[i].ldin, pr[i].dw, pr[i].n_out, pr[i].n_in, pr[i].db));
      rc = RT_OK;
    }
    return rc;
  };

  // ---- feed-forward: out = f + dropout(o), o = W2 hdrop + b2, hdrop = dropout(relu(W1 f + b1)); attention output: y = q + Wo A + bo
  if (tail_fused(b)) {   // g_o, g_h, g_y (LN2 backward on chip) and g_A from one launch; the three weight gradients fork behind it
    const float* g_o_c = b.p_drop > 0.f ? g_o : g_out;
    float* ln_part = reinterpret_cast<float*>(ln_ws1);      // (M / 64 partials fit the LayerNorm backward's own workspace)
    { Timed t(T_FFN_BWD, M, 2 * dff + d, d, stream);
      RT_TRY(rt_block_tail_bwd(g_out, v.hdrop, v.y, v.mean2, v.rstd2, b.ln2_w, b.out_wp, b.w1_wp, b.w2_wp, b.wp_stride, g_o, g_h, g_y, g_A, ln_part, M, d,
                               dff, b.p_drop, b.seed_o, b.sid_o, stream)); }
    RT_TRY(fork());
    { Timed t(T_MISC, 0, 0, 0, ws); RT_TRY(rt_layernorm_bwd_reduce(ln_part, M / 64, d, d_ln2w, d_ln2b, ws)); }
    const rt_wgrad_problem pr[3] = {{g_o_c, d, v.hdrop, dff, d_w2, d_b2, d, dff}, {g_h, dff, v.f, d, d_w1, d_b1, dff, d}, {g_y, d, v.A, d, d_wo, d_bo, d, d}};
    RT_TRY(wgrads(pr, 3));
  } else {
  if (ffn_fused(b)) {   // g_o, g_h, g_f from one launch; both weight gradients fork behind it
    const float* g_o_c = b.p_drop > 0.f ? g_o : g_out;
    { Timed t(T_FFN_BWD, M, 2 * dff, d, stream);
      RT_TRY(rt_ffn_fused_bwd(g_out, v.hdrop, b.w1_wp, b.w2_wp, b.wp_stride, g_o, g_h, g_f, M, d, dff, b.p_drop, b.seed_o, b.sid_o, stream)); }
    RT_TRY(fork());
    RT_TRY(wgrad(g_o_c, d, v.hdrop, dff, d_w2, d, dff, d_b2));
    RT_TRY(wgrad(g_h, dff, v.f, d, d_w1, dff, d, d_b1));
  } else {
  const float* g_o_c = g_out;
  if (b.p_drop > 0.f) {
    Timed t(T_DROP_BWD, 0, 0, 0, stream);
    RT_TRY(rt_act_dropout_bwd(g_out, g_out, 0, b.p_drop, b.seed_o, b.sid_o, (int64_t)M * d, g_o, stream));
    g_o_c = g_o;
  }
  RT_TRY(fork());
  RT_TRY(wgrad(g_o_c, d, v.hdrop, dff, d_w2, d, dff, d_b2));
  { Timed t(T_GEMM, M, dff, d, stream);                                                       // g_hd = g_o W2  (W2 [d, dff])
    int rc = wp_one(g_o_c, d, b.w2_wp, b.wp_stride, dff, 1, g_hd, dff, nullptr, nullptr, 0, M, dff, d, 0, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(g_o_c, d, 1, b.w2, dff, 0, g_hd, dff, nullptr, nullptr, 0, nullptr, M, dff, d, 0, 1, nullptr, 0, stream);
    RT_TRY(rc); }
  { Timed t(T_DROP_BWD, 0, 0, 0, stream);   // dropout mask and relu'(h) in one pass
    RT_TRY(rt_act_dropout_bwd(g_hd, v.h, 1, b.p_drop, b.seed_h, b.sid_h, (int64_t)M * dff, g_h, stream)); }
  RT_TRY(fork());
  RT_TRY(wgrad(g_h, dff, v.f, d, d_w1, dff, d, d_b1));
  { Timed t(T_GEMM, M, d, dff, stream);     // g_f = g_h W1 + g_out: the residual branch rides in the dgrad epilogue
    int rc = wp_one(g_h, dff, b.w1_wp, b.wp_stride, d, 1, g_f, d, nullptr, g_out, d, M, d, dff, 0, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(g_h, dff, 1, b.w1, d, 0, g_f, d, nullptr, g_out, d, nullptr, M, d, dff, 0, 1, nullptr, 0, stream);
    RT_TRY(rc); }
  }
  { Timed t(T_LN_BWD, 0, 0, 0, stream);
    RT_TRY(rt_layernorm_bwd_rows(g_f, v.y, b.ln2_w, v.mean2, v.rstd2, nullptr, nullptr, 0, 0, M, d, g_y, ln_ws1, lnws, stream)); }
  RT_TRY(fork());      // d ln_w / d ln_b are the optimiser's: their combine leaves the critical path
  { Timed t(T_MISC, 0, 0, 0, ws);
    RT_TRY(rt_layernorm_bwd_combine(ln_ws1, lnws, M, d, d_ln2w, d_ln2b, ws)); }
  // ---- attention: y = q + Wo A + bo
  RT_TRY(fork());
  RT_TRY(wgrad(g_y, d, v.A, d, d_wo, d, d, d_bo));
  { Timed t(T_GEMM, M, d, d, stream);
    int rc = wp_one(g_y, d, b.out_wp, b.wp_stride, d, 1, g_A, d, nullptr, nullptr, 0, M, d, d, 0, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(g_y, d, 1, b.out_w, d, 0, g_A, d, nullptr, nullptr, 0, nullptr, M, d, d, 0, 1, nullptr, 0, stream);
    RT_TRY(rc); }
  }
  RT_TRY(zero_tail(gQ, b, d, stream));      // rows behind the sessions must read as zero in the weight gradients
  RT_TRY(zero_tail(gKV, b, 2 * d, stream));
  const float* bk = b.pad_keys ? b.in_b + d : nullptr;
  const float* bv = b.pad_keys ? b.in_b + 2 * d : nullptr;
  { Timed t(T_ATTN_BWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_bwd(v.Q, d, v.KV, 2 * d, v.KV + d, 2 * d, v.A, d, g_A, d, v.lse, b.cu, bk, bv, b.B, b.H, hd, b.window, b.window, b.p_drop,
                             b.seed_attn, gQ, d, gKV, 2 * d, gKV + d, 2 * d, delta, b.pad_keys ? part : nullptr, stream)); }
  RT_TRY(fork());
  {
    const rt_wgrad_problem pr[2] = {{gQ, d, v.q, d, d_in_w, d_in_b, d, d}, {gKV, 2 * d, x, d, d_in_w + (size_t)d * d, d_in_b + d, 2 * d, d}};
    RT_TRY(wgrads(pr, 2));
  }
  if (b.pad_keys) {   // the pad keys' share: b_k has no gradient in total (it shifts every logit of a query alike), b_v gets theirs
    Timed t(T_MISC, 0, 0, 0, ws);
    RT_CHECK_HIP(hipMemsetAsync(d_in_b + d, 0, (size_t)d * sizeof(float), ws));
    RT_TRY(rt_colsum(part, d, b.B, d, d_in_b + 2 * d, ws));
  }
  {
    Timed t(T_GEMM_GROUPED, (long long)M * d * d + (long long)M * d * 2 * d, 1, 1, stream);
    int rc = RT_ERR_UNSUPPORTED;
    if (b.in_wp != nullptr) {
      rt_gemm_wp_problem wp[2] = {{gQ, d, b.in_wp, b.wp_stride, d, g_q, d, nullptr, g_y, d, M, d, d, 0},
                                  {gKV, 2 * d, b.in_wp + (size_t)d * d, b.wp_stride, d, g_kv, d, nullptr, nullptr, 0, M, d, 2 * d, 0}};
      rc = rt_gemm_wp(wp, 2, 1, stream);
    }
    if (rc == RT_ERR_UNSUPPORTED) {
      rt_gemm_problem pr[2] = {{gQ, d, b.in_w, d, g_q, d, nullptr, g_y, d, M, d, d, 0},                                    // g_q = gQ Wq + g_y
                               {gKV, 2 * d, b.in_w + (size_t)d * d, d, g_kv, d, nullptr, nullptr, 0, M, d, 2 * d, 0}};     // g_kv = gKV Wkv
      rc = rt_gemm_grouped(pr, 2, 1, 0, stream);
    }
    RT_TRY(rc);
  }
  { Timed t(T_LN_BWD, 0, 0, 0, stream);     // g_x = LN1'(g_q) + g_kv
    RT_TRY(rt_layernorm_bwd_rows(g_q, x, b.ln1_w, v.mean1, v.rstd1, g_kv, nullptr, 0, 0, M, d, g_x, ln_ws2, lnws, stream)); }
  RT_TRY(fork());      // d ln_w / d ln_b are the optimiser's: their combine leaves the critical path
  { Timed t(T_MISC, 0, 0, 0, ws);
    RT_TRY(rt_layernorm_bwd_combine(ln_ws2, lnws, M, d, d_ln1w, d_ln1b, ws)); }
  return RT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// One packed Pre-LN block (net_blocks.py:223-262: BERT4Rec's stack) with key-padding masks — on packed rows the pad keys do not exist.
//   h = LN1(x); qkv = h Win^T + bin; A = attention(qkv) (bidirectional, or causal); x1 = x + drop1(A Wo^T + bo);
//   g = LN2(x1); a = drop_h(gelu(g W1^T + b1)); x2 = x1 + drop2(a W2^T + b2); out = drop3(x2)
// Same kernels, order and dropout streams as `nn.PreLNTransformerLayer.forward_packed` (the cross-check of the test suite).
// ------------------------------------------------------------------------------------------------------------------------------------
struct rt_preln_block {
  int32_t rows, rows_real, B, H, d, dff, window, causal;
  float p_drop, eps1, eps2;
  uint64_t seed_attn, seed1, sid1, seed_h, sid_h, seed2, sid2, seed3, sid3;     // dropout streams in forward order (ops.RNG)
  const int64_t* cu;
  const float *ln1_w, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
  const uint16_t *in_wp, *out_wp, *w1_wp, *w2_wp;      // optional bf16 planes of the weights (see rt_sasrec_block)
  int64_t wp_stride;
};

size_t rt_preln_block_saved_floats(int32_t rows, int32_t d, int32_t dff, int32_t H) {
  const size_t M = (size_t)rows;
  return al(M * d) * 6 /* h A x1 g + two spares */ + al(M * 3 * d) /* qkv */ + al(M * dff) * 2 /* z a */ + al(M * H) /* lse */ + 4 * al(M);
}
size_t rt_preln_block_bwd_scratch_bytes(int32_t rows, int32_t d, int32_t dff, int32_t H, int32_t wgrad_splits) {
  const size_t M = (size_t)rows;
  const size_t fl = al(M * d) * 7 /* g_x2 g_f g_g g_x1 g_mo g_A g_h */ + al(M * dff) * 2 /* g_a g_z */ + al(M * 3 * d) /* dqkv */ + al(M * H) /* delta */;
  size_t by = fl * 4 + 2 * ((rt_layernorm_bwd_workspace_bytes(rows, d) + 255) & ~(size_t)255);
  size_t sk = 0;
  const int spl = wgrad_splits > 1 ? wgrad_splits : 1;
  const size_t cands[4] = {rt_gemm_workspace_bytes(d, dff, rows, spl), rt_gemm_workspace_bytes(dff, d, rows, spl), rt_gemm_workspace_bytes(d, d, rows, spl),
                           rt_gemm_workspace_bytes(3 * d, d, rows, spl)};
  for (size_t c : cands) sk = c > sk ? c : sk;
  return by + ((sk + 255) & ~(size_t)255) + 256;
}

namespace {
struct PreLNSaved { float *h, *A, *x1, *g, *spare, *spare2, *qkv, *z, *a, *lse, *mean1, *rstd1, *mean2, *rstd2; };
PreLNSaved carve_preln(const rt_preln_block& b, float* base) {
  const size_t M = (size_t)b.rows;
  PreLNSaved v;
  float* p = base;
  v.h = p; p += al(M * b.d); v.A = p; p += al(M * b.d); v.x1 = p; p += al(M * b.d); v.g = p; p += al(M * b.d); v.spare = p; p += al(M * b.d); v.spare2 = p; p += al(M * b.d);
  v.qkv = p; p += al(M * 3 * b.d);
  v.z = p; p += al(M * b.dff); v.a = p; p += al(M * b.dff);
  v.lse = p; p += al(M * b.H);
  v.mean1 = p; p += al(M); v.rstd1 = p; p += al(M); v.mean2 = p; p += al(M); v.rstd2 = p; p += al(M);
  return v;
}
// forward product y = A W^T (+ bias) (+ R): pre-split planes where the block carries them and the shape is an exact tile grid
int fwd_gemm(const float* A, int lda, const float* W, const uint16_t* Wp, int64_t wps, int ldw, float* C, int ldc, const float* bias,
             const float* R, int ldr, int M, int N, int K, hipStream_t s) {
  Timed t(T_GEMM, M, N, K, s);
  int rc = wp_one(A, lda, Wp, wps, ldw, 0, C, ldc, bias, R, ldr, M, N, K, 0, s);
  if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(A, lda, 1, W, ldw, 1, C, ldc, bias, R, ldr, nullptr, M, N, K, 0, 1, nullptr, 0, s);
  return rc;
}
// data gradient dx = dy W (+ R), W [N_out, N_in] row-major
int dgrad_gemm(const float* dy, int ldy, const float* W, const uint16_t* Wp, int64_t wps, int n_out, int n_in, float* dx, const float* R, int M,
               hipStream_t s) {
  Timed t(T_GEMM, M, n_in, n_out, s);
  int rc = wp_one(dy, ldy, Wp, wps, n_in, 1, dx, n_in, nullptr, R, n_in, M, n_in, n_out, 0, s);
  if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(dy, ldy, 1, W, n_in, 0, dx, n_in, nullptr, R, n_in, nullptr, M, n_in, n_out, 0, 1, nullptr, 0, s);
  return rc;
}
}  // namespace

int rt_preln_block_packed_fwd(const rt_preln_block* blk, const float* x, float* saved, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (blk == nullptr || x == nullptr || saved == nullptr || out == nullptr) return RT_ERR_INVALID_ARG;
  const rt_preln_block& b = *blk;
  const int M = b.rows, d = b.d, dff = b.dff, hd = d / b.H;
  if (M <= 0 || d <= 0 || b.H <= 0 || d % b.H != 0 || b.cu == nullptr) return RT_ERR_INVALID_ARG;
  const PreLNSaved v = carve_preln(b, saved);
  const int64_t nd = (int64_t)M * d, nf = (int64_t)M * dff;
  { Timed t(T_LN_FWD, 0, 0, 0, stream); RT_TRY(rt_layernorm_fwd(x, b.ln1_w, b.ln1_b, b.eps1, M, d, v.h, v.mean1, v.rstd1, stream)); }
  RT_TRY(fwd_gemm(v.h, d, b.in_w, b.in_wp, b.wp_stride, d, v.qkv, 3 * d, b.in_b, nullptr, 0, M, 3 * d, d, stream));
  if (b.rows_real < b.rows) {   // rows behind the sessions must read as finite zeros
    RT_CHECK_HIP(hipMemsetAsync(v.A + (size_t)b.rows_real * d, 0, (size_t)(b.rows - b.rows_real) * d * sizeof(float), stream));
    RT_CHECK_HIP(hipMemsetAsync(v.lse + (size_t)b.rows_real * b.H, 0, (size_t)(b.rows - b.rows_real) * b.H * sizeof(float), stream));
  }
  if (b.causal) {
    Timed t(T_ATTN_FWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_train_fwd(v.qkv, 3 * d, v.qkv + d, 3 * d, v.qkv + 2 * d, 3 * d, b.cu, nullptr, nullptr, b.B, b.H, hd, b.window, b.window,
                                   b.p_drop, b.seed_attn, v.A, d, v.lse, stream));
  } else {
    Timed t(T_ATTN_BIDIR_FWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_bidir_fwd(v.qkv, 3 * d, v.qkv + d, 3 * d, v.qkv + 2 * d, 3 * d, b.cu, b.B, b.H, hd, b.window, b.p_drop, b.seed_attn, v.A, d,
                                   v.lse, stream));
  }
  if (b.p_drop > 0.f) {
    RT_TRY(fwd_gemm(v.A, d, b.out_w, b.out_wp, b.wp_stride, d, v.spare, d, b.out_b, nullptr, 0, M, d, d, stream));            // mo = A Wo^T + bo
    { Timed t(T_DROP_FWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_fwd(v.spare, 0, b.p_drop, b.seed1, b.sid1, nd, x, v.x1, stream)); }   // x1 = x + drop(mo)
  } else {
    RT_TRY(fwd_gemm(v.A, d, b.out_w, b.out_wp, b.wp_stride, d, v.x1, d, b.out_b, x, d, M, d, d, stream));
  }
  { Timed t(T_LN_FWD, 0, 0, 0, stream); RT_TRY(rt_layernorm_fwd(v.x1, b.ln2_w, b.ln2_b, b.eps2, M, d, v.g, v.mean2, v.rstd2, stream)); }
  RT_TRY(fwd_gemm(v.g, d, b.w1, b.w1_wp, b.wp_stride, d, v.z, dff, b.b1, nullptr, 0, M, dff, d, stream));                       // z = g W1^T + b1
  { Timed t(T_DROP_FWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_fwd(v.z, 2 /* gelu */, b.p_drop, b.seed_h, b.sid_h, nf, nullptr, v.a, stream)); }
  if (b.p_drop > 0.f) {
    RT_TRY(fwd_gemm(v.a, dff, b.w2, b.w2_wp, b.wp_stride, dff, v.spare, d, b.b2, nullptr, 0, M, d, dff, stream));              // f = a W2^T + b2
    { Timed t(T_DROP_FWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_fwd(v.spare, 0, b.p_drop, b.seed2, b.sid2, nd, v.x1, v.spare2, stream)); }   // x2 = x1 + drop(f)
    { Timed t(T_DROP_FWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_fwd(v.spare2, 0, b.p_drop, b.seed3, b.sid3, nd, nullptr, out, stream)); }   // dropout_3 (net_blocks.py:260)
  } else {
    RT_TRY(fwd_gemm(v.a, dff, b.w2, b.w2_wp, b.wp_stride, dff, out, d, b.b2, v.x1, d, M, d, dff, stream));
  }
  return RT_OK;
}

// Backward: g_x [rows, d] and the flat parameter gradient in the order and offsets of rt_sasrec_block_grad_offsets(d, dff) (ln1_w, ln1_b,
// in_w [3d, d], in_b, out_w, out_b, ln2_w, ln2_b, w1, b1, w2, b2).  Weight gradients on the side stream as in rt_sasrec_block_packed_bwd.
int rt_preln_block_packed_bwd(const rt_preln_block* blk, const float* x, const float* saved, const float* g_out, float* g_x, float* grads,
                              void* scratch, size_t scratch_bytes, int32_t wgrad_splits, int32_t use_side, hipStream_t stream) {
  (void)hipGetLastError();
  if (blk == nullptr || x == nullptr || saved == nullptr || g_out == nullptr || g_x == nullptr || grads == nullptr || scratch == nullptr)
    return RT_ERR_INVALID_ARG;
  const rt_preln_block& b = *blk;
  const int M = b.rows, d = b.d, dff = b.dff, hd = d / b.H;
  const int sp = wgrad_splits > 1 ? wgrad_splits : 1;
  if (scratch_bytes < rt_preln_block_bwd_scratch_bytes(M, d, dff, b.H, sp)) return RT_ERR_WORKSPACE;
  const PreLNSaved v = carve_preln(b, const_cast<float*>(saved));
  int64_t go[13];
  rt_sasrec_block_grad_offsets(d, dff, go);
  float *d_ln1w = grads + go[0], *d_ln1b = grads + go[1], *d_in_w = grads + go[2], *d_in_b = grads + go[3], *d_wo = grads + go[4],
        *d_bo = grads + go[5], *d_ln2w = grads + go[6], *d_ln2b = grads + go[7], *d_w1 = grads + go[8], *d_b1 = grads + go[9],
        *d_w2 = grads + go[10], *d_b2 = grads + go[11];
  float* p = reinterpret_cast<float*>(scratch);
  const size_t Md = al((size_t)M * d), Mf = al((size_t)M * dff);
  float* g_x2 = p; p += Md; float* g_f = p; p += Md; float* g_g = p; p += Md; float* g_x1 = p; p += Md; float* g_mo = p; p += Md;
  float* g_A = p; p += Md; float* g_h = p; p += Md;
  float* g_a = p; p += Mf; float* g_z = p; p += Mf;
  float* dqkv = p; p += al((size_t)M * 3 * d);
  float* delta = p; p += al((size_t)M * b.H);
  unsigned char* bp = reinterpret_cast<unsigned char*>(p);
  const size_t lnws = (rt_layernorm_bwd_workspace_bytes(M, d) + 255) & ~(size_t)255;
  void* ln_ws1 = bp; bp += lnws; void* ln_ws2 = bp; bp += lnws;
  void* sk_ws = bp;
  const size_t sk_bytes = scratch_bytes - (size_t)(bp - reinterpret_cast<unsigned char*>(scratch));
  const int64_t nd = (int64_t)M * d, nf = (int64_t)M * dff;

  Side* side = (use_side && side_enabled()) ? side_of_current_device() : nullptr;
  hipStream_t ws = side != nullptr ? side->stream : stream;
  auto fork = [&]() -> int {
    if (side == nullptr) return RT_OK;
    RT_CHECK_HIP(hipEventRecord(side->fork, stream));
    RT_CHECK_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
    side->dirty = true;
    return RT_OK;
  };
  auto wgrad = [&](const float* dy, int ldy, const float* in, int ldin, float* dw, int n_out, int n_in, float* db) -> int {
    Timed t(T_GEMM, n_out, n_in, M, ws);
    return rt_gemm(dy, ldy, 0, in, ldin, 0, dw, n_in, nullptr, nullptr, 0, db, n_out, n_in, M, 0, sp, sp > 1 ? sk_ws : nullptr,
                   sp > 1 ? sk_bytes : 0, ws);
  };

  // ---- out = drop3(x2), x2 = x1 + drop2(f), f = a W2^T + b2, a = drop_h(gelu(z)), z = g W1^T + b1, g = LN2(x1)
  const float* g_x2c = g_out;    // gradient of x2 (= of x1 through the skip)
  const float* g_fc = g_out;     // gradient of f
  if (b.p_drop > 0.f) {
    { Timed t(T_DROP_BWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_bwd(g_out, g_out, 0, b.p_drop, b.seed3, b.sid3, nd, g_x2, stream)); }
    { Timed t(T_DROP_BWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_bwd(g_x2, g_x2, 0, b.p_drop, b.seed2, b.sid2, nd, g_f, stream)); }
    g_x2c = g_x2; g_fc = g_f;
  }
  RT_TRY(fork());
  RT_TRY(wgrad(g_fc, d, v.a, dff, d_w2, d, dff, d_b2));
  RT_TRY(dgrad_gemm(g_fc, d, b.w2, b.w2_wp, b.wp_stride, d, dff, g_a, nullptr, M, stream));                        // g_a = g_f W2
  { Timed t(T_DROP_BWD, 0, 0, 0, stream); RT_TRY(rt_act_dropout_bwd(g_a, v.z, 2 /* gelu */, b.p_drop, b.seed_h, b.sid_h, nf, g_z, stream)); }
  RT_TRY(fork());
  RT_TRY(wgrad(g_z, dff, v.g, d, d_w1, dff, d, d_b1));
  RT_TRY(dgrad_gemm(g_z, dff, b.w1, b.w1_wp, b.wp_stride, dff, d, g_g, nullptr, M, stream));                       // g_g = g_z W1
  { Timed t(T_LN_BWD, 0, 0, 0, stream);     // g_x1 = LN2'(g_g) + g_x2 (the skip)
    RT_TRY(rt_layernorm_bwd_rows(g_g, v.x1, b.ln2_w, v.mean2, v.rstd2, g_x2c, nullptr, 0, 0, M, d, g_x1, ln_ws1, lnws, stream)); }
  RT_TRY(fork());      // d ln_w / d ln_b are the optimiser's: their combine leaves the critical path
  { Timed t(T_MISC, 0, 0, 0, ws);
    RT_TRY(rt_layernorm_bwd_combine(ln_ws1, lnws, M, d, d_ln2w, d_ln2b, ws)); }
  // ---- x1 = x + drop1(mo), mo = A Wo^T + bo, A = attention(qkv), qkv = h Win^T + bin, h = LN1(x)
  const float* g_moc = g_x1;
  if (b.p_drop > 0.f) {
    Timed t(T_DROP_BWD, 0, 0, 0, stream);
    RT_TRY(rt_act_dropout_bwd(g_x1, g_x1, 0, b.p_drop, b.seed1, b.sid1, nd, g_mo, stream));
    g_moc = g_mo;
  }
  RT_TRY(fork());
  RT_TRY(wgrad(g_moc, d, v.A, d, d_wo, d, d, d_bo));
  RT_TRY(dgrad_gemm(g_moc, d, b.out_w, b.out_wp, b.wp_stride, d, d, g_A, nullptr, M, stream));                     // g_A = g_mo Wo
  if (b.rows_real < b.rows)     // rows behind the sessions must read as zero in the weight gradients
    RT_CHECK_HIP(hipMemsetAsync(dqkv + (size_t)b.rows_real * 3 * d, 0, (size_t)(b.rows - b.rows_real) * 3 * d * sizeof(float), stream));
  if (b.causal) {
    Timed t(T_ATTN_BWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_bwd(v.qkv, 3 * d, v.qkv + d, 3 * d, v.qkv + 2 * d, 3 * d, v.A, d, g_A, d, v.lse, b.cu, nullptr, nullptr, b.B, b.H, hd, b.window,
                             b.window, b.p_drop, b.seed_attn, dqkv, 3 * d, dqkv + d, 3 * d, dqkv + 2 * d, 3 * d, delta, nullptr, stream));
  } else {
    Timed t(T_ATTN_BIDIR_BWD, 0, 0, 0, stream);
    RT_TRY(rt_mha_varlen_bidir_bwd(v.qkv, 3 * d, v.qkv + d, 3 * d, v.qkv + 2 * d, 3 * d, v.A, d, g_A, d, v.lse, b.cu, b.B, b.H, hd, b.window, b.p_drop,
                                   b.seed_attn, dqkv, 3 * d, dqkv + d, 3 * d, dqkv + 2 * d, 3 * d, delta, stream));
  }
  RT_TRY(fork());
  RT_TRY(wgrad(dqkv, 3 * d, v.h, d, d_in_w, 3 * d, d, d_in_b));
  RT_TRY(dgrad_gemm(dqkv, 3 * d, b.in_w, b.in_wp, b.wp_stride, 3 * d, d, g_h, nullptr, M, stream));                // g_h = dqkv Win
  { Timed t(T_LN_BWD, 0, 0, 0, stream);     // g_x = LN1'(g_h) + g_x1 (the skip)
    RT_TRY(rt_layernorm_bwd_rows(g_h, x, b.ln1_w, v.mean1, v.rstd1, g_x1, nullptr, 0, 0, M, d, g_x, ln_ws2, lnws, stream)); }
  RT_TRY(fork());      // d ln_w / d ln_b are the optimiser's: their combine leaves the critical path
  { Timed t(T_MISC, 0, 0, 0, ws);
    RT_TRY(rt_layernorm_bwd_combine(ln_ws2, lnws, M, d, d_ln1w, d_ln1b, ws)); }
  return RT_OK;
}

// Inference (recommend(): eval mode, no dropout).  last_rows == NULL: out [rows, d] for every row.  last_rows [B] (= cu[1:] - 1): the
// block's output at the LAST position of every session only, out [B, d] — the key / value projection is then the only product over
// all rows (lightning.py:393-397 keeps session_embs[:, -1, :]).  scratch: rt_sasrec_block_infer_scratch_floats floats.
size_t rt_sasrec_block_infer_scratch_floats(int32_t rows, int32_t B, int32_t d, int32_t dff, int32_t last_only) {
  const size_t M = (size_t)rows, R = last_only ? (size_t)B : M;
  // (last_only: the KV area doubles as qk | xbar [B, H, d] each of the projection-free last-query form: H <= d / 32 heads)
  const size_t kv = M * 2 * d, qx = last_only ? 2 * (R + d) * (size_t)(d / 32) * d : 0;      // (+ the two head-expanded weights)
  return (last_only ? al(R * d) : 0) /* x_last */ + al(R * d) * 5 /* q Q A y f */ + al(kv > qx ? kv : qx) /* KV */ + al(R * dff) /* h */ + 2 * al(R);
}
int rt_sasrec_block_packed_infer(const rt_sasrec_block* blk, const float* x, const float* q_in, const float* Q_in, const float* kv_in,
                                 const int64_t* last_rows, float* scratch, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  const bool pre = q_in != nullptr && Q_in != nullptr && kv_in != nullptr && last_rows == nullptr;      // LN1(x), Q and K | V handed in: x unused
  if (blk == nullptr || (x == nullptr && !pre) || scratch == nullptr || out == nullptr || (q_in != nullptr) != (Q_in != nullptr) ||
      (q_in != nullptr && !pre))
    return RT_ERR_INVALID_ARG;
  const rt_sasrec_block& b = *blk;
  const int M = b.rows, d = b.d, dff = b.dff, hd = d / b.H;
  const bool last = last_rows != nullptr;
  const int R = last ? b.B : M;
  float* p = scratch;
  const float* xin = x;
  if (last) { float* xl = p; p += al((size_t)R * d); RT_TRY(rt_gather_rows(x, d, last_rows, R, d, xl, d, stream)); xin = xl; }
  float* q = p; p += al((size_t)R * d); float* Q = p; p += al((size_t)R * d); float* A = p; p += al((size_t)R * d);
  float* y = p; p += al((size_t)R * d); float* f = p; p += al((size_t)R * d);
  float* KV = p;
  { const size_t kv = (size_t)M * 2 * d, qx = last ? 2 * ((size_t)R + d) * (d / 32) * d : 0; p += al(kv > qx ? kv : qx); }
  float* h = p; p += al((size_t)R * dff);
  float* mean = p; p += al((size_t)R); float* rstd = p;
  const float* bk = b.pad_keys ? b.in_b + d : nullptr;
  const float* bv = b.pad_keys ? b.in_b + 2 * d : nullptr;
  if (pre) { q = const_cast<float*>(q_in); Q = const_cast<float*>(Q_in); }      // (read only from here on)
  else RT_TRY(rt_layernorm_fwd(xin, b.ln1_w, b.ln1_b, b.eps1, R, d, q, mean, rstd, stream));
  auto lin = [&](const float* A, int lda, const float* W, const uint16_t* Wp, int ldw, float* C, int ldc, const float* bias, const float* R, int ldr,
                 int rows, int N, int K, int relu) -> int {      // one forward product: pre-split planes where the shape allows
    int rc = wp_one(A, lda, Wp, b.wp_stride, ldw, 0, C, ldc, bias, R, ldr, rows, N, K, relu, stream);
    if (rc == RT_ERR_UNSUPPORTED) rc = rt_gemm(A, lda, 1, W, ldw, 1, C, ldc, bias, R, ldr, nullptr, rows, N, K, relu, 1, nullptr, 0, stream);
    return rc;
  };
  if (!last && kv_in != nullptr) {
    // keys | values handed in ([rows, 2d]: the FIRST block of recommend(), whose input is embedding row + positional row — W_kv (e + p) +
    // b = (W_kv e) + (W_kv p + b): two projected TABLES and a gather, models / nn.TransformerTorchBackbone.encode_last_packed): only the
    // query projection runs over the rows
    if (!pre) RT_TRY(lin(q, d, b.in_w, b.in_wp, d, Q, d, b.in_b, nullptr, 0, M, d, d, 0));
    { rt_sasrec_block bb = b; RT_TRY(zero_tail(A, bb, d, stream)); }
    RT_TRY(rt_mha_varlen_fwd(Q, d, kv_in, 2 * d, kv_in + d, 2 * d, b.cu, bk, bv, b.B, b.H, hd, b.window, b.window, A, d, stream));
  } else if (!last) {
    int rc = RT_ERR_UNSUPPORTED;
    if (b.in_wp != nullptr) {
      rt_gemm_wp_problem wp[2] = {{q, d, b.in_wp, b.wp_stride, d, Q, d, b.in_b, nullptr, 0, M, d, d, 0},
                                  {x, d, b.in_wp + (size_t)d * d, b.wp_stride, d, KV, 2 * d, b.in_b + d, nullptr, 0, M, 2 * d, d, 0}};
      rc = rt_gemm_wp(wp, 2, 0, stream);
    }
    if (rc == RT_ERR_UNSUPPORTED) {
      rt_gemm_problem pr[2] = {{q, d, b.in_w, d, Q, d, b.in_b, nullptr, 0, M, d, d, 0},
                               {x, d, b.in_w + (size_t)d * d, d, KV, 2 * d, b.in_b + d, nullptr, 0, M, 2 * d, d, 0}};
      rc = rt_gemm_grouped(pr, 2, 1, 1, stream);
    }
    RT_TRY(rc);
    { rt_sasrec_block bb = b; RT_TRY(zero_tail(A, bb, d, stream)); }
    RT_TRY(rt_mha_varlen_fwd(Q, d, KV, 2 * d, KV + d, 2 * d, b.cu, bk, bv, b.B, b.H, hd, b.window, b.window, A, d, stream));
  } else {
    RT_TRY(lin(q, d, b.in_w, b.in_wp, d, Q, d, b.in_b, nullptr, 0, R, d, d, 0));
    // The last query needs no key / value ROWS: q_h . (W_k,h x_j + b_k,h) = (W_k,h^T q_h) . x_j + const and sum_j p_j (W_v,h x_j + b_v,h) =
    // W_v,h (sum_j p_j x_j) + b_v,h — two [B, .] products per head around ONE pass over the block input (rt_mha_varlen_last_x_fwd) instead
    // of the [rows, d] x [d, 2d] projection (the largest product of the final block) and a pass over its 2 KB-per-row output.
    int rc = RT_ERR_UNSUPPORTED;
    if (b.H <= d / 32 && hd >= 8 && (d == 64 || d == 128 || d == 256 || d == 512)) {
      const size_t Hd = (size_t)b.H * d;
      float* qk = KV; float* xbar = qk + (size_t)R * Hd; float* Ek = xbar + (size_t)R * Hd; float* Ev = Ek + (size_t)d * Hd;
      // the per-head products as ONE exact-tile product each over head-expanded weights (rt_mha_last_x_expand: 2 x d x H d floats per call)
      RT_TRY(rt_mha_last_x_expand(b.in_w + (size_t)d * d, d, b.H, Ek, stream));
      RT_TRY(rt_mha_last_x_expand(b.in_w + 2 * (size_t)d * d, d, b.H, Ev, stream));
      RT_TRY(rt_gemm(Q, d, 1, Ek, (int64_t)Hd, 0, qk, (int64_t)Hd, nullptr, nullptr, 0, nullptr, R, (int32_t)Hd, d, 0, 1, nullptr, 0, stream));
      rc = rt_mha_varlen_last_x_fwd(qk, x, d, b.cu, b.B, b.H, d, b.window, b.window, b.pad_keys ? 1 : 0, -1, xbar, stream);
      if (rc == RT_OK)
        rc = rt_gemm(xbar, (int64_t)Hd, 1, Ev, (int64_t)Hd, 1, A, d, b.in_b + 2 * d, nullptr, 0, nullptr, R, d, (int32_t)Hd, 0, 1, nullptr, 0, stream);
    }
    if (rc == RT_ERR_UNSUPPORTED) {      // (a width the projection-free kernel does not tile: keys and values of every row)
      RT_TRY(lin(x, d, b.in_w + (size_t)d * d, b.in_wp != nullptr ? b.in_wp + (size_t)d * d : nullptr, d, KV, 2 * d, b.in_b + d, nullptr, 0, M, 2 * d, d, 0));
      rc = rt_mha_varlen_last_fwd(Q, d, KV, 2 * d, KV + d, 2 * d, b.cu, bk, bv, b.B, b.H, hd, b.window, b.window, A, d, stream);
    }
    RT_TRY(rc);
  }
  // The block's tail with the rows resident on chip (attn and q in, out out, + the scratch rows f — no y, h, statistics) pays while the
  // launch is a round or two of 64-row workgroups: every workgroup streams all three weights (1.2 MB of planes) for its 64 rows, twice the
  // weight traffic per row of the 128 x 128-tile products — at recommend()'s 10^5..10^6 rows per launch the separate products win
  // (visit v4j of round 4: encoder 23.7 vs 21.4 ms per 16,384 users)
  static const int infer_tail_rows = [] { const char* e = getenv("RT_INFER_TAIL_MAX_ROWS"); return e ? atoi(e) : 2 * 64 * rt_num_cus(); }();
  if (ffn_mode() == 2 && R <= infer_tail_rows && b.out_wp != nullptr && b.w1_wp != nullptr && b.w2_wp != nullptr &&
      rt_ffn_fused_supported(R, d, dff) == 1) {
    RT_TRY(rt_block_tail_fwd(A, q, b.out_wp, b.out_b, b.ln2_w, b.ln2_b, b.eps2, nullptr, f, nullptr, nullptr, b.w1_wp, b.w2_wp, b.wp_stride, b.b1, b.b2,
                             nullptr, out, R, d, dff, 0.f, 0, 0, 0, 0, 0, stream));
    return RT_OK;
  }
  RT_TRY(lin(A, d, b.out_w, b.out_wp, d, y, d, b.out_b, q, d, R, d, d, 0));
  RT_TRY(rt_layernorm_fwd(y, b.ln2_w, b.ln2_b, b.eps2, R, d, f, mean, rstd, stream));
  RT_TRY(lin(f, d, b.w1, b.w1_wp, d, h, dff, b.b1, nullptr, 0, R, dff, d, 1));
  RT_TRY(lin(h, dff, b.w2, b.w2_wp, dff, out, d, b.b2, f, d, R, d, dff, 0));
  return RT_OK;
}

}  // extern "C"
