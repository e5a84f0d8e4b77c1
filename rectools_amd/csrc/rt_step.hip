// One packed SASRec TRAINING STEP behind one C call (lightning.py:311-321 around sasrec.py:271-304 and the sampled losses,
// lightning.py:164-212): embedding lookup, every block, the last LayerNorm, the sampled loss, the whole backward pass and the Adam step.
//
// Why: the block executors (rt_block.hip) left ~22 ctypes calls, 18 torch.empty and five autograd nodes per C2 step to Python —
// 1.0 ms of host time per step (0.5 ms of it inside the autograd engine's backward: scripts/host_profile.py), which is the step's
// wall time on the pool's slower hosts (1.45 ms of host issue against 1.39 ms of device time).  This file issues the SAME entry points
// in the SAME order on the same two streams (the caller's and the library's weight-gradient side stream) from compiled code; Python
// hands over one descriptor and one arena.  `lightning.NativeSasrecStep` decides when a model is the stock one this sequence
// restates; everything else keeps the autograd path, which is also this file's cross-check (tests/test_native_step_gpu.py: parameters
// equal after N steps to the run-to-run noise of either path).
//
// Arena: [parameter gradients, fixed layout (independent of the batch's row count)] [activations and workspaces of `rows` rows].
// Nothing is allocated or freed here; every buffer of a step is dead when the step's Adam launch has been issued on `stream`
// (rt_side_join in front of it), so the next step reuses the arena.
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include "../../include/rectools_hip.h"      // (before rt_common.h: its status macros shadow the header's enum of the same values)
#include "rt_common.h"

namespace {

inline size_t al(size_t floats) { return (floats + 63) & ~(size_t)63; }      // 256-byte aligned regions
inline size_t alb(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

#define RT_TRY(call)                 \
  do {                               \
    const int rc__ = (call);         \
    if (rc__ != RT_OK) return rc__;  \
  } while (0)

constexpr int MAX_BLOCKS = 16;

struct Layout {
  // gradients (floats from the arena's start)
  size_t d_table, d_pos, d_lnf_w, d_lnf_b, d_blk[MAX_BLOCKS], grads_end;
  int64_t blk_off[13];
  // activations / workspaces (BYTES from the arena's start)
  size_t x[MAX_BLOCKS + 1], saved[MAX_BLOCKS], y, mean, rstd, logits, loss_pos, du, loss_ws, emb_ws, g[MAX_BLOCKS + 1], ln_ws,
      scratch[MAX_BLOCKS], total;
  size_t loss_ws_bytes, emb_ws_bytes, ln_ws_bytes, scratch_bytes;
};

bool layout_of(const rt_sasrec_step& s, Layout& L) {
  if (s.n_blocks < 1 || s.n_blocks > MAX_BLOCKS || s.rows <= 0 || s.d <= 0 || s.dff <= 0 || s.V <= 0) return false;
  const size_t M = (size_t)s.rows, d = (size_t)s.d;
  size_t f = 0;
  L.d_table = f; f += al((size_t)s.V * d);
  L.d_pos = f; f += s.pos != nullptr ? al((size_t)s.pos_rows * d) : 0;
  L.d_lnf_w = f; f += al(d);
  L.d_lnf_b = f; f += al(d);
  rt_sasrec_block_grad_offsets(s.d, s.dff, L.blk_off);
  for (int b = 0; b < s.n_blocks; ++b) { L.d_blk[b] = f; f += al((size_t)L.blk_off[12]); }
  L.grads_end = f;
  size_t by = f * 4;
  auto take = [&](size_t bytes) { const size_t at = by; by += alb(bytes); return at; };
  for (int b = 0; b <= s.n_blocks; ++b) L.x[b] = take(M * d * 4);
  const size_t saved_fl = rt_sasrec_block_saved_floats(s.rows, s.d, s.dff, s.H, s.p_blk > 0.f ? 1 : 0);
  for (int b = 0; b < s.n_blocks; ++b) L.saved[b] = take(saved_fl * 4);
  L.y = take(M * d * 4);
  L.mean = take(M * 4);
  L.rstd = take(M * 4);
  L.logits = take(M * (size_t)(s.n_neg + 1) * 4);
  L.loss_pos = take(M * 4);
  L.du = take(M * d * 4);
  L.loss_ws_bytes = rt_sampled_loss_bwd_workspace_bytes(s.rows, s.n_neg, s.V, s.d);
  L.loss_ws = take(L.loss_ws_bytes);
  L.emb_ws_bytes = rt_embed_bwd_workspace_bytes(s.rows, s.V, s.d);
  L.emb_ws = take(L.emb_ws_bytes);
  // one data-gradient buffer per block boundary: the side stream still reads block b's g_out (its weight-gradient products) while
  // the caller's stream is already writing block b - 1's
  for (int b = 0; b <= s.n_blocks; ++b) L.g[b] = take(M * d * 4);
  L.ln_ws_bytes = rt_layernorm_bwd_workspace_bytes(s.rows, s.d);
  L.ln_ws = take(L.ln_ws_bytes > 4 ? L.ln_ws_bytes : 4);
  L.scratch_bytes = rt_sasrec_block_bwd_scratch_bytes(s.rows, s.B_attn, s.d, s.dff, s.H, s.wgrad_splits);
  for (int b = 0; b < s.n_blocks; ++b) L.scratch[b] = take(L.scratch_bytes);
  L.total = by;
  return true;
}

}  // namespace

extern "C" {

size_t rt_sasrec_step_arena_bytes(const rt_sasrec_step* s) {
  Layout L;
  if (s == nullptr || !layout_of(*s, L)) return 0;
  return L.total;
}

// Where the step leaves the gradient of every segment of the flat parameter buffer (seg_role): out [n_seg] device pointers into the arena,
// NULL for a segment without a gradient.  What phase 2 hands to rt_adam_step_segments; a caller that exchanges or inspects gradients
// between the phases (tests against the oracle, a data-parallel pack) reads them here.
int rt_sasrec_step_grad_ptrs(const rt_sasrec_step* sp, const float** out) {
  if (sp == nullptr || out == nullptr) return RT_ERR_INVALID_ARG;
  const rt_sasrec_step& s = *sp;
  Layout L;
  if (!layout_of(s, L)) return RT_ERR_INVALID_ARG;
  if (s.arena == nullptr || s.arena_bytes < L.total || s.n_seg <= 0 || s.n_seg > 1024 || s.seg_role == nullptr) return RT_ERR_INVALID_ARG;
  const float* gbase = static_cast<const float*>(s.arena);
  for (int i = 0; i < s.n_seg; ++i) {
    const int r = s.seg_role[i];
    const float* p = nullptr;
    if (r == 0) p = gbase + L.d_table;
    else if (r == 1) p = s.pos != nullptr ? gbase + L.d_pos : nullptr;
    else if (r == 2) p = gbase + L.d_lnf_w;
    else if (r == 3) p = gbase + L.d_lnf_b;
    else if (r >= 16 && r < 16 + 12 * s.n_blocks) p = gbase + L.d_blk[(r - 16) / 12] + L.blk_off[(r - 16) % 12];
    else if (r != -1) return RT_ERR_INVALID_ARG;
    out[i] = p;      // NULL: a parameter without a gradient is skipped, as torch.optim.Adam does
  }
  return RT_OK;
}

// phase & 1: forward + loss + backward (gradients land in the arena's gradient region; loss_out[0] = the loss);
// phase & 2: rt_side_join + the segmented Adam step over (flat_p, adam_m, adam_v) reading those gradients.
int rt_sasrec_step_run(const rt_sasrec_step* sp, int32_t phase, hipStream_t stream) {
  if (sp == nullptr) return RT_ERR_INVALID_ARG;
  const rt_sasrec_step& s = *sp;
  Layout L;
  if (!layout_of(s, L)) return RT_ERR_INVALID_ARG;
  if (s.arena == nullptr || s.arena_bytes < L.total) return RT_ERR_WORKSPACE;
  if (s.blocks == nullptr || s.ids == nullptr || s.dist == nullptr || s.y == nullptr || s.neg == nullptr || s.yw == nullptr || s.cu == nullptr ||
      s.cu_attn == nullptr || s.table == nullptr || s.lnf_w == nullptr || s.lnf_b == nullptr || s.loss_out == nullptr || s.upstream == nullptr)
    return RT_ERR_INVALID_ARG;
  if (s.pos != nullptr && s.pos_rows != s.window) return RT_ERR_UNSUPPORTED;      // (a longer positional table needs its gradient zeroed first)
  char* base = static_cast<char*>(s.arena);
  float* gbase = static_cast<float*>(s.arena);
  auto F = [&](size_t byte_off) { return reinterpret_cast<float*>(base + byte_off); };
  const int M = s.rows, d = s.d, nb = s.n_blocks;
  float* d_table = gbase + L.d_table;
  float* d_pos = s.pos != nullptr ? gbase + L.d_pos : nullptr;

  if (phase & 1) {
    // ---- forward ---------------------------------------------------------------------------------------------------------------
    RT_TRY(rt_embed_packed_fwd(s.ids, s.dist, s.table, s.pos, s.emb_scale, M, d, s.p_emb, s.seed_emb, s.sid_emb, F(L.x[0]), stream));
    {   // the counting sort of the rows by id (the lookup's backward reads it): a function of the ids alone, on the side stream now
      void* side = nullptr;
      RT_TRY(rt_side_fork(stream, &side));
      RT_TRY(rt_embed_bwd_prepare(s.ids, M, d, s.V, base + L.emb_ws, L.emb_ws_bytes, side != nullptr ? static_cast<hipStream_t>(side) : stream));
    }
    if (s.planes != nullptr) RT_TRY(rt_split_planes(s.planes_src, s.planes_n, s.planes, s.planes_stride, stream));
    rt_sasrec_block blk[MAX_BLOCKS];
    for (int b = 0; b < nb; ++b) {
      blk[b] = s.blocks[b];
      blk[b].rows = M; blk[b].rows_real = s.rows_real; blk[b].B = s.B_attn; blk[b].H = s.H; blk[b].d = d; blk[b].dff = s.dff;
      blk[b].window = s.window; blk[b].pad_keys = s.pad_keys; blk[b].p_drop = s.p_blk; blk[b].cu = s.cu_attn;
      RT_TRY(rt_sasrec_block_packed_fwd(&blk[b], F(L.x[b]), F(L.saved[b]), F(L.x[b + 1]), stream));
    }
    RT_TRY(rt_layernorm_fwd(F(L.x[nb]), s.lnf_w, s.lnf_b, s.eps_last, M, d, F(L.y), F(L.mean), F(L.rstd), stream));
    RT_TRY(rt_sampled_loss_fwd_train(F(L.y), d, s.table, s.y, s.neg, s.yw, M, s.n_neg, d, s.V, s.loss, s.cosine, s.logits_t, s.gbce_beta,
                                     F(L.logits), F(L.loss_pos), F(L.du), d, base + L.loss_ws, L.loss_ws_bytes, 0, stream));
    RT_TRY(rt_loss_reduce(F(L.loss_pos), s.y, M, s.loss == 2 ? 0 : 1, s.loss_out, stream));

    // ---- backward --------------------------------------------------------------------------------------------------------------
    {   // the loss's table half (read by the lookup's backward and the optimiser) on the side stream; its session half is a scaled copy of
        // the unit gradient the forward left — the last LayerNorm's backward takes the unit gradient itself and scales on load
      void* side = nullptr;
      RT_TRY(rt_side_fork(stream, &side));
      RT_TRY(rt_sampled_loss_bwd(F(L.y), d, s.table, s.y, s.neg, M, s.n_neg, d, s.V, s.cosine, s.logits_t, F(L.logits), s.loss_out + 1, 1.0f,
                                 s.upstream, F(L.du), d, nullptr, d, d_table, base + L.loss_ws, L.loss_ws_bytes, 0,
                                 side != nullptr ? static_cast<hipStream_t>(side) : stream));
      if (side != nullptr) RT_TRY(rt_side_mark());      // the lookup's backward waits for THIS point, not for the weight gradients queued behind it
      // (Tried: this half on a third library stream, so that its ~310 us chain does not sit in front of the last block's weight gradients:
      //  + 0.3 % on the C2 step — and a fourth hardware queue in the process, which cost the LATER loops of the same process 20 - 25 %
      //  (HSTU 22.8 -> 17.7 k, eSASRec 13.8 -> 10.9 k seqs/s in the default bench line: their side / prefetch streams then share queues).)
    }
    {   // the last LayerNorm: rows here, the combine of dw / db (optimiser only) on the side stream
      RT_TRY(rt_layernorm_bwd_rows_scaled(F(L.du), s.loss_out + 1, 1.0f, s.upstream, F(L.x[nb]), s.lnf_w, F(L.mean), F(L.rstd), M, d, F(L.g[nb]),
                                          base + L.ln_ws, L.ln_ws_bytes, stream));
      void* side = nullptr;
      RT_TRY(rt_side_fork(stream, &side));
      RT_TRY(rt_layernorm_bwd_combine(base + L.ln_ws, L.ln_ws_bytes, M, d, gbase + L.d_lnf_w, gbase + L.d_lnf_b,
                                      side != nullptr ? static_cast<hipStream_t>(side) : stream));
    }
    for (int b = nb - 1; b >= 0; --b)
      RT_TRY(rt_sasrec_block_packed_bwd(&blk[b], F(L.x[b]), F(L.saved[b]), F(L.g[b + 1]), F(L.g[b]), gbase + L.d_blk[b], base + L.scratch[b],
                                        L.scratch_bytes, s.wgrad_splits, 1, stream));
    RT_TRY(rt_side_wait_mark(stream));
    RT_TRY(rt_embed_packed_bwd(s.ids, s.cu, s.B, F(L.g[0]), s.emb_scale, M, s.window, d, s.V, s.p_emb, s.seed_emb, s.sid_emb, d_table, 1, d_pos,
                               base + L.emb_ws, L.emb_ws_bytes, 1, stream));
  }

  if (phase & 2) {
    if (s.n_seg <= 0 || s.n_seg > 1024 || s.seg_offsets == nullptr || s.seg_lens == nullptr || s.seg_role == nullptr || s.flat_p == nullptr ||
        s.adam_m == nullptr || s.adam_v == nullptr)
      return RT_ERR_INVALID_ARG;
    RT_TRY(rt_side_join(stream));
    const float* ptrs[1024];
    RT_TRY(rt_sasrec_step_grad_ptrs(sp, ptrs));
    RT_TRY(rt_adam_step_segments(s.flat_p, s.adam_m, s.adam_v, s.n_seg, s.seg_offsets, s.seg_lens, ptrs, s.adam_step, s.lr, s.beta1, s.beta2,
                                 s.adam_eps, 1.0f, stream));
  }
  return RT_OK;
}

}  // extern "C"
