// Row-wise / element-wise kernels of the transformer path (all fp32, HBM-bound; one wave per row or
// float4 grid-stride streams).  Reference call sites:
//   K2  embed + inverse positions + dropout      torch_backbone.py:245-247, net_blocks.py:388-399, item_net.py:280
//   K3  LayerNorm                                sasrec.py:221,226,303; net_blocks.py:247,257; ligr.py:90,102; hstu.py:256,291
//   K13 Adam                                     lightning.py:214-218 (torch.optim.Adam, betas (0.9,0.98), eps 1e-8)
//   dropout / activations / gates / row masks    net_blocks.py:63-64,108-109; ligr.py:99-105; sasrec.py:228,300; hstu.py:257,291
// Dropout masks are counter-based (rt_common.h rt_drop4: a hash of (seed, stream, element/4)): the backward kernels regenerate
// the forward mask from the same (seed, stream) pair instead of storing it.
#include "rt_common.h"
#include "rt_scan.h"

namespace {

__device__ __forceinline__ f32x4 drop4(f32x4 v, unsigned long long seed, unsigned long long stream,
                                        unsigned long long idx4, float p, float inv_keep) {
  return rt_drop4(v, seed, stream, idx4, p, inv_keep);      // rt_common.h: 16 hash bits per element
}

// ---------------------------------------------------------------------------------------------------
// K2 embed: out[m,:] = drop( table[ids[m]] * scale + pos[L-1-(m % L)] )
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                                        const float* __restrict__ pos, float scale, int M, int L, int d,
                                                        float p, unsigned long long seed, unsigned long long stream,
                                                        float* __restrict__ out, const long long* __restrict__ dist = nullptr) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long long id = ids[m];
  const int l = m % L;
  const float* trow = table + id * (long long)d;
  // positional row: by the slot of the padded window, or (packed rows) by the row's distance from its session's end
  const float* prow = pos ? pos + (dist ? dist[m] : (long long)(L - 1 - l)) * d : nullptr;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(trow + c);
    v *= scale;
    if (prow) v += *reinterpret_cast<const f32x4*>(prow + c);
    if (p > 0.f) v = drop4(v, seed, stream, ((unsigned long long)m * d + c) >> 2, p, inv_keep);
    *reinterpret_cast<f32x4*>(out + (long long)m * d + c) = v;
  }
}

// Backward of K2 without float atomics (device-scope atomics on gfx950 resolve memory-side and made the old scatter
// 5x slower than the data movement): positions are counting-sorted by item id, one wave reduces each table row
// (rows hit by more than EMB_HEAVY_T positions — popularity skew — are cut into chunks, each reduced by its own
// workgroup into a slab row that the row's wave then combines), and the positional
// gradient is a strided column sum with one workgroup per position.  g = dropped gout (mask regenerated).
//   gtable[id] = scale * sum_{m: ids[m] == id} g[m]   (row 0 = padding_idx stays 0: item_net.py:260-264)
//   gpos[L-1-l] = sum_b g[b*L + l]
constexpr int EMB_HEAVY_T = 128, EMB_HEAVY_CH = 128, EMB_HEAVY_WAVES = 4;   // popular rows are cut into 128-position chunks (rt_scan.h)
inline long long emb_chunk_cap(long long M) { return M / EMB_HEAVY_CH + M / EMB_HEAVY_T + 2; }

struct EmbBwdArgs {
  const long long* ids; const float* gout; float scale; int M, L, d, V; float p;
  unsigned long long seed, stream;
  float* gtable; float* gpos;
  int accumulate;   // 1: gtable already holds another gradient of the same table: rows with positions are ADDED to, others left alone
  const long long* cu; int B;   // packed rows (cu != nullptr): session b = rows cu[b] .. cu[b+1]-1, positional row = distance from its end
  int* count; int* offsets; int* cursor; int* blocksum; int* order; int* rank; int* heavy_count; int* heavy_ids; int* heavy_chunk;
  float* slab;   // [chunks][d] partial rows of the popular ids
};

__global__ __launch_bounds__(256) void embed_order_kernel(EmbBwdArgs a) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  const long long id = a.ids[m];
  if (id != 0) a.order[a.offsets[id] + a.rank[m]] = m;
}

template <int NDV, int U>
__device__ __forceinline__ void embed_accumulate_u(const EmbBwdArgs& a, int& k, int end, int lane, f32x4 (&acc)[NDV]) {
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  for (; k + U <= end; k += U) {
    int mm[U]; f32x4 v[U][NDV];
#pragma unroll
    for (int u = 0; u < U; ++u) mm[u] = a.order[k + u];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[u][i] = (c < a.d) ? *reinterpret_cast<const f32x4*>(a.gout + (long long)mm[u] * a.d + c) : z;
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (a.p > 0.f && c < a.d) v[u][i] = drop4(v[u][i], a.seed, a.stream, ((unsigned long long)mm[u] * a.d + c) >> 2, a.p, inv_keep);
        acc[i] += v[u][i];
      }
  }
}
// a wave's serial chain is ~1-2 us per round of gathers: 16 rows per round, then 4, then 1
template <int NDV>
__device__ __forceinline__ void embed_accumulate(const EmbBwdArgs& a, int beg, int end, int lane, f32x4 (&acc)[NDV]) {
  int k = beg;
  embed_accumulate_u<NDV, (NDV == 1 ? 16 : 8)>(a, k, end, lane, acc);
  embed_accumulate_u<NDV, 4>(a, k, end, lane, acc);
  embed_accumulate_u<NDV, 1>(a, k, end, lane, acc);
}

template <int NDV>
__device__ __forceinline__ void embed_bwd_rows_body(const EmbBwdArgs& a, int id) {
  const int lane = threadIdx.x & 63;
  if (id >= a.V) return;
  const int beg = a.offsets[id], end = a.offsets[id + 1];
  if (a.accumulate && beg == end) return;   // untouched row of an accumulation target: nothing to add, nothing to write
  f32x4 acc[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int slot = a.cursor[id];
  if (slot >= 0) {   // popular id: combine the partial rows of its chunks (fixed order)
    const int n_ch = (end - beg + EMB_HEAVY_CH - 1) / EMB_HEAVY_CH;
    for (int c = 0; c < n_ch; ++c)
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int col = lane * 4 + 256 * i;
        if (col < a.d) acc[i] += *reinterpret_cast<const f32x4*>(a.slab + (long long)(slot + c) * a.d + col);
      }
  } else {
    embed_accumulate<NDV>(a, beg, end, lane, acc);
  }
#pragma unroll
  for (int i = 0; i < NDV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < a.d) {
      f32x4* dst = reinterpret_cast<f32x4*>(a.gtable + (long long)id * a.d + c);
      *dst = a.accumulate ? *dst + acc[i] * a.scale : acc[i] * a.scale;   // one wave owns the row: no race
    }
  }
}

template <int NDV>
__global__ __launch_bounds__(256) void embed_bwd_rows_kernel(EmbBwdArgs a) {
  embed_bwd_rows_body<NDV>(a, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// one 4-wave workgroup per chunk of a popular id: 32 positions per wave, LDS combine, partial row -> slab
template <int NDV>
__global__ __launch_bounds__(EMB_HEAVY_WAVES * 64) void embed_bwd_heavy_kernel(EmbBwdArgs a) {
  __shared__ f32x4 s_part[EMB_HEAVY_WAVES][NDV][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_chunks = *a.heavy_count;
  for (int h = blockIdx.x; h < n_chunks; h += gridDim.x) {
    const int id = a.heavy_ids[h];
    const int beg = a.offsets[id] + a.heavy_chunk[h] * EMB_HEAVY_CH;
    const int end = min(beg + EMB_HEAVY_CH, a.offsets[id + 1]);
    const int per = EMB_HEAVY_CH / EMB_HEAVY_WAVES;
    const int wb = min(beg + wave * per, end), we = min(wb + per, end);
    f32x4 acc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    embed_accumulate<NDV>(a, wb, we, lane, acc);
#pragma unroll
    for (int i = 0; i < NDV; ++i) s_part[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < a.d)
          *reinterpret_cast<f32x4*>(a.slab + (long long)h * a.d + c) =
              (s_part[0][i][lane] + s_part[1][i][lane]) + (s_part[2][i][lane] + s_part[3][i][lane]);
      }
    }
    __syncthreads();
  }
}

// one workgroup per position l: 4 waves stride the batch, lanes own float4 columns; fixed-order LDS combine
template <int NDV>
__device__ __forceinline__ void embed_bwd_pos_body(const EmbBwdArgs& a, int l, f32x4 (&s_part)[4][NDV][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = a.cu ? a.B : a.M / a.L;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  f32x4 acc[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int b = wave; b < B; b += 4) {
    long long m = (long long)b * a.L + l;
    if (a.cu) {   // packed: the row at distance (L-1-l) from the end of session b, if the session is that long
      const long long e = a.cu[b + 1];
      m = e - 1 - (a.L - 1 - l);
      if (m < a.cu[b]) continue;
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < a.d) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.gout + m * a.d + c);
        if (a.p > 0.f) v = drop4(v, a.seed, a.stream, ((unsigned long long)m * a.d + c) >> 2, a.p, inv_keep);
        acc[i] += v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NDV; ++i) s_part[wave][i][lane] = acc[i];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < a.d)
        *reinterpret_cast<f32x4*>(a.gpos + (long long)(a.L - 1 - l) * a.d + c) =
            (s_part[0][i][lane] + s_part[1][i][lane]) + (s_part[2][i][lane] + s_part[3][i][lane]);
    }
  }
}

// The row reductions (one wave per table row) with the positional rows' reduction riding in front: the first `pos_blocks` workgroups
// of the launch run `embed_bwd_pos_body` — 200 light workgroups that took 18 us as a launch of their own at the end of a training step,
// beside nothing; here the 6,687 row workgroups fill the machine around them.  Independent outputs (gpos / gtable).
template <int NDV>
__global__ __launch_bounds__(256) void embed_bwd_rows_pos_kernel(EmbBwdArgs a, int pos_blocks) {
  __shared__ f32x4 s_part[4][NDV][64];
  if ((int)blockIdx.x < pos_blocks) { embed_bwd_pos_body<NDV>(a, blockIdx.x, s_part); return; }
  embed_bwd_rows_body<NDV>(a, ((int)blockIdx.x - pos_blocks) * 4 + (threadIdx.x >> 6));
}

// the counting sort of the rows by id (depends on the ids alone: rt_embed_bwd_prepare runs it ahead of the backward pass)
int launch_embed_order(const EmbBwdArgs& a, hipStream_t stream) {
  const int n = a.V + 1;
  RT_CHECK_HIP(hipMemsetAsync(a.count, 0, sizeof(int) * (((size_t)n + 1 + 15) & ~(size_t)15), stream));   // + heavy_count (+ the pad)
  agg_rank_kernel<<<(a.M + AGG_T - 1) / AGG_T, AGG_T, 0, stream>>>(a.ids, a.M, a.count, a.rank, 1);
  RT_CHECK_LAUNCH();
  {
    const int rc = exclusive_scan_counts(a.count, n, a.offsets, a.cursor, a.blocksum, stream, EMB_HEAVY_T, EMB_HEAVY_CH,
                                         a.heavy_count, a.heavy_ids, a.heavy_chunk);
    if (rc != RT_OK) return rc;
  }
  embed_order_kernel<<<(a.M + 255) / 256, 256, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
template <int NDV>
int launch_embed_bwd(const EmbBwdArgs& a, bool prepared, hipStream_t stream) {
  if (!prepared) {
    const int rc = launch_embed_order(a, stream);
    if (rc != RT_OK) return rc;
  }
  embed_bwd_heavy_kernel<NDV><<<(int)min(emb_chunk_cap(a.M), (long long)rt_num_cus() * 8), EMB_HEAVY_WAVES * 64, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  if (a.gpos) embed_bwd_rows_pos_kernel<NDV><<<(a.V + 3) / 4 + a.L, 256, 0, stream>>>(a, a.L);
  else embed_bwd_rows_kernel<NDV><<<(a.V + 3) / 4, 256, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// ---------------------------------------------------------------------------------------------------
// K3 LayerNorm (one wave per row).  Optional fused multiplier: y = LN(x) * mul (HSTU: u * LN(attn), hstu.py:291)
// ---------------------------------------------------------------------------------------------------
// ids (nullable): the row mask `x * (ids != 0)` applied to the INPUT on the fly (`seqs *= timeline_mask` in front of a LayerNorm,
// sasrec.py:300 / :313); x0 (nullable) receives the masked rows the backward pass needs.
// COLS: only the columns c with (c % grp) < grp_real exist (a model whose width / head size the kernels cannot tile runs on rows padded
// with zero columns, head by head: rt_layernorm_*_cols) — statistics over the real columns only, zeros written to the others.
__device__ __forceinline__ f32x4 ln_keep_cols(f32x4 v, int c, int grp, int grp_real) {
  const int r = c % grp;      // grp % 4 == 0: the four columns of a float4 lie in one group
  v[0] = r < grp_real ? v[0] : 0.f;      v[1] = r + 1 < grp_real ? v[1] : 0.f;
  v[2] = r + 2 < grp_real ? v[2] : 0.f;  v[3] = r + 3 < grp_real ? v[3] : 0.f;
  return v;
}
template <bool COLS = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, int M, int d,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, const long long* __restrict__ ids = nullptr,
                                                            float* __restrict__ x0 = nullptr, int grp = 0, int grp_real = 0) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (long long)m * d;
  const int n_cols = COLS ? d / grp * grp_real : d;
  // a SELECT, not a multiply by 0: a non-finite value in a padded row must become an exact zero (as rt_mul_mask and the backward's
  // mask_dx do), not NaN — SASRec's causal-only attention lets real queries see those rows as keys
  const bool pad = ids != nullptr && ids[m] == 0;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = pad ? zero4 : *reinterpret_cast<const f32x4*>(xr + c);
    if (COLS) v = ln_keep_cols(v, c, grp, grp_real);
    if (x0 != nullptr) *reinterpret_cast<f32x4*>(x0 + (long long)m * d + c) = v;
    s += v[0] + v[1] + v[2] + v[3];
  }
  const float mu = wave_sum(s) / n_cols;
  float q = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = pad ? zero4 : *reinterpret_cast<const f32x4*>(xr + c);
    v -= mu;
    if (COLS) v = ln_keep_cols(v, c, grp, grp_real);
    q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / n_cols + eps);
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = pad ? zero4 : *reinterpret_cast<const f32x4*>(xr + c);
    f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
    f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
    v = (v - mu) * rs * ww + bb;
    if (COLS) v = ln_keep_cols(v, c, grp, grp_real);
    *reinterpret_cast<f32x4*>(y + (long long)m * d + c) = v;
  }
  if (lane == 0) { mean[m] = mu; rstd[m] = rs; }
}

struct DyScale { const float* norm = nullptr; const float* upstream = nullptr; float gscale = 1.f; };

// dx = rstd * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat));  dw += sum dy*xhat;  db += sum dy
// Each wave keeps the dw/db partial sums of its columns in registers over its rows (NDV float4 per lane,
// d <= 256*NDV); the 4 waves are combined through LDS and one atomicAdd per column leaves the block.
// Optional fusions of the passes that surround a LayerNorm in the backward of a transformer block (`res`, `ids` nullable):
//   mask_dy: rows with ids[row] == 0 take dy = 0 (the forward multiplied the LayerNorm OUTPUT by the row mask)
//   res:     dx += res (the LayerNorm input also fed a skip connection)       mask_dx: rows with ids[row] == 0 get dx = 0
template <int NDV, bool COLS = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int M, int d, int rows_per_block,
                                                            float* __restrict__ dx, float* __restrict__ partial,
                                                            const float* __restrict__ res = nullptr,
                                                            const long long* __restrict__ ids = nullptr, int mask_dy = 0,
                                                            int mask_dx = 0, int grp = 0, int grp_real = 0, DyScale dsc = DyScale{}) {
  const int n_cols = COLS ? d / grp * grp_real : d;
  // dy handed over UNSCALED (rt_layernorm_bwd_rows_scaled): dy * gscale * upstream[0] / norm[0] on load — the sampled loss's
  // scale_rows_kernel, expression for expression, without its launch and without the round trip of the scaled rows through memory
  const bool scaled = dsc.norm != nullptr;
  const float dy_sc = scaled ? dsc.gscale * (dsc.upstream != nullptr ? dsc.upstream[0] : 1.f) / dsc.norm[0] : 1.f;
  extern __shared__ float red[];  // [4 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  f32x4 pw[NDV], pb[NDV], wv[NDV];
#pragma unroll
  for (int i = 0; i < NDV; ++i) {
    const int c = lane * 4 + 256 * i;
    pw[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[i] = pw[i];
    wv[i] = (c < d) ? *reinterpret_cast<const f32x4*>(w + c) : pw[i];
  }
  // two rows per iteration: their loads are issued together (the row reductions in between are latency chains)
  for (int m = r0 + wave; m < r1; m += 8) {
    const int m2 = m + 4;
    const bool has2 = m2 < r1;
    const int mb = has2 ? m2 : m;
    const float mu[2] = {mean[m], mean[mb]}, rs[2] = {rstd[m], rstd[mb]};
    const long long ro[2] = {(long long)m * d, (long long)mb * d};
    const bool pad[2] = {ids != nullptr && ids[m] == 0, ids != nullptr && ids[mb] == 0};
    f32x4 g[2][NDV], xh[2][NDV];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        g[u][i] = (c < d && !(mask_dy && pad[u])) ? *reinterpret_cast<const f32x4*>(dy + ro[u] + c) : z;
        if (scaled) g[u][i] = g[u][i] * dy_sc;
        xh[u][i] = (c < d) ? *reinterpret_cast<const f32x4*>(x + ro[u] + c) : z;
        if (COLS && c < d) g[u][i] = ln_keep_cols(g[u][i], c, grp, grp_real);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        xh[u][i] = (c < d) ? (xh[u][i] - mu[u]) * rs[u] : z;
        if (COLS && c < d) xh[u][i] = ln_keep_cols(xh[u][i], c, grp, grp_real);
        f32x4 gw = g[u][i] * wv[i];
        s1[u] += gw[0] + gw[1] + gw[2] + gw[3];
        s2[u] += gw[0] * xh[u][i][0] + gw[1] * xh[u][i][1] + gw[2] * xh[u][i][2] + gw[3] * xh[u][i][3];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) { s1[u] = wave_sum(s1[u]) / n_cols; s2[u] = wave_sum(s2[u]) / n_cols; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !has2) break;
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < d) {
          f32x4 o = (g[u][i] * wv[i] - s1[u] - xh[u][i] * s2[u]) * rs[u];
          if (res != nullptr) o += *reinterpret_cast<const f32x4*>(res + ro[u] + c);
          if (COLS) o = ln_keep_cols(o, c, grp, grp_real);
          if (mask_dx && pad[u]) o = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(dx + ro[u] + c) = o;
          pw[i] += g[u][i] * xh[u][i];
          pb[i] += g[u][i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NDV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < d) {
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * d + c) = pw[i];
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * d + c) = pb[i];
    }
  }
  __syncthreads();
  // partials [blocks][2][d]; layernorm_bwd_reduce_kernel sums them in a fixed order (same-address float atomics from
  // hundreds of blocks serialise memory-side and cost more than the whole streaming pass)
  float* part = partial + (long long)blockIdx.x * 2 * d;
  for (int c = threadIdx.x; c < d; c += 256) {
    part[c] = (red[0 * d + c] + red[2 * d + c]) + (red[4 * d + c] + red[6 * d + c]);
    part[d + c] = (red[1 * d + c] + red[3 * d + c]) + (red[5 * d + c] + red[7 * d + c]);
  }
}

// dw[c] = sum_b partial[b][0][c], db[c] = sum_b partial[b][1][c]; block = 16 columns x 16 row phases (short chains)
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int blocks, int d,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;   // column in [0, 2d): dw then db
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * d) {
    int b = ph;
    for (; b + 48 < blocks; b += 64) {
      s0 += partial[(long long)b * 2 * d + c];         s1 += partial[(long long)(b + 16) * 2 * d + c];
      s2 += partial[(long long)(b + 32) * 2 * d + c];  s3 += partial[(long long)(b + 48) * 2 * d + c];
    }
    for (; b < blocks; b += 16) s0 += partial[(long long)b * 2 * d + c];
  }
  red[ph][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && c < 2 * d) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    if (c < d) dw[c] = t; else db[c - d] = t;
  }
}

// ---------------------------------------------------------------------------------------------------
// element-wise streams (n4 = number of float4 groups)
// ---------------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_SIGMOID = 4 };

__device__ __forceinline__ float act_f(float z, int kind) {
  switch (kind) {
    case ACT_RELU: return fmaxf(z, 0.f);
    case ACT_GELU: return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
    case ACT_SILU: return z / (1.f + __expf(-z));
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
    default: return z;
  }
}
__device__ __forceinline__ float act_df(float z, int kind) {
  switch (kind) {
    case ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case ACT_GELU: return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
    case ACT_SILU: { float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }
    case ACT_SIGMOID: { float s = 1.f / (1.f + __expf(-z)); return s * (1.f - s); }
    default: return 1.f;
  }
}

// y = drop(act(z)) [+ r]      (dz = drop(dy) * act'(z) with the same seed/stream; the residual passes dy through)
__global__ void act_dropout_fwd_kernel(const f32x4* __restrict__ z, int kind, float p, unsigned long long seed,
                                       unsigned long long stream, long long n4, const f32x4* __restrict__ r,
                                       f32x4* __restrict__ y) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = z[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_f(v[j], kind);
    if (p > 0.f) v = drop4(v, seed, stream, (unsigned long long)i, p, inv_keep);
    if (r != nullptr) v += r[i];
    y[i] = v;
  }
}
__global__ void act_dropout_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ z, int kind, float p,
                                       unsigned long long seed, unsigned long long stream, long long n4,
                                       f32x4* __restrict__ dz) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 g = dy[i];
    if (p > 0.f) g = drop4(g, seed, stream, (unsigned long long)i, p, inv_keep);
    if (kind != ACT_NONE) {
      f32x4 v = z[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] *= act_df(v[j], kind);
    }
    dz[i] = g;
  }
}

// swiglu: y = drop(silu(a) * b)   (net_blocks.py:108)
__global__ void swiglu_fwd_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, float p,
                                  unsigned long long seed, unsigned long long stream, long long n4, f32x4* __restrict__ y) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 va = a[i], vb = b[i], v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_f(va[j], ACT_SILU) * vb[j];
    if (p > 0.f) v = drop4(v, seed, stream, (unsigned long long)i, p, inv_keep);
    y[i] = v;
  }
}
__global__ void swiglu_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                  float p, unsigned long long seed, unsigned long long stream, long long n4,
                                  f32x4* __restrict__ da, f32x4* __restrict__ db) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 g = dy[i];
    if (p > 0.f) g = drop4(g, seed, stream, (unsigned long long)i, p, inv_keep);
    f32x4 va = a[i], vb = b[i], ga, gb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ga[j] = g[j] * vb[j] * act_df(va[j], ACT_SILU);
      gb[j] = g[j] * act_f(va[j], ACT_SILU);
    }
    da[i] = ga; db[i] = gb;
  }
}

// gated residual: y = x + sigmoid(gz) * drop(a)   (ligr.py:99-100, 104-105)
__global__ void gate_fwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ gz, const f32x4* __restrict__ a,
                                float p, unsigned long long seed, unsigned long long stream, long long n4,
                                f32x4* __restrict__ y) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 va = a[i];
    if (p > 0.f) va = drop4(va, seed, stream, (unsigned long long)i, p, inv_keep);
    f32x4 vx = x[i], vg = gz[i], v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = vx[j] + act_f(vg[j], ACT_SIGMOID) * va[j];
    y[i] = v;
  }
}
// dgz = dy * drop(a) * s(1-s);  da = drop(dy * s)   (dx = dy is returned by the caller as-is)
__global__ void gate_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ gz, const f32x4* __restrict__ a,
                                float p, unsigned long long seed, unsigned long long stream, long long n4,
                                f32x4* __restrict__ dgz, f32x4* __restrict__ da) {
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 g = dy[i], vg = gz[i], va = a[i], o1, o2;
    if (p > 0.f) va = drop4(va, seed, stream, (unsigned long long)i, p, inv_keep);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = act_f(vg[j], ACT_SIGMOID);
      o1[j] = g[j] * va[j] * s * (1.f - s);
      o2[j] = g[j] * s;
    }
    if (p > 0.f) o2 = drop4(o2, seed, stream, (unsigned long long)i, p, inv_keep);
    dgz[i] = o1; da[i] = o2;
  }
}

// y = a*alpha + b (residual add); b may be null
__global__ void axpy_kernel(const f32x4* __restrict__ a, float alpha, const f32x4* __restrict__ b, long long n4,
                            f32x4* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = a[i] * alpha;
    if (b) v += b[i];
    y[i] = v;
  }
}
// y = a * b (* rowmask)
__global__ void mul_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const long long* __restrict__ ids,
                           int d4, long long n4, f32x4* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = a[i];
    if (b) v *= b[i];
    if (ids && ids[i / d4] == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
    y[i] = v;
  }
}

// y[r, :] = a[r, :] * b[r, :] * (ids[r] != 0) with row strides (operands / results that are column slices of packed buffers)
__global__ void mul_ld_kernel(const float* __restrict__ a, long long lda, const float* __restrict__ b, long long ldb,
                              const long long* __restrict__ ids, int d4, long long n4, float* __restrict__ y, long long ldy) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / d4;
    const int c = (int)(i - r * d4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a + r * lda + c);
    if (b) v *= *reinterpret_cast<const f32x4*>(b + r * ldb + c);
    if (ids && ids[r] == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(y + r * ldy + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// K13 Adam on flat buffers
// ---------------------------------------------------------------------------------------------------
__global__ void adam_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                            long long n4, float lr_bc1, float inv_sqrt_bc2, float b1, float b2, float eps,
                            float grad_scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 gg = g[i] * grad_scale, mm = m[i], vv = v[i], pp = p[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
      vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= lr_bc1 * (mm[j] / denom);
    }
    m[i] = mm; v[i] = vv; p[i] = pp;
  }
}

// Segmented variant: parameters / moments flat, every segment's gradient behind its own pointer (autograd's own
// output tensors: no AccumulateGrad add into a flat gradient buffer, no zero fill).  Block = one 4 KiB-float4 chunk
// of one segment; the chunk table travels in the kernel arguments.
constexpr int ADAM_SEGS = 48, ADAM_CHUNK4 = 1024;
struct AdamSegs {
  const float* grad[ADAM_SEGS];
  long long ofs4[ADAM_SEGS];     // segment start in the flat buffers (float4 units)
  long long len[ADAM_SEGS];      // floats (any length; a ragged tail is handled element-wise)
  int first_chunk[ADAM_SEGS + 1];
  int n;
};
__global__ __launch_bounds__(256) void adam_segs_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                        AdamSegs sg, float lr_bc1, float inv_sqrt_bc2, float b1, float b2,
                                                        float eps, float grad_scale) {
  int s = 0;
  while (s + 1 < sg.n && (int)blockIdx.x >= sg.first_chunk[s + 1]) ++s;
  const long long c0 = (long long)(blockIdx.x - sg.first_chunk[s]) * ADAM_CHUNK4;   // float4 index inside the segment
  const float* __restrict__ g = sg.grad[s];
  const long long len = sg.len[s];
  const bool g_vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
#pragma unroll
  for (int u = 0; u < ADAM_CHUNK4 / 256; ++u) {
    const long long i4 = c0 + u * 256 + threadIdx.x;
    const long long e = i4 * 4;
    if (e >= len) break;
    f32x4 gg = {0.f, 0.f, 0.f, 0.f};
    if (g_vec && e + 4 <= len) gg = *reinterpret_cast<const f32x4*>(g + e);
    else for (int j = 0; j < 4; ++j) if (e + j < len) gg[j] = g[e + j];
    gg *= grad_scale;
    const long long f = sg.ofs4[s] + i4;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[f], vv = reinterpret_cast<f32x4*>(v)[f], pp = reinterpret_cast<f32x4*>(p)[f];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
      vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= lr_bc1 * (mm[j] / denom);
    }
    reinterpret_cast<f32x4*>(m)[f] = mm; reinterpret_cast<f32x4*>(v)[f] = vv; reinterpret_cast<f32x4*>(p)[f] = pp;
  }
}

inline int stream_grid(long long n4) {
  long long b = (n4 + 255) / 256;
  long long cap = (long long)rt_num_cus() * 8;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

extern "C" {

int rt_embed_fwd(const int64_t* ids, const float* table, const float* pos, float scale, int32_t M, int32_t L,
                 int32_t d, float p, uint64_t seed, uint64_t stream_id, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) != 0 || L <= 0) return RT_ERR_INVALID_ARG;
  embed_fwd_kernel<<<(M + 3) / 4, 256, 0, stream>>>(reinterpret_cast<const long long*>(ids), table, pos, scale, M, L, d, p,
                                                     seed, stream_id, out);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"

// The FIRST block of recommend() without a product over the rows (round 6).  Its input is x = scale E[id] + P[dist]; SASRec projects keys and
// values from x and queries from LN1(x) (sasrec.py:221-224), all three LINEAR in the two table rows once the row's LayerNorm statistics
// are known:   K | V = scale (E W_kv^T)[id] + (P W_kv^T + b_kv)[dist]
//              Q     = rstd (scale (E G W_q^T)[id] + (P G W_q^T)[dist] - mean (W_q g)) + (W_q beta + b_q)      (G = diag(g), g / beta = LN1's)
// One wave per row: x in registers -> mean / rstd -> q = LN1(x) (the block's skip branch) -> Q and K | V gathered from the PROJECTED tables
// the caller made once per call.  Replaces the embedding kernel, LayerNorm_1, the query projection and the key / value projection
// (8 KB read out of the cached tables + 4 KB written per row at d = 256).
struct EmbedBlock1Args {
  const long long *ids, *dist; const float *E, *P; float scale;
  const float *ln_w, *ln_b; float eps;
  const float *QE, *QP, *wg, *wb, *KVE, *KVP;      // affine tables [V | L, AFF d], [AFF d] x 2; plain tables [V | L, PLN d]
  int M, d; float *q_out, *Q_out, *KV_out, *x_out;  // LN1(x) [M, d] (nullable), affine [M, AFF d], plain [M, PLN d], x [M, d] (nullable)
};
// AFF / PLN: widths of the affine (read LN1(x)) and the plain (read x) projections in units of d — SASRec: Q affine, K | V plain (1, 2);
// Pre-LN / LiGR blocks: Q | K | V all read LN1(x) (3, 0) and the skip branch wants x itself (x_out).
template <int NV, int AFF, int PLN>      // NV float4 per lane: d <= 256 NV
__global__ __launch_bounds__(256) void embed_block1_kernel(EmbedBlock1Args a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.M) return;
  const int d = a.d;
  const long long id = a.ids[m], pd = a.dist[m];
  const float* er = a.E + id * (long long)d;
  const float* pr = a.P + pd * (long long)d;
  const float* qe = a.QE + id * (long long)(AFF * d);
  const float* qp = a.QP + pd * (long long)(AFF * d);
  // every table row of this output row is requested before anything is reduced: 8 loads of 16 bytes in flight per lane at d = 256
  f32x4 x[NV], tq[AFF * NV], tk[PLN * NV > 0 ? PLN * NV : 1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    x[i] = zero4;
    if (c < d) x[i] = *reinterpret_cast<const f32x4*>(er + c) * a.scale + *reinterpret_cast<const f32x4*>(pr + c);
  }
#pragma unroll
  for (int i = 0; i < AFF * NV; ++i) {
    const int c = lane * 4 + 256 * i;
    tq[i] = zero4;
    if (c < AFF * d) tq[i] = *reinterpret_cast<const f32x4*>(qe + c) * a.scale + *reinterpret_cast<const f32x4*>(qp + c);
  }
  if (PLN > 0) {
    const float* ke = a.KVE + id * (long long)(PLN * d);
    const float* kp = a.KVP + pd * (long long)(PLN * d);
#pragma unroll
    for (int i = 0; i < PLN * NV; ++i) {
      const int c = lane * 4 + 256 * i;
      tk[i] = zero4;
      if (c < PLN * d) tk[i] = *reinterpret_cast<const f32x4*>(ke + c) * a.scale + *reinterpret_cast<const f32x4*>(kp + c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += x[i][0] + x[i][1] + x[i][2] + x[i][3];      // (columns behind d hold zeros)
  const float mu = wave_sum(s) / d;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < d) { const f32x4 v = x[i] - mu; q2 += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q2) / d + a.eps);      // (layernorm_fwd_kernel's own two passes)
  if (PLN > 0) {
#pragma unroll
    for (int i = 0; i < PLN * NV; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < PLN * d) *reinterpret_cast<f32x4*>(a.KV_out + (long long)m * PLN * d + c) = tk[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < d) {
      if (a.x_out != nullptr) *reinterpret_cast<f32x4*>(a.x_out + (long long)m * d + c) = x[i];
      if (a.q_out != nullptr) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(a.ln_w + c), bb = *reinterpret_cast<const f32x4*>(a.ln_b + c);
        *reinterpret_cast<f32x4*>(a.q_out + (long long)m * d + c) = (x[i] - mu) * rs * ww + bb;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < AFF * NV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < AFF * d) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.wg + c), b4 = *reinterpret_cast<const f32x4*>(a.wb + c);
      *reinterpret_cast<f32x4*>(a.Q_out + (long long)m * AFF * d + c) = (tq[i] - g4 * mu) * rs + b4;
    }
  }
}

template <int AFF, int PLN>
int launch_embed_block1(const EmbedBlock1Args& a, hipStream_t stream) {
  const int grid = (a.M + 3) / 4;
  if (a.d <= 256) embed_block1_kernel<1, AFF, PLN><<<grid, 256, 0, stream>>>(a);
  else if (a.d <= 512) embed_block1_kernel<2, AFF, PLN><<<grid, 256, 0, stream>>>(a);
  else embed_block1_kernel<4, AFF, PLN><<<grid, 256, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

extern "C" {
int rt_embed_block1_fwd(const int64_t* ids, const int64_t* dist, const float* E, const float* P, float scale, const float* ln_w, const float* ln_b,
                        float eps, const float* QE, const float* QP, const float* wg, const float* wb, const float* KVE, const float* KVP, int32_t M,
                        int32_t d, float* q_out, float* Q_out, float* KV_out, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if (ids == nullptr || dist == nullptr || E == nullptr || P == nullptr || ln_w == nullptr || ln_b == nullptr || QE == nullptr || QP == nullptr ||
      wg == nullptr || wb == nullptr || KVE == nullptr || KVP == nullptr || q_out == nullptr || Q_out == nullptr || KV_out == nullptr || (d & 3) != 0)
    return RT_ERR_INVALID_ARG;
  if (d > 1024) return RT_ERR_UNSUPPORTED;
  EmbedBlock1Args a{reinterpret_cast<const long long*>(ids), reinterpret_cast<const long long*>(dist), E, P, scale, ln_w, ln_b, eps, QE, QP, wg, wb,
                    KVE, KVP, M, d, q_out, Q_out, KV_out, nullptr};
  return launch_embed_block1<1, 2>(a, stream);
}
// The Pre-LN form (net_blocks.py:236-262, ligr.py:161-191: queries, keys AND values read LN1(x), the skip branch reads x): x [M, d] and
// qkv [M, 3d] = rstd (scale QKVE[id] + QKVP[dist] - mean W g) + (W beta + b) from the tables QKVE = (E diag(g)) W^T [V, 3d],
// QKVP = (P diag(g)) W^T [L, 3d], wg = W g, wb = W beta + b (W / b: the packed in_proj parameters).
int rt_embed_block1_preln_fwd(const int64_t* ids, const int64_t* dist, const float* E, const float* P, float scale, float eps, const float* QKVE,
                              const float* QKVP, const float* wg, const float* wb, int32_t M, int32_t d, float* x_out, float* qkv_out,
                              hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if (ids == nullptr || dist == nullptr || E == nullptr || P == nullptr || QKVE == nullptr || QKVP == nullptr || wg == nullptr || wb == nullptr ||
      x_out == nullptr || qkv_out == nullptr || (d & 3) != 0)
    return RT_ERR_INVALID_ARG;
  if (d > 512) return RT_ERR_UNSUPPORTED;      // (3 d / 256 float4 per lane held in registers)
  EmbedBlock1Args a{reinterpret_cast<const long long*>(ids), reinterpret_cast<const long long*>(dist), E, P, scale, nullptr, nullptr, eps, QKVE, QKVP,
                    wg, wb, nullptr, nullptr, M, d, nullptr, qkv_out, nullptr, x_out};
  return launch_embed_block1<3, 0>(a, stream);
}

// Packed rows: as rt_embed_fwd, the positional row of row m is pos[dist[m]] (dist = distance of the row from its
// session's end, int64 [M]); rows with id 0 (the unused tail up to the GEMM tile) read table row 0 and are ignored downstream.
int rt_embed_packed_fwd(const int64_t* ids, const int64_t* dist, const float* table, const float* pos, float scale, int32_t M,
                        int32_t d, float p, uint64_t seed, uint64_t stream_id, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) != 0 || (pos != nullptr && dist == nullptr)) return RT_ERR_INVALID_ARG;
  embed_fwd_kernel<<<(M + 3) / 4, 256, 0, stream>>>(reinterpret_cast<const long long*>(ids), table, pos, scale, M, 1, d, p, seed,
                                                     stream_id, out, reinterpret_cast<const long long*>(dist));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Host arithmetic: bytes of the int scratch rt_embed_bwd needs.
size_t rt_embed_bwd_workspace_bytes(int32_t M, int32_t V, int32_t d) {
  const size_t n = (size_t)V + 1, cap = (size_t)emb_chunk_cap(M);
  return 4 * (cap * ((size_t)d + 2) + 3 * n + 1 + scan_blocks(n) + 64 + 2 * (size_t)M + 4) + 256 + 64;   // (+ the aligned cleared region)
}

}  // extern "C"
namespace {
void carve_embed_workspace(EmbBwdArgs& a, void* workspace, int M, int V, int d) {
  const size_t n = (size_t)V + 1, cap = (size_t)emb_chunk_cap(M);
  a.slab = reinterpret_cast<float*>(workspace);   // first: 16-byte aligned rows
  int* ip = reinterpret_cast<int*>(a.slab + cap * (size_t)d);
  ip = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(ip) + 255) & ~(uintptr_t)255);   // the cleared region: 256-byte aligned, a multiple
  a.count = ip; ip += n;                                                                    // of 64 bytes long — ONE fill kernel (an unaligned
  a.heavy_count = ip; ip += 1;          // directly behind count: one memset clears both    // memset is three: head, body, tail)
  ip += (16 - ((n + 1) & 15)) & 15;
  a.offsets = ip; ip += n;
  a.cursor = ip; ip += n;
  a.blocksum = ip; ip += scan_blocks(n) + 64;
  a.order = ip; ip += M;
  a.rank = ip; ip += M;
  a.heavy_ids = ip; ip += cap;
  a.heavy_chunk = ip;
}
}  // namespace
extern "C" {

// The rows' counting sort by id, ahead of the backward pass (it depends on the ids alone; on another stream it leaves eight small
// launches out of the tail of a training step): fills `workspace`; rt_embed_bwd / rt_embed_packed_bwd called with prepared = 1 on the
// SAME workspace and the same ids start at the row reductions.
int rt_embed_bwd_prepare(const int64_t* ids, int32_t M, int32_t d, int32_t V, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if ((d & 3) != 0 || V <= 0 || M < 0 || d > 1024 || ids == nullptr) return RT_ERR_INVALID_ARG;
  if (workspace == nullptr || workspace_bytes < rt_embed_bwd_workspace_bytes(M, V, d)) return RT_ERR_WORKSPACE;
  EmbBwdArgs a{};
  a.ids = reinterpret_cast<const long long*>(ids); a.M = M; a.d = d; a.V = V;
  carve_embed_workspace(a, workspace, M, V, d);
  return launch_embed_order(a, stream);
}

// gtable [V,d] and gpos [L,d] (optional) are fully overwritten; M must be a multiple of L when gpos is given.
// accumulate = 1: gtable already holds another gradient of the same table (the loss's): the rows that occur in `ids` are
// added to in place and no other row is touched — no second [V,d] tensor, no [V,d] add (3 GB of traffic at V = 1M, d = 256).
int rt_embed_bwd(const int64_t* ids, const float* gout, float scale, int32_t M, int32_t L, int32_t d, int32_t V, float p,
                 uint64_t seed, uint64_t stream_id, float* gtable, int32_t accumulate, float* gpos, void* workspace,
                 size_t workspace_bytes, int32_t prepared, hipStream_t stream) {
  (void)hipGetLastError();
  if ((d & 3) != 0 || L <= 0 || V <= 0 || M < 0 || d > 1024 || (gpos && (M % L) != 0)) return RT_ERR_INVALID_ARG;
  if (workspace == nullptr || workspace_bytes < rt_embed_bwd_workspace_bytes(M, V, d)) return RT_ERR_WORKSPACE;
  EmbBwdArgs a{};
  a.ids = reinterpret_cast<const long long*>(ids); a.gout = gout; a.scale = scale; a.M = M; a.L = L; a.d = d; a.V = V; a.p = p;
  a.seed = seed; a.stream = stream_id; a.gtable = gtable; a.gpos = gpos; a.accumulate = accumulate;
  carve_embed_workspace(a, workspace, M, V, d);
  if (d <= 256) return launch_embed_bwd<1>(a, prepared != 0, stream);
  if (d <= 512) return launch_embed_bwd<2>(a, prepared != 0, stream);
  return launch_embed_bwd<4>(a, prepared != 0, stream);
}

// Backward of rt_embed_packed_fwd: gtable as rt_embed_bwd; gpos [L, d] (optional, fully overwritten): gpos[t] = sum over the
// sessions longer than t of the gradient row at distance t from the session's end (cu_seqlens [B+1], rows of session b =
// cu[b] .. cu[b+1]-1).  Workspace: rt_embed_bwd_workspace_bytes(M, V, d).
int rt_embed_packed_bwd(const int64_t* ids, const int64_t* cu_seqlens, int32_t B, const float* gout, float scale, int32_t M, int32_t L,
                        int32_t d, int32_t V, float p, uint64_t seed, uint64_t stream_id, float* gtable, int32_t accumulate,
                        float* gpos, void* workspace, size_t workspace_bytes, int32_t prepared, hipStream_t stream) {
  (void)hipGetLastError();
  if ((d & 3) != 0 || L <= 0 || V <= 0 || M < 0 || B < 0 || d > 1024 || (gpos && cu_seqlens == nullptr)) return RT_ERR_INVALID_ARG;
  if (workspace == nullptr || workspace_bytes < rt_embed_bwd_workspace_bytes(M, V, d)) return RT_ERR_WORKSPACE;
  EmbBwdArgs a{};
  a.ids = reinterpret_cast<const long long*>(ids); a.gout = gout; a.scale = scale; a.M = M; a.L = L; a.d = d; a.V = V; a.p = p;
  a.seed = seed; a.stream = stream_id; a.gtable = gtable; a.gpos = gpos; a.accumulate = accumulate;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B;
  carve_embed_workspace(a, workspace, M, V, d);
  if (d <= 256) return launch_embed_bwd<1>(a, prepared != 0, stream);
  if (d <= 512) return launch_embed_bwd<2>(a, prepared != 0, stream);
  return launch_embed_bwd<4>(a, prepared != 0, stream);
}

int rt_layernorm_fwd(const float* x, const float* w, const float* b, float eps, int32_t M, int32_t d, float* y,
                     float* mean, float* rstd, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) != 0) return RT_ERR_INVALID_ARG;
  layernorm_fwd_kernel<false><<<(M + 3) / 4, 256, 0, stream>>>(x, w, b, eps, M, d, y, mean, rstd);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// LayerNorm of the row-masked input: y = LN(x * (ids != 0)); x0 (nullable) = the masked input (what the backward reads).
int rt_layernorm_fwd_masked(const float* x, const int64_t* ids, const float* w, const float* b, float eps, int32_t M, int32_t d,
                            float* x0, float* y, float* mean, float* rstd, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) != 0 || ids == nullptr) return RT_ERR_INVALID_ARG;
  layernorm_fwd_kernel<false><<<(M + 3) / 4, 256, 0, stream>>>(x, w, b, eps, M, d, y, mean, rstd, reinterpret_cast<const long long*>(ids), x0);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// LayerNorm over rows that carry ZERO COLUMNS (rt_layernorm_fwd_cols / _bwd_cols, include/rectools_hip.h): column c exists iff
// (c % grp) < grp_real; ids / x0 as rt_layernorm_fwd_masked (both nullable).
static bool bad_cols(int d, int grp, int grp_real) { return (d & 3) != 0 || grp <= 0 || (grp & 3) != 0 || d % grp != 0 || grp_real <= 0 || grp_real > grp; }
int rt_layernorm_fwd_cols(const float* x, const int64_t* ids, const float* w, const float* b, float eps, int32_t M, int32_t d, int32_t grp,
                          int32_t grp_real, float* x0, float* y, float* mean, float* rstd, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if (bad_cols(d, grp, grp_real)) return RT_ERR_INVALID_ARG;
  layernorm_fwd_kernel<true><<<(M + 3) / 4, 256, 0, stream>>>(x, w, b, eps, M, d, y, mean, rstd, reinterpret_cast<const long long*>(ids), x0, grp,
                                                             grp_real);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

static int ln_bwd_blocks(int M, int& rpb) {
  int blocks = rt_num_cus() * 2;
  rpb = (M + blocks - 1) / blocks;
  if (rpb < 8) rpb = 8;
  return (M + rpb - 1) / rpb;
}
// Host arithmetic: bytes of the float scratch rt_layernorm_bwd needs (per-block dw/db partial sums).
size_t rt_layernorm_bwd_workspace_bytes(int32_t M, int32_t d) {
  if (M <= 0) return 0;
  int rpb;
  return (size_t)ln_bwd_blocks(M, rpb) * 2 * (size_t)d * sizeof(float);
}

// dx [M,d], dw [d], db [d] are fully overwritten (deterministic two-stage reduction, no atomics).
int rt_layernorm_bwd_fused(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d,
                           float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream);
int rt_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, int32_t M,
                     int32_t d, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return rt_layernorm_bwd_fused(dy, x, w, mean, rstd, nullptr, nullptr, 0, 0, M, d, dx, dw, db, workspace, workspace_bytes, stream);
}
// Same with the surrounding passes fused: dy rows of padded positions read as zero (mask_dy), dx += res, dx rows of padded
// positions written as zero (mask_dx); res / ids nullable.
static int layernorm_bwd_any(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, int32_t grp,
                           int32_t grp_real, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream,
                           bool combine = true, DyScale dsc = DyScale{});
int rt_layernorm_bwd_fused(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d,
                           float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return layernorm_bwd_any(dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, 0, 0, dx, dw, db, workspace, workspace_bytes, stream);
}
int rt_layernorm_bwd_cols(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                          const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, int32_t grp,
                          int32_t grp_real, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (bad_cols(d, grp, grp_real)) return RT_ERR_INVALID_ARG;
  return layernorm_bwd_any(dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, grp, grp_real, dx, dw, db, workspace, workspace_bytes, stream);
}
static int layernorm_bwd_any(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           const float* res, const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, int32_t grp,
                           int32_t grp_real, float* dx, float* dw, float* db, void* workspace, size_t workspace_bytes, hipStream_t stream, bool combine,
                           DyScale dsc) {
  (void)hipGetLastError();
  if (dsc.norm != nullptr && grp > 0) return RT_ERR_UNSUPPORTED;
  if ((mask_dy || mask_dx) && ids == nullptr) return RT_ERR_INVALID_ARG;
  const long long* idp = reinterpret_cast<const long long*>(ids);
  if ((d & 3) != 0) return RT_ERR_INVALID_ARG;
  if (d > 1024) return RT_ERR_UNSUPPORTED;
  if (M <= 0) {
    RT_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * d, stream));
    RT_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * d, stream));
    return RT_OK;
  }
  if (workspace == nullptr || workspace_bytes < rt_layernorm_bwd_workspace_bytes(M, d)) return RT_ERR_WORKSPACE;
  int rpb;
  const int blocks = ln_bwd_blocks(M, rpb);
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t lds = 8 * (size_t)d * sizeof(float);
  if (grp > 0) {
    if (d <= 256) layernorm_bwd_kernel<1, true><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, grp, grp_real);
    else if (d <= 512) layernorm_bwd_kernel<2, true><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, grp, grp_real);
    else layernorm_bwd_kernel<4, true><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, grp, grp_real);
  } else if (d <= 256) layernorm_bwd_kernel<1><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, 0, 0, dsc);
  else if (d <= 512) layernorm_bwd_kernel<2><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, 0, 0, dsc);
  else layernorm_bwd_kernel<4><<<blocks, 256, lds, stream>>>(dy, x, w, mean, rstd, M, d, rpb, dx, partial, res, idp, mask_dy, mask_dx, 0, 0, dsc);
  RT_CHECK_LAUNCH();
  if (!combine) return RT_OK;      // (the caller combines the partial sums itself: rt_layernorm_bwd_combine, possibly on another stream)
  layernorm_bwd_reduce_kernel<<<(2 * d + 15) / 16, 256, 0, stream>>>(partial, blocks, d, dw, db);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
// The two halves of rt_layernorm_bwd_fused as calls of their own: `rows` writes dx and the per-block partial sums of dw / db into
// `workspace`, `combine` reduces those into dw [d], db [d] (fixed order).  dx is what the backward pass waits for; dw / db are read by
// the optimiser only — a caller may issue `combine` on another stream behind an event (the block executors: the weight-gradient side
// stream, one launch and one kernel boundary less on the critical path per LayerNorm).
int rt_layernorm_bwd_rows(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, const float* res,
                          const int64_t* ids, int32_t mask_dy, int32_t mask_dx, int32_t M, int32_t d, float* dx, void* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
  if (M <= 0) return RT_OK;
  return layernorm_bwd_any(dy, x, w, mean, rstd, res, ids, mask_dy, mask_dx, M, d, 0, 0, dx, nullptr, nullptr, workspace, workspace_bytes, stream,
                           false);
}
// rt_layernorm_bwd_rows over dy * (gscale * upstream[0] / norm[0]): the unit gradient of a sampled loss (rt_sampled_loss_fwd_train's
// d_sess_unit) taken as it is — what rt_sampled_loss_bwd's session half would write first (same bits), without that launch.
int rt_layernorm_bwd_rows_scaled(const float* dy, const float* norm, float gscale, const float* upstream, const float* x, const float* w,
                                 const float* mean, const float* rstd, int32_t M, int32_t d, float* dx, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  if (M <= 0) return RT_OK;
  if (norm == nullptr) return RT_ERR_INVALID_ARG;
  DyScale dsc;
  dsc.norm = norm; dsc.upstream = upstream; dsc.gscale = gscale;
  return layernorm_bwd_any(dy, x, w, mean, rstd, nullptr, nullptr, 0, 0, M, d, 0, 0, dx, nullptr, nullptr, workspace, workspace_bytes, stream, false, dsc);
}
int rt_layernorm_bwd_combine(const void* workspace, size_t workspace_bytes, int32_t M, int32_t d, float* dw, float* db, hipStream_t stream) {
  (void)hipGetLastError();
  if (dw == nullptr || db == nullptr || (d & 3) != 0) return RT_ERR_INVALID_ARG;
  if (M <= 0) {
    RT_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * d, stream));
    RT_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * d, stream));
    return RT_OK;
  }
  if (workspace == nullptr || workspace_bytes < rt_layernorm_bwd_workspace_bytes(M, d)) return RT_ERR_WORKSPACE;
  int rpb;
  const int blocks = ln_bwd_blocks(M, rpb);
  layernorm_bwd_reduce_kernel<<<(2 * d + 15) / 16, 256, 0, stream>>>(reinterpret_cast<const float*>(workspace), blocks, d, dw, db);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// dw[c] = sum_b partial[b][0][c], db[c] = sum_b partial[b][1][c] over `blocks` partials of 2 d floats (fixed order): the second stage of
// the LayerNorm backward for callers that produce the per-block partials themselves (rt_block_tail_bwd).
int rt_layernorm_bwd_reduce(const float* partial, int32_t blocks, int32_t d, float* dw, float* db, hipStream_t stream) {
  (void)hipGetLastError();
  if (partial == nullptr || dw == nullptr || db == nullptr || blocks <= 0 || d <= 0) return RT_ERR_INVALID_ARG;
  layernorm_bwd_reduce_kernel<<<(2 * d + 15) / 16, 256, 0, stream>>>(partial, blocks, d, dw, db);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_act_dropout_fwd(const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                       const float* residual, float* y, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  act_dropout_fwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(z), kind, p, seed, stream_id,
                                                                 n / 4, reinterpret_cast<const f32x4*>(residual),
                                                                 reinterpret_cast<f32x4*>(y));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_act_dropout_bwd(const float* dy, const float* z, int32_t kind, float p, uint64_t seed, uint64_t stream_id,
                       int64_t n, float* dz, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  act_dropout_bwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(dy),
                                                                 reinterpret_cast<const f32x4*>(z), kind, p, seed, stream_id,
                                                                 n / 4, reinterpret_cast<f32x4*>(dz));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_swiglu_fwd(const float* a, const float* b, float p, uint64_t seed, uint64_t stream_id, int64_t n, float* y,
                  hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  swiglu_fwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(b),
                                                            p, seed, stream_id, n / 4, reinterpret_cast<f32x4*>(y));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_swiglu_bwd(const float* dy, const float* a, const float* b, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                  float* da, float* db, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  swiglu_bwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(a),
                                                            reinterpret_cast<const f32x4*>(b), p, seed, stream_id, n / 4,
                                                            reinterpret_cast<f32x4*>(da), reinterpret_cast<f32x4*>(db));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_gate_fwd(const float* x, const float* gz, const float* a, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                float* y, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  gate_fwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(gz),
                                                          reinterpret_cast<const f32x4*>(a), p, seed, stream_id, n / 4,
                                                          reinterpret_cast<f32x4*>(y));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_gate_bwd(const float* dy, const float* gz, const float* a, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                float* dgz, float* da, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  gate_bwd_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(gz),
                                                          reinterpret_cast<const f32x4*>(a), p, seed, stream_id, n / 4,
                                                          reinterpret_cast<f32x4*>(dgz), reinterpret_cast<f32x4*>(da));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_axpy(const float* a, float alpha, const float* b, int64_t n, float* y, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0) return RT_ERR_INVALID_ARG;
  axpy_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(a), alpha, reinterpret_cast<const f32x4*>(b),
                                                      n / 4, reinterpret_cast<f32x4*>(y));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// y = a * b * (ids[row] != 0);  b and ids are optional (null)
int rt_mul_mask(const float* a, const float* b, const int64_t* ids, int32_t d, int64_t n, float* y, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0 || (d & 3) != 0) return RT_ERR_INVALID_ARG;
  mul_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<const f32x4*>(a), reinterpret_cast<const f32x4*>(b),
                                                     reinterpret_cast<const long long*>(ids), d / 4, n / 4,
                                                     reinterpret_cast<f32x4*>(y));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// y[r, 0:d] = a[r, 0:d] * b[r, 0:d] * (ids[r] != 0) for `rows` rows with row strides lda / ldb / ldy (floats, multiples of 4):
// operands and results may be column slices of packed projection buffers (HSTU's u / v / q / k; hstu.py:259-291)
int rt_mul_mask_ld(const float* a, int64_t lda, const float* b, int64_t ldb, const int64_t* ids, int64_t rows, int32_t d,
                   float* y, int64_t ldy, hipStream_t stream) {
  (void)hipGetLastError();
  if (rows <= 0 || d <= 0) return RT_OK;
  if ((d & 3) != 0 || (lda & 3) != 0 || (ldy & 3) != 0 || (b != nullptr && (ldb & 3) != 0)) return RT_ERR_INVALID_ARG;
  const long long n4 = (long long)rows * (d / 4);
  mul_ld_kernel<<<stream_grid(n4), 256, 0, stream>>>(a, lda, b, ldb, reinterpret_cast<const long long*>(ids), d / 4, n4, y, ldy);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// One Adam step over flat fp32 buffers (n padded to a multiple of 4).  step >= 1.
int rt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr, float beta1, float beta2,
                 float eps, float grad_scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (n <= 0) return RT_OK;
  if ((n & 3) != 0 || step < 1) return RT_ERR_INVALID_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  adam_kernel<<<stream_grid(n / 4), 256, 0, stream>>>(reinterpret_cast<f32x4*>(p), reinterpret_cast<const f32x4*>(g),
                                                      reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), n / 4,
                                                      (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Adam over `n_seg` parameter segments of the flat buffers p/m/v: segment i covers floats [offsets[i], offsets[i]+lens[i])
// (offsets multiples of 4; padding between segments is never touched) and reads its gradient from grads[i] (device
// pointer; the arrays themselves are host memory).  Segments are processed ADAM_SEGS per launch.
int rt_adam_step_segments(float* p, float* m, float* v, int32_t n_seg, const int64_t* offsets, const int64_t* lens,
                          const float* const* grads, int32_t step, float lr, float beta1, float beta2, float eps,
                          float grad_scale, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_seg < 0 || step < 1) return RT_ERR_INVALID_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  for (int s0 = 0; s0 < n_seg;) {   // s0 resumes where the previous launch stopped (skipped segments do not count)
    AdamSegs sg{};
    int chunks = 0, k = 0;
    int s = s0;
    for (; s < n_seg && k < ADAM_SEGS; ++s) {
      if ((offsets[s] & 3) != 0 || lens[s] < 0) return RT_ERR_INVALID_ARG;
      if (lens[s] == 0 || grads[s] == nullptr) continue;
      sg.grad[k] = grads[s]; sg.ofs4[k] = offsets[s] / 4; sg.len[k] = lens[s];
      sg.first_chunk[k] = chunks;
      chunks += (int)((lens[s] + 4 * ADAM_CHUNK4 - 1) / (4 * ADAM_CHUNK4));
      ++k;
    }
    sg.first_chunk[k] = chunks; sg.n = k;
    s0 = s;
    if (k == 0) continue;
    adam_segs_kernel<<<chunks, 256, 0, stream>>>(p, m, v, sg, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale);
    RT_CHECK_LAUNCH();
  }
  return RT_OK;
}

}  // extern "C"
