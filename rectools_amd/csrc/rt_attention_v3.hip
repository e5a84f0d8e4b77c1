// K4v3 — the packed softmax attention as STREAMED chunks.  Same contract, arithmetic and register geometry as K4v2
// (rt_attention_v2.hip: causal attention inside every session of a packed batch, the reference's left-pad keys as ONE virtual key per
// query — sasrec.py:186-231, torch_backbone.py:245-260; six bf16 MFMAs of an exact three-way split per fp32 product; a lane owns a
// query — in the dK/dV pass a key —, probabilities / dS feed the next product from registers), another decomposition of the work:
//
//  * K4v2 gave a (session, head) to ONE workgroup that held the whole session's K and V as LDS images (154 KB at 200 rows: one
//    workgroup per CU).  The ablation of round 6 (profiles/r6_attn_ablation.md) showed what that costs: with the products, the softmax
//    arithmetic or the memory traffic switched off the kernels kept 60–70 % of their time — the critical path is ONE long session on
//    the four SIMDs of one CU (49 tile steps of a 200-row session against 11 per SIMD on average), twice per launch (512 workgroups on
//    256 CUs), each behind its own staging prologue.
//  * Here a workgroup is 4 waves = 64 OWNER rows of a (session, head) — one 16-row tile per wave, so the waves of a workgroup differ by
//    at most one step — and the PARTNER rows stream through one 64-row chunk pair (two images, 48 KB at hd 64) that all four waves
//    read: a 200-row session is 4 workgroups (10 chunk stagings instead of one 200-row one: the price), three workgroups share a CU,
//    so one workgroup's staging runs under the others' products, and the launch has ~1,000 workgroups of at most 8 steps to balance
//    over 256 CUs instead of 512 of up to 49.  The online softmax carries (m, l, O) in registers across chunks, as it did across tiles.
//  * No LDS limit on the session length any more: any window runs (the whole-image kernels stopped at 207 rows of hd 64).
//  * Heavy workgroups first: the causal forward / dQ pass launches the LAST owner block of every session first (it sees every chunk),
//    the dK/dV pass the first key block.  Workgroups of one (session, head) are a multiple of 8 apart: same XCD, the chunks they share
//    come out of one L2.
//  * The pad keys' value-bias gradient (one row per session and head) is summed by extra workgroups of the dK/dV launch, one per
//    (session, head), in a fixed order: no atomics, bit-reproducible like everything else in the step.
#include "rt_attn_planes.h"

namespace {
using namespace rt_varlen;
using namespace rt_planes;

constexpr int CH = 64;     // partner rows per chunk (two 32-row steps) = owner rows per workgroup (4 waves x 16)
constexpr int NT = 256;    // threads per workgroup

// Ablation builds, RT_V2_ABLATE bit 1024: thread 0 of every workgroup of the forward kernel stamps s_memtime at its phase boundaries
// (slot 0 entry, 1 session offsets read, 2 + 2 c chunk c staged (behind the barrier), 3 + 2 c chunk c computed, 14 epilogue done, 15 = n
// << 32 | owner block; 13 / 12 = the device-wide 100 MHz clock at entry / exit) — read back through rt_v3_trace_read (scripts/attn_trace_v3.py): the launch's timeline as the device saw it.
#ifdef RT_ABLATION_BUILD
__device__ unsigned long long rt_v3_trace_buf[8192 * 16];
#define RT_TR(a, slot, val)                                                                                      \
  do {                                                                                                           \
    if (RT_ABL(a, 1024) && threadIdx.x == 0 && (slot) < 16 && blockIdx.x < 8192) rt_v3_trace_buf[blockIdx.x * 16 + (slot)] = (val); \
  } while (0)
#define RT_NOW() __builtin_readcyclecounter()
#else
#define RT_TR(a, slot, val) do {} while (0)
#define RT_NOW() 0ull
#endif

// Partner rows per chunk: 64 (two 32-row steps) up to hd 64; 32 at hd 128, where a row's three planes are 768 bytes (two 32-row images
// = 48 KB, as two 64-row ones at hd 64).
constexpr int chunk_rows(int hd) { return hd > 64 ? 32 : 64; }

// Session b of a launch: rows cu[b] .. cu[b+1] - 1, or — cu == NULL — `uniform_len` rows each (the padded [B, L] window seen as B
// sessions of L rows: rt_mha_fwd / rt_mha_bwd without a key-padding mask).
__device__ __forceinline__ void session_rows(const VarlenArgs& a, int b, long long& row0, int& n) {
  if (a.cu != nullptr) {
    row0 = a.cu[b];
    n = (int)(a.cu[b + 1] - row0);
  } else {
    row0 = (long long)b * a.uniform_len;
    n = a.uniform_len;
  }
}

// Rows [0, rows) of two [*, ld] fp32 matrices (columns [0, HD) of this head) -> two ROWS-row LDS images (value * scale, three bf16 planes,
// K4v2's row layout and swizzle on the LOCAL row index); rows [rows, 64) are zero-filled.  Every load is issued before any split.
template <int HD, int ROWS = CH>
__device__ __forceinline__ void stage_chunk2(const float* __restrict__ srcA, long long ldA, float scaleA, unsigned char* imgA,
                                             const float* __restrict__ srcB, long long ldB, float scaleB, unsigned char* imgB, int rows, int tid,
                                             const float* __restrict__ preA = nullptr, const float* __restrict__ preB = nullptr, int n_pre = 0) {
  // n_pre > 0 (a session behind a shared pad prefix, see `prefix_len`): the chunk's first n_pre rows come from preA / preB (rows 0 ..
  // n_pre - 1 of them), the rows behind from srcA / srcB (rows 0 ..)
  using L = Lay<HD>;
  constexpr int C4 = HD / 4, U = ROWS * C4 / NT;      // float4 per thread and image: 4 (hd 64 / 64 rows, hd 128 / 32 rows), 2 (hd 32)
  f32x4 xa[U], xb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = tid + u * NT, r = idx / C4, c4 = idx % C4;
    if (r < rows) {
      const bool pre = r < n_pre;
      const float* pa = pre ? preA + (long long)r * ldA : srcA + (long long)(r - n_pre) * ldA;
      const float* pb = pre ? preB + (long long)r * ldB : srcB + (long long)(r - n_pre) * ldB;
      xa[u] = *reinterpret_cast<const f32x4*>(pa + c4 * 4);
      xb[u] = *reinterpret_cast<const f32x4*>(pb + c4 * 4);
    } else {
      xa[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      xb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto put = [&](unsigned char* img, int idx, const f32x4& x) {
    const int r = idx / C4, c4 = idx % C4;
    u32x2 h, m, l;
    { unsigned a, b, c; split2(x[0], x[1], a, b, c); h[0] = a; m[0] = b; l[0] = c; }
    { unsigned a, b, c; split2(x[2], x[3], a, b, c); h[1] = a; m[1] = b; l[1] = c; }
    unsigned char* p = img + r * L::ROW3 + (((unsigned)c4 ^ (L::swz(r) << 1)) << 3);
    *reinterpret_cast<u32x2*>(p) = h;
    *reinterpret_cast<u32x2*>(p + L::ROWB) = m;
    *reinterpret_cast<u32x2*>(p + 2 * L::ROWB) = l;
  };
#pragma unroll
  for (int u = 0; u < U; ++u) { put(imgA, tid + u * NT, xa[u] * scaleA); put(imgB, tid + u * NT, xb[u] * scaleB); }
}

// The shared pad prefix (VarlenArgs.prefix_sessions > 0): a stack that neither masks pad keys nor re-zeroes pad rows (LiGR, ligr.py:161-191;
// Pre-LN without a key-padding mask, net_blocks.py:290-310) gives the left pads of its [B, L] window a state — which depends on the
// position only: a pad row sees pad rows, and every session's pads start from the same rows (zero item row + positional row).  The
// packed batch carries that state ONCE, as session number `prefix_sessions` (L rows: the window's positions 0 .. L - 1 as pads); a real
// session of n rows sees its first n_pre = window - n rows as keys in front of its own.  Keys and queries are numbered by their
// window position (virtual index): pad key j -> j, own row r -> n_pre + r.
__device__ __forceinline__ int prefix_len(const VarlenArgs& a, int b, int n) {
  return (a.prefix_sessions > 0 && b < a.prefix_sessions && a.window > n) ? a.window - n : 0;
}

// blockIdx -> (owner block, session, head).  heavy_last: the LAST owner block of a session is the heaviest (causal queries) and is
// launched first; else block 0 is (causal keys) / all weigh the same (bidirectional).
struct Work { int ob, b, h, bh; };
__device__ __forceinline__ Work work_of(const VarlenArgs& a, int n_ob, bool heavy_last) {
  const int per = a.B * a.H, o = (int)blockIdx.x / per;
  Work w;
  w.ob = heavy_last ? n_ob - 1 - o : o;
  w.bh = (int)blockIdx.x % per;
  w.b = w.bh / a.H;
  w.h = w.bh % a.H;
  return w;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: a lane owns a query; the session's keys / values stream through the chunk images
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool TRAIN, bool CAUSAL>
__global__ __launch_bounds__(NT, HD > 64 ? 2 : 3) void v3_fwd_kernel(VarlenArgs a, int n_ob) {
  using L = Lay<HD>;
  constexpr int CR = chunk_rows(HD);      // partner rows per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  RT_TR(a, 0, RT_NOW());
  RT_TR(a, 13, __builtin_amdgcn_s_memrealtime());      // (the device-wide 100 MHz clock: s_memtime counts per XCD)
  const Work wk = work_of(a, n_ob, CAUSAL);
  const int h = wk.h;
  long long row0; int n;
  session_rows(a, wk.b, row0, n);
  RT_TR(a, 15, ((unsigned long long)(unsigned)n << 32) | (unsigned)wk.ob);
  RT_TR(a, 1, RT_NOW());
  RT_TR(a, 12, __builtin_amdgcn_s_memrealtime());      // (overwritten at the exit of a workgroup with rows)
  if (wk.ob * CH >= n) return;
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + (size_t)CR * L::ROW3;

  const int q0 = wk.ob * CH + 16 * wave;          // this wave's owner tile
  const bool active = q0 < n;
  const int qrow = q0 + i;
  const bool qok = qrow < n;
  const long long grow = row0 + (qok ? qrow : n - 1);
  const int n_pre = CAUSAL ? prefix_len(a, wk.b, n) : 0;      // rows of the shared pad prefix in front of this session's keys
  const long long pre0 = n_pre > 0 ? a.cu[a.prefix_sessions] : 0;
  const int vq0 = n_pre + q0, vq = n_pre + qrow, nv = n_pre + n;      // window positions of the tile's first query / this lane's query; keys in all
  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = CAUSAL && a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = TRAIN ? drop_thr16(a.p_drop) : 0u;
  const float inv_keep = (TRAIN && a.p_drop > 0.f) ? 1.f / (1.f - a.p_drop) : 1.f;
  const float qscale = a.scale * LOG2E;          // scores live in the base-2 domain: p = exp2(s' - m')

  // the owner rows' loads are issued here and split after the first chunk's barrier: they fly under the chunk's staging
  f32x4 Qraw[L::NCB];
  if (active && !RT_ABL(a, 64)) load_owner_raw<HD>(a.q + grow * a.ldq + h * HD, g, Qraw);
  P3 Qp[L::NS];
  float m = -INFINITY, lsum = 0.f;               // lsum: this lane's share of the row sum (its own keys), reduced at the end
  f32x4 oT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) oT[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_chunks = CAUSAL ? min(n_pre + wk.ob * CH + CH - 1, nv - 1) / CR + 1 : (n + CR - 1) / CR;
  for (int c = 0; c < n_chunks; ++c) {
    if (c > 0 && !RT_ABL(a, 32)) __syncthreads();                  // the previous chunk's readers are done
    if (!RT_ABL(a, 1)) {
      const int own0 = max(c * CR - n_pre, 0), pre_rows = min(max(n_pre - c * CR, 0), CR);      // (keys by window position: pads, then own rows)
      stage_chunk2<HD, CR>(a.k + (row0 + own0) * a.ldk + h * HD, a.ldk, 1.f, Kimg, a.v + (row0 + own0) * a.ldv + h * HD, a.ldv, 1.f, Vimg,
                           min(CR, nv - c * CR), tid, a.k + (pre0 + c * CR) * a.ldk + h * HD, a.v + (pre0 + c * CR) * a.ldv + h * HD, pre_rows);
    }
    if (!RT_ABL(a, 32)) __syncthreads();
    RT_TR(a, 2 + 2 * c, RT_NOW());
    if (!active || RT_ABL(a, 2)) continue;
    if (c == 0 && !RT_ABL(a, 64)) split_owner_raw<HD>(Qraw, qscale, Qp);
#pragma unroll
    for (int s = 0; s < CR / 32; ++s) {
      const int t0 = c * CR + 32 * s;            // first key of the step (window position)
      if (CAUSAL ? t0 > vq0 + 15 : t0 >= n) break;                // (wave-uniform) nothing of the step is visible to the tile
      f32x4 sT[2];
      rows_times_owner<HD>(Kimg, 32 * s, CR, Qp, i, g, sT, RT_ABLV(a));      // sT[kb][r]: key t0 + 16 kb + 4 g + r
      float sc[8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[4 * kb + r] = sT[kb][r];
      if (!RT_ABL(a, 4)) {
        if (CAUSAL ? t0 + 31 > vq0 : t0 + 31 >= n) {               // the causal edge (keys behind the session's end lie behind it too)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int key = t0 + 16 * (e >> 2) + 4 * g + (e & 3);
            sc[e] = (CAUSAL ? key <= vq : key < n) ? sc[e] : -INFINITY;
          }
        }
        float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        mx = fmaxf(m, quad_max(mx));                               // finite: every query sees key t0 of every step it visits
        const float alpha = __builtin_amdgcn_exp2f(m - mx);
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = __builtin_amdgcn_exp2f(sc[e] - mx); ps += sc[e]; }
        lsum = lsum * alpha + ps;
        m = mx;
        if (TRAIN && thr16 != 0u) {      // dropout acts on the normalised probabilities: the row sum above stays undropped
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const unsigned key = (unsigned)(t0 + 16 * (e >> 2) + 4 * g + (e & 3));
            const unsigned hsh = drop_hash(a.seed, (unsigned)wk.bh, (unsigned)vq, key >> 1);
            sc[e] = (hsh & 0xFFFFu) >= thr16 ? sc[e] * inv_keep : 0.f;
            sc[e + 1] = (hsh >> 16) >= thr16 ? sc[e + 1] * inv_keep : 0.f;
          }
        }
#pragma unroll
        for (int cb = 0; cb < L::NCB; ++cb) oT[cb] *= alpha;
      }
      const P3 Pp = RT_SPLIT8(a, sc);
      cols_times_slots<HD>(Vimg, 32 * s, CR, Pp, i, g, oT, RT_ABLV(a));      // oT[cb][r]: column 16 cb + 4 g + r of query `qrow`
    }
    RT_TR(a, 3 + 2 * c, RT_NOW());
  }
  if (!active || RT_ABL(a, 2)) return;

  if (pads) {   // the window's pad keys: one virtual key, logit q.b_k / sqrt(hd), value b_v, multiplicity n_pad
    float dp = 0.f;
    const float* qp = a.q + grow * a.ldq + h * HD;
#pragma unroll
    for (int s = 0; s < L::NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) dp += qp[32 * s + 8 * g + e] * a.bk[h * HD + 32 * s + 8 * g + e];
    dp = quad_sum(dp) * qscale;
    const float mx = fmaxf(m, dp);
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    const float e1 = __builtin_amdgcn_exp2f(dp - mx);
    lsum = lsum * alpha + (g == 0 ? (float)n_pad * e1 : 0.f);
    m = mx;
    float wv = (float)n_pad * e1;
    if (TRAIN && thr16 != 0u) {   // value side: the pads that survive the dropout, counted by the four lanes of the query
      const int kept = pads_kept_quad(a.seed, (unsigned)wk.bh, (unsigned)qrow, n, n_pad, thr16, g);
      wv = (float)kept * inv_keep * e1;
    }
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) {
      const f32x4 bv4 = *reinterpret_cast<const f32x4*>(a.bv + h * HD + 16 * cb + 4 * g);
      oT[cb] = oT[cb] * alpha + bv4 * wv;
    }
  }

  const float l = quad_sum(lsum);
  if (qok && !RT_ABL(a, 128)) {
    if (a.lse != nullptr && g == 0) a.lse[(row0 + qrow) * a.H + h] = (m + __builtin_amdgcn_logf(l)) * LN2;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    float* op = a.o + (row0 + qrow) * a.ldo + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(op + 16 * cb + 4 * g) = oT[cb] * inv;
  }
  RT_TR(a, 14, RT_NOW());
  RT_TR(a, 12, __builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, pass 1: dQ and delta = rowsum(dO * O).  A lane owns a query; S^T and dP^T = V dO^T are recomputed per step, dS^T = P (drop *
// dP - delta) stays in registers and feeds dQ^T = K^T dS^T (transpose reads of the K image).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NT, HD > 64 ? 1 : 3) void v3_bwd_dq_kernel(VarlenArgs a, int n_ob) {
  using L = Lay<HD>;
  constexpr int CR = chunk_rows(HD);      // partner rows per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const Work wk = work_of(a, n_ob, CAUSAL);
  const int h = wk.h;
  long long row0; int n;
  session_rows(a, wk.b, row0, n);
  if (wk.ob * CH >= n) return;
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + (size_t)CR * L::ROW3;

  const int q0 = wk.ob * CH + 16 * wave;
  const bool active = q0 < n;
  const int qrow = q0 + i;
  const bool qok = qrow < n;
  const long long grow = row0 + (qok ? qrow : n - 1);
  const int n_pre = CAUSAL ? prefix_len(a, wk.b, n) : 0;      // (the shared pad prefix: see the forward kernel)
  const long long pre0 = n_pre > 0 ? a.cu[a.prefix_sessions] : 0;
  const int vq0 = n_pre + q0, vq = n_pre + qrow, nv = n_pre + n;
  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = CAUSAL && a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float qscale = a.scale * LOG2E;
  const float* qp = a.q + grow * a.ldq + h * HD;
  const float* dop = a.dout + grow * a.lddo + h * HD;
  const float* op = a.o + grow * a.ldo + h * HD;

  P3 Qp[L::NS], Dp[L::NS];
  float dl = 0.f, lse2 = 0.f;                     // delta = rowsum(dO * O): 16 of the HD columns per lane
  if (active && !RT_ABL(a, 64)) {
    load_owner_planes<HD>(qp, g, qscale, Qp);
    load_owner_planes<HD>(dop, g, qok ? 1.f : 0.f, Dp);
#pragma unroll
    for (int s = 0; s < L::NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += dop[32 * s + 8 * g + e] * op[32 * s + 8 * g + e];
    dl = qok ? quad_sum(dl) : 0.f;
    lse2 = a.lse[grow * a.H + h] * LOG2E;
    if (qok && g == 0) a.delta[grow * a.H + h] = dl;
  }
  f32x4 dqT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) dqT[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_chunks = CAUSAL ? min(n_pre + wk.ob * CH + CH - 1, nv - 1) / CR + 1 : (n + CR - 1) / CR;
  for (int c = 0; c < n_chunks; ++c) {
    if (c > 0 && !RT_ABL(a, 32)) __syncthreads();
    if (!RT_ABL(a, 1)) {
      const int own0 = max(c * CR - n_pre, 0), pre_rows = min(max(n_pre - c * CR, 0), CR);
      stage_chunk2<HD, CR>(a.k + (row0 + own0) * a.ldk + h * HD, a.ldk, 1.f, Kimg, a.v + (row0 + own0) * a.ldv + h * HD, a.ldv, 1.f, Vimg,
                           min(CR, nv - c * CR), tid, a.k + (pre0 + c * CR) * a.ldk + h * HD, a.v + (pre0 + c * CR) * a.ldv + h * HD, pre_rows);
    }
    if (!RT_ABL(a, 32)) __syncthreads();
    if (!active || RT_ABL(a, 2)) continue;
#pragma unroll
    for (int s = 0; s < CR / 32; ++s) {
      const int t0 = c * CR + 32 * s;              // (window position of the step's first key)
      if (CAUSAL ? t0 > vq0 + 15 : t0 >= n) break;
      f32x4 sT[2], dpT[2];
      rows_times_owner<HD>(Kimg, 32 * s, CR, Qp, i, g, sT, RT_ABLV(a));
      rows_times_owner<HD>(Vimg, 32 * s, CR, Dp, i, g, dpT, RT_ABLV(a));
      float ds[8];
      if (RT_ABL(a, 4)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ds[e] = sT[e >> 2][e & 3] + dpT[e >> 2][e & 3];
      } else {
        const bool edge = CAUSAL ? t0 + 31 > vq0 : t0 + 31 >= n;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = t0 + 16 * (e >> 2) + 4 * g + (e & 3);
          ds[e] = (!edge || (CAUSAL ? key <= vq : key < n)) ? __builtin_amdgcn_exp2f(sT[e >> 2][e & 3] - lse2) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float d0 = dpT[e >> 2][e & 3], d1 = dpT[e >> 2][(e & 3) + 1];
          if (thr16 != 0u) {
            const unsigned key = (unsigned)(t0 + 16 * (e >> 2) + 4 * g + (e & 3));
            const unsigned hsh = drop_hash(a.seed, (unsigned)wk.bh, (unsigned)vq, key >> 1);
            d0 = (hsh & 0xFFFFu) >= thr16 ? d0 * inv_keep : 0.f;
            d1 = (hsh >> 16) >= thr16 ? d1 * inv_keep : 0.f;
          }
          ds[e] *= d0 - dl;                   // dS^T
          ds[e + 1] *= d1 - dl;
        }
      }
      const P3 Sp = RT_SPLIT8(a, ds);
      cols_times_slots<HD>(Kimg, 32 * s, CR, Sp, i, g, dqT, RT_ABLV(a));   // dQ^T[c][q] += sum_j K[j][c] dS^T[j][q]  (scale at the store)
    }
  }
  if (!active || RT_ABL(a, 2)) return;

  if (pads) {   // the virtual pad key: dS_p = P_p (drop * dO.b_v - delta), dq += dS_p b_k   (d_b_v: v3_pad_dbv, the dK/dV launch)
    float sp = 0.f, dpp = 0.f;
#pragma unroll
    for (int s = 0; s < L::NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 32 * s + 8 * g + e;
        sp += qp[c] * a.bk[h * HD + c];
        dpp += dop[c] * a.bv[h * HD + c];
      }
    sp = quad_sum(sp) * qscale;
    dpp = qok ? quad_sum(dpp) : 0.f;
    const float e1 = __builtin_amdgcn_exp2f(sp - lse2);                        // one pad key's probability
    float kept = (float)n_pad;
    if (thr16 != 0u) {
      kept = (float)pads_kept_quad(a.seed, (unsigned)wk.bh, (unsigned)qrow, n, n_pad, thr16, g) * inv_keep;
    }
    const float dsp = e1 * (kept * dpp - (float)n_pad * dl);
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) {
      const f32x4 bk4 = *reinterpret_cast<const f32x4*>(a.bk + h * HD + 16 * cb + 4 * g);
      dqT[cb] += bk4 * dsp;
    }
  }
  if (qok && !RT_ABL(a, 128)) {
    float* dqp = a.dq + grow * a.lddq + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(dqp + 16 * cb + 4 * g) = dqT[cb] * a.scale;
  }
}

// The pad keys' share of the value-bias gradient of ONE (session, head): d_bv[c] = sum over the session's queries of (the pad keys'
// dropped probability mass of the query) * dO[query][c] — what K4v2's dQ kernel summed over the queries it owned.  Needs lse only
// (written by the forward): independent of the dQ pass.  One workgroup, fixed order.
template <int HD>
__device__ __forceinline__ void v3_pad_dbv(const VarlenArgs& a, int b, int h, float* red) {
  using L = Lay<HD>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int bh = b * a.H + h;
  long long row0; int n;
  session_rows(a, b, row0, n);
  float* dbv = a.dbv_part + (long long)b * a.H * HD + h * HD;
  const int n_pad = a.window > n ? a.window - n : 0;
  if (n <= 0 || n_pad <= 0) {
    if (tid < HD) dbv[tid] = 0.f;
    return;
  }
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float qscale = a.scale * LOG2E;
  f32x4 acc[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int qt = wave; qt * 16 < n; qt += NT / 64) {
    const int qrow = qt * 16 + i;
    const bool qok = qrow < n;
    const long long grow = row0 + (qok ? qrow : n - 1);
    const float* qp = a.q + grow * a.ldq + h * HD;
    const float* dop = a.dout + grow * a.lddo + h * HD;
    float sp = 0.f;
#pragma unroll
    for (int s = 0; s < L::NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) sp += qp[32 * s + 8 * g + e] * a.bk[h * HD + 32 * s + 8 * g + e];
    sp = quad_sum(sp) * qscale;
    const float e1 = __builtin_amdgcn_exp2f(sp - a.lse[grow * a.H + h] * LOG2E);
    float kept = (float)n_pad;
    if (thr16 != 0u) {
      kept = (float)pads_kept_quad(a.seed, (unsigned)bh, (unsigned)qrow, n, n_pad, thr16, g) * inv_keep;
    }
    const float wv = qok ? e1 * kept : 0.f;                                    // dropped pad mass that multiplied b_v
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) acc[cb] += *reinterpret_cast<const f32x4*>(dop + 16 * cb + 4 * g) * wv;
  }
  // over the queries: the 16 lanes of a group, then the waves through LDS
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[cb][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      acc[cb][r] = v;
    }
  if (i == 0)
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(red + wave * HD + 16 * cb + 4 * g) = acc[cb];
  __syncthreads();
  if (tid < HD) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) v += red[w * HD + tid];
    dbv[tid] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV.  A lane owns a KEY; the session's queries stream through the chunk images (Q pre-scaled by log2(e) / sqrt(hd),
// dO) with their lse / delta beside them.  Workgroups behind the key blocks sum the pad keys' value-bias gradient (v3_pad_dbv).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(NT, HD > 64 ? 1 : 2) void v3_bwd_dkv_kernel(VarlenArgs a, int n_ob) {
  using L = Lay<HD>;
  constexpr int CR = chunk_rows(HD);      // partner rows per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int per = a.B * a.H;
  if ((int)blockIdx.x >= per * n_ob) {                    // (only launched when the batch has pad keys)
    const int bh = (int)blockIdx.x - per * n_ob;
    v3_pad_dbv<HD>(a, bh / a.H, bh % a.H, reinterpret_cast<float*>(smem));
    return;
  }
  const Work wk = work_of(a, n_ob, false);                // key block 0 sees every query chunk: heaviest first as it is
  const int h = wk.h;
  long long row0; int n;
  session_rows(a, wk.b, row0, n);
  if (wk.ob * CH >= n) return;
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + (size_t)CR * L::ROW3;
  float* Ls = reinterpret_cast<float*>(smem + 2 * (size_t)CR * L::ROW3);   // [CR] lse * log2(e) of the chunk's queries
  float* Dl = Ls + CR;                                                      // [CR] delta
  const float qscale = a.scale * LOG2E;

  const int k0 = wk.ob * CH + 16 * wave;           // this wave's key tile
  const bool active = k0 < n;
  const int krow = k0 + i;
  const bool kok = krow < n;
  const long long grow = row0 + (kok ? krow : n - 1);
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const int n_pre = CAUSAL ? prefix_len(a, wk.b, n) : 0;
  P3 Kp[L::NS], Vp[L::NS];
  if (active && !RT_ABL(a, 64)) {
    load_owner_planes<HD>(a.k + grow * a.ldk + h * HD, g, 1.f, Kp);
    load_owner_planes<HD>(a.v + grow * a.ldv + h * HD, g, 1.f, Vp);
  }
  f32x4 dkT[L::NCB], dvT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) { dkT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int n_chunks = (n + CR - 1) / CR;
  const int c_first = CAUSAL ? wk.ob * CH / CR : 0;
  for (int c = c_first; c < n_chunks; ++c) {                 // causal: query chunks at or behind the key block
    if (c > c_first && !RT_ABL(a, 32)) __syncthreads();
    const int rows = min(CR, n - c * CR);
    if (!RT_ABL(a, 1)) {
      stage_chunk2<HD, CR>(a.q + (row0 + c * CR) * a.ldq + h * HD, a.ldq, qscale, Qimg, a.dout + (row0 + c * CR) * a.lddo + h * HD, a.lddo, 1.f, Dimg,
                           rows, tid);
      if (tid < CR) {
        Ls[tid] = tid < rows ? a.lse[(row0 + c * CR + tid) * a.H + h] * LOG2E : 0.f;
        Dl[tid] = tid < rows ? a.delta[(row0 + c * CR + tid) * a.H + h] : 0.f;
      }
    }
    if (!RT_ABL(a, 32)) __syncthreads();
    if (!active || RT_ABL(a, 2)) continue;
#pragma unroll
    for (int s = 0; s < CR / 32; ++s) {
      const int t0 = c * CR + 32 * s;              // first query of the step
      if (t0 >= n) break;
      if (CAUSAL && t0 + 31 < k0) continue;        // (wave-uniform) every query of the step lies before every key of the tile
      f32x4 sm[2], dpm[2];                         // S[q][key], dP[q][key]: register (qb, r) = query t0 + 16 qb + 4 g + r
      rows_times_owner<HD>(Qimg, 32 * s, CR, Kp, i, g, sm, RT_ABLV(a));
      rows_times_owner<HD>(Dimg, 32 * s, CR, Vp, i, g, dpm, RT_ABLV(a));
      const bool edge = (CAUSAL && t0 < k0 + 16) || t0 + 31 >= n || k0 + 15 >= n;
      float pd[8], ds[8];
      if (RT_ABL(a, 4)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { pd[e] = sm[e >> 2][e & 3]; ds[e] = dpm[e >> 2][e & 3]; }
      } else
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const f32x4 ls4 = *reinterpret_cast<const f32x4*>(Ls + 32 * s + 16 * qb + 4 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Dl + 32 * s + 16 * qb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qr = t0 + 16 * qb + 4 * g + r;
          float pr = __builtin_amdgcn_exp2f(sm[qb][r] - ls4[r]);
          if (edge) pr = ((!CAUSAL || krow <= qr) && qr < n && kok) ? pr : 0.f;
          float keepf = 1.f;
          if (thr16 != 0u)      // (numbered by window position when the session sits behind a shared pad prefix)
            keepf = drop_kept(a.seed, (unsigned)wk.bh, (unsigned)(n_pre + qr), (unsigned)(n_pre + krow), thr16) ? inv_keep : 0.f;
          pd[4 * qb + r] = pr * keepf;                               // dropped probabilities (for dV)
          ds[4 * qb + r] = pr * (dpm[qb][r] * keepf - dl4[r]);       // dS
        }
      }
      const P3 Pp = RT_SPLIT8(a, pd);
      cols_times_slots<HD>(Dimg, 32 * s, CR, Pp, i, g, dvT, RT_ABLV(a));    // dV^T[c][key] += sum_q dO[q][c] P~[q][key]
      const P3 Sp = RT_SPLIT8(a, ds);
      cols_times_slots<HD>(Qimg, 32 * s, CR, Sp, i, g, dkT, RT_ABLV(a));    // dK^T[c][key] += sum_q Q'[q][c] dS[q][key]
    }
  }
  if (!active || RT_ABL(a, 2)) return;
  if (kok && !RT_ABL(a, 128)) {
    float* dkp = a.dk + grow * a.lddk + h * HD;
    float* dvp = a.dv + grow * a.lddv + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) {
      *reinterpret_cast<f32x4*>(dkp + 16 * cb + 4 * g) = dkT[cb] * LN2;     // Q' = Q * scale * log2(e): scale is in, log2(e) comes out
      *reinterpret_cast<f32x4*>(dvp + 16 * cb + 4 * g) = dvT[cb];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, pass 3 (shared pad prefix only): the prefix rows' dK / dV from the sessions BEHIND the prefix.  A lane owns a prefix key
// (window position j); the queries of session b see it iff j < n_pre(b) = window - n_b, all of them (every pad precedes every real row).
// A workgroup = 64 prefix keys x one head x one GROUP of sessions (b = group, group + 16, ...): it streams those sessions' query
// chunks like the dK/dV pass and writes ONE partial row block; v3_prefix_reduce_kernel adds the 16 groups' partials in a fixed order to
// what the dK/dV pass wrote for the prefix session's own queries.  No atomics.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int PREFIX_GROUPS = 16;

template <int HD>
__global__ __launch_bounds__(NT, HD > 64 ? 1 : 2) void v3_prefix_dkv_kernel(VarlenArgs a, int n_kb) {
  using L = Lay<HD>;
  constexpr int CR = chunk_rows(HD);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = (int)blockIdx.x % a.H, kb = ((int)blockIdx.x / a.H) % n_kb, grp = (int)blockIdx.x / (a.H * n_kb);
  const int Lw = a.window;
  const long long pre0 = a.cu[a.prefix_sessions];
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + (size_t)CR * L::ROW3;
  float* Ls = reinterpret_cast<float*>(smem + 2 * (size_t)CR * L::ROW3);
  float* Dl = Ls + CR;
  const float qscale = a.scale * LOG2E;
  const int k0 = kb * CH + 16 * wave;
  const bool active = k0 < Lw;
  const int krow = k0 + i;                     // the key's window position
  const bool kok = krow < Lw;
  const long long grow = pre0 + (kok ? krow : Lw - 1);
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  P3 Kp[L::NS], Vp[L::NS];
  if (active) {
    load_owner_planes<HD>(a.k + grow * a.ldk + h * HD, g, 1.f, Kp);
    load_owner_planes<HD>(a.v + grow * a.ldv + h * HD, g, 1.f, Vp);
  }
  f32x4 dkT[L::NCB], dvT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) { dkT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int b = grp; b < a.prefix_sessions; b += PREFIX_GROUPS) {
    const long long row0 = a.cu[b];
    const int n = (int)(a.cu[b + 1] - row0);
    const int n_pre = Lw > n ? Lw - n : 0;
    if (n <= 0 || kb * CH >= n_pre) continue;              // (uniform) none of this block's keys is a pad of session b
    const unsigned bh = (unsigned)(b * a.H + h);
    for (int c = 0; c * CR < n; ++c) {
      __syncthreads();
      const int rows = min(CR, n - c * CR);
      stage_chunk2<HD, CR>(a.q + (row0 + c * CR) * a.ldq + h * HD, a.ldq, qscale, Qimg, a.dout + (row0 + c * CR) * a.lddo + h * HD, a.lddo, 1.f, Dimg,
                           rows, tid);
      if (tid < CR) {
        Ls[tid] = tid < rows ? a.lse[(row0 + c * CR + tid) * a.H + h] * LOG2E : 0.f;
        Dl[tid] = tid < rows ? a.delta[(row0 + c * CR + tid) * a.H + h] : 0.f;
      }
      __syncthreads();
      if (!active || k0 >= n_pre) continue;                // (wave-uniform) this tile's keys all lie behind the session's pads
#pragma unroll
      for (int s = 0; s < CR / 32; ++s) {
        const int t0 = c * CR + 32 * s;                    // first query (own row) of the step
        if (t0 >= n) break;
        f32x4 sm[2], dpm[2];
        rows_times_owner<HD>(Qimg, 32 * s, CR, Kp, i, g, sm);
        rows_times_owner<HD>(Dimg, 32 * s, CR, Vp, i, g, dpm);
        float pd[8], ds[8];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const f32x4 ls4 = *reinterpret_cast<const f32x4*>(Ls + 32 * s + 16 * qb + 4 * g);
          const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Dl + 32 * s + 16 * qb + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qr = t0 + 16 * qb + 4 * g + r;
            float pr = (krow < n_pre && qr < n) ? __builtin_amdgcn_exp2f(sm[qb][r] - ls4[r]) : 0.f;
            float keepf = 1.f;
            if (thr16 != 0u) keepf = drop_kept(a.seed, bh, (unsigned)(n_pre + qr), (unsigned)krow, thr16) ? inv_keep : 0.f;
            pd[4 * qb + r] = pr * keepf;
            ds[4 * qb + r] = pr * (dpm[qb][r] * keepf - dl4[r]);
          }
        }
        const P3 Pp = split8(pd);
        cols_times_slots<HD>(Dimg, 32 * s, CR, Pp, i, g, dvT);
        const P3 Sp = split8(ds);
        cols_times_slots<HD>(Qimg, 32 * s, CR, Sp, i, g, dkT);
      }
    }
  }
  if (active && kok) {
    float* wk_ = a.prefix_ws + ((long long)grp * Lw + krow) * (2 * a.H * HD) + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) {
      *reinterpret_cast<f32x4*>(wk_ + 16 * cb + 4 * g) = dkT[cb] * LN2;
      *reinterpret_cast<f32x4*>(wk_ + a.H * HD + 16 * cb + 4 * g) = dvT[cb];
    }
  }
}

// dk / dv rows of the prefix session += the 16 groups' partials, summed in group order
__global__ __launch_bounds__(256) void v3_prefix_reduce_kernel(VarlenArgs a) {
  const int j = blockIdx.x, Lw = a.window, w2 = 2 * a.H * a.hd, d = a.H * a.hd;
  const long long row = a.cu[a.prefix_sessions] + j;
  for (int c = threadIdx.x; c < w2; c += 256) {
    float v = 0.f;
    for (int grp = 0; grp < PREFIX_GROUPS; ++grp) v += a.prefix_ws[((long long)grp * Lw + j) * w2 + c];
    float* dst = c < d ? a.dk + row * a.lddk + c : a.dv + row * a.lddv + (c - d);
    *dst += v;
  }
}

#ifdef RT_ABLATION_BUILD
inline int v3_ablate_env() { const char* e = getenv("RT_V2_ABLATE"); return e != nullptr ? atoi(e) : 0; }
#endif

// Owner blocks per (session, head) of a launch.  At least two: the unused tail of a packed row block (up to 128 rows) rides along as
// one more session whatever the window, and every row of it must be written (finite values: zero gradients flow in, but a weight
// gradient sums 0 x whatever the row holds).
inline int owner_blocks(int max_len) { const int n = (max_len + CH - 1) / CH; return n < 2 ? 2 : n; }

template <int HD, bool TRAIN, bool CAUSAL>
int launch_fwd(VarlenArgs a, int max_len, hipStream_t stream) {
  const int n_ob = owner_blocks(max_len);
  const size_t lds = 2 * (size_t)chunk_rows(HD) * Lay<HD>::ROW3;
#ifdef RT_ABLATION_BUILD
  a.ablate = v3_ablate_env();
#endif
  v3_fwd_kernel<HD, TRAIN, CAUSAL><<<a.B * a.H * n_ob, NT, lds, stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

template <int HD, bool CAUSAL>
int launch_bwd(VarlenArgs a, int max_len, hipStream_t stream) {
  const int n_ob = owner_blocks(max_len);
  const size_t lds = 2 * (size_t)chunk_rows(HD) * Lay<HD>::ROW3, lds_kv = lds + 2 * chunk_rows(HD) * sizeof(float);
  const bool pad_wgs = CAUSAL && a.dbv_part != nullptr && a.bk != nullptr && a.bv != nullptr;
  int skip = 0;
#ifdef RT_ABLATION_BUILD
  a.ablate = v3_ablate_env();
  skip = a.ablate;
#endif
  if (!(skip & 256)) v3_bwd_dq_kernel<HD, CAUSAL><<<a.B * a.H * n_ob, NT, lds, stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  if (!(skip & 512)) v3_bwd_dkv_kernel<HD, CAUSAL><<<a.B * a.H * (n_ob + (pad_wgs ? 1 : 0)), NT, lds_kv, stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  if (CAUSAL && a.prefix_sessions > 0) {      // the prefix rows' dK / dV from the sessions behind it
    if (a.prefix_ws == nullptr) return RT_ERR_WORKSPACE;
    const int n_kb = (a.window + CH - 1) / CH;
    v3_prefix_dkv_kernel<HD><<<PREFIX_GROUPS * n_kb * a.H, NT, lds_kv, stream>>>(a, n_kb);
    RT_CHECK_LAUNCH();
    v3_prefix_reduce_kernel<<<a.window, 256, 0, stream>>>(a);
    RT_CHECK_LAUNCH();
  }
  return RT_OK;
}


// =================================================================================================================================
// K6v3 — HSTU's pointwise attention (hstu.py:270-288: silu(q k^T + rab) / L, causal; rab of hstu.py:84-128 in-kernel) on the streamed
// decomposition above.  K6v2 (rt_attention_v2.hip) gave a (session, head) to ONE workgroup: the launch lasted as long as its longest
// session (a 512-row session is 280 tile steps on four SIMDs while the average SIMD of the chip holds ~14), and the owner rows'
// accumulators travelled through memory between its 192-row chunks.  Here a 512-row session is 8 workgroups of 64 owner rows whose
// accumulators stay in registers across the 64-row chunks; no softmax, so nothing is renormalised.  Tables: the chunk's partner
// timestamps, the thresholds, the time weights, and the 127 position weights one (owner block, chunk) pair can reach live in the LDS
// behind the two images (50.8 KB: three workgroups per CU; the dQ pass adds the two bias-gradient accumulators: 55.4 KB, two).
// =================================================================================================================================
struct HstuLds {
  unsigned char* img0; unsigned char* img1;
  long long* ts_p;       // [CH] timestamps of the chunk's partner rows
  long long* thr;        // [NBUCK]
  float* tw;             // [NBUCK + 3]
  float* pw;             // [128]: pos_w[pw_base .. pw_base + 126] of this (owner block, chunk)
  float* dtw; float* dpw;   // (dQ pass) [NBUCK + 3], [2 Lw]
};
template <int HD>
__device__ __forceinline__ HstuLds hstu3_carve(unsigned char* smem, int Lw) {
  using L = Lay<HD>;
  HstuLds l;
  l.img0 = smem; l.img1 = smem + (size_t)CH * L::ROW3;
  unsigned char* p = smem + 2 * (size_t)CH * L::ROW3;
  l.ts_p = reinterpret_cast<long long*>(p); p += CH * 8;
  l.thr = reinterpret_cast<long long*>(p); p += NBUCK * 8;
  l.tw = reinterpret_cast<float*>(p); p += (NBUCK + 3) * 4;
  l.pw = reinterpret_cast<float*>(p); p += 128 * 4;
  l.dtw = reinterpret_cast<float*>(p); p += (NBUCK + 3) * 4;
  l.dpw = reinterpret_cast<float*>(p);
  return l;
}
template <int HD> inline size_t hstu3_lds_bytes(int Lw, bool grads) {
  return 2 * (size_t)CH * Lay<HD>::ROW3 + CH * 8 + NBUCK * 8 + (NBUCK + 3) * 4 + 128 * 4 + (grads ? (NBUCK + 3) * 4 + (size_t)(2 * Lw) * 4 : 0);
}
__device__ __forceinline__ void hstu3_load_tables(const HstuV2Args& a, const HstuLds& l, int tid, bool grads) {
  const int nw = a.time_thr != nullptr ? (int)a.time_thr[NBUCK] : 1;     // entries of time_w; later buckets read the last one
  for (int j = tid; j < NBUCK; j += NT) {
    l.thr[j] = a.time_thr != nullptr ? a.time_thr[j] : 0;
    l.tw[j] = a.time_w != nullptr ? a.time_w[j < nw ? j : nw - 1] : 0.f;
    if (grads) l.dtw[j] = 0.f;
  }
  if (grads) for (int j = tid; j < 2 * a.Lw - 1; j += NT) l.dpw[j] = 0.f;
}
// the position weights of one (owner block, partner chunk) pair: slot Lw - 1 + key - query spans 127 values starting at `base`
__device__ __forceinline__ void hstu3_load_pos(const HstuV2Args& a, const HstuLds& l, int base, int tid) {
  if (tid < 128) l.pw[tid] = a.pos_w != nullptr ? a.pos_w[min(max(base + tid, 0), 2 * a.Lw - 2)] : 0.f;
}
__device__ __forceinline__ Work hstu3_work(const HstuV2Args& a, int n_ob, bool heavy_last) {
  const int per = a.B * a.H, o = (int)blockIdx.x / per;
  Work w;
  w.ob = heavy_last ? n_ob - 1 - o : o;
  w.bh = (int)blockIdx.x % per;
  w.b = w.bh / a.H;
  w.h = w.bh % a.H;
  return w;
}

// ---- forward: a lane owns a query; the keys / values stream ----------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(NT, 3) void v3_hstu_fwd_kernel(HstuV2Args a, int n_ob) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const Work wk = hstu3_work(a, n_ob, true);
  const int h = wk.h, b = wk.b;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (wk.ob * CH >= n) return;
  const HstuLds l = hstu3_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  hstu3_load_tables(a, l, tid, false);
  const float inv_l = 1.0f / (float)a.Lw;

  const int q0 = wk.ob * CH + 16 * wave;
  const bool active = q0 < n;
  const int qrow = q0 + i;
  const bool qok = qrow < n;
  const int qsafe = qok ? qrow : n - 1;
  const long long grow = row0 + qsafe;
  f32x4 Qraw[L::NCB];
  if (active) load_owner_raw<HD>(a.q + grow * a.ldq + h * HD, g, Qraw);
  const long long t_q1 = (tbias && active) ? tsb[qsafe + 1] : 0;
  P3 Qp[L::NS];
  f32x4 oT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) oT[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c <= wk.ob; ++c) {
    const int c0 = c * CH, len = min(CH, n - c0);
    if (c > 0) __syncthreads();
    stage_chunk2<HD>(a.k + (row0 + c0) * a.ldk + h * HD, a.ldk, 1.f, l.img0, a.v + (row0 + c0) * a.ldv + h * HD, a.ldv, 1.f, l.img1, len, tid);
    if (tbias && tid < CH) l.ts_p[tid] = tsb[c0 + min(tid, len - 1)];
    const int pw_base = a.Lw - 1 + c0 - (wk.ob * CH + CH - 1);
    if (pbias) hstu3_load_pos(a, l, pw_base, tid);
    __syncthreads();
    if (!active) continue;
    if (c == 0) split_owner_raw<HD>(Qraw, 1.f, Qp);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t0 = 32 * s;                         // local first key of the step
      if (c0 + t0 > q0 + 15 || t0 >= len) break;     // nothing of the step is visible to the tile
      f32x4 sT[2];
      rows_times_owner<HD>(l.img0, t0, CH, Qp, i, g, sT);
      float pr[8];
      int bk8[8];
      if (tbias) hstu_buckets8(l.thr, l.ts_p, t_q1, true, t0, g, len, qok && t0 + 19 + 4 * g < len && c0 + t0 + 19 + 4 * g <= qrow, bk8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kl = t0 + 16 * (e >> 2) + 4 * g + (e & 3), key = c0 + kl;
        const bool valid = qok && kl < len && key <= qrow;
        float bias = 0.f;
        if (tbias) bias += l.tw[bk8[e]];
        if (pbias) bias += l.pw[a.Lw - 1 + key - qrow - pw_base];
        pr[e] = valid ? hstu_silu(sT[e >> 2][e & 3] + bias) * inv_l : 0.f;
      }
      const P3 Pp = split8(pr);
      cols_times_slots<HD>(l.img1, t0, CH, Pp, i, g, oT);
    }
  }
  if (active && qok) {
    float* op = a.o + grow * a.ldo + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(op + 16 * cb + 4 * g) = oT[cb];
  }
}

// ---- backward, pass 1: dQ and the bias gradients.  A lane owns a query -----------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(NT, 2) void v3_hstu_bwd_dq_kernel(HstuV2Args a, int n_ob) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const Work wk = hstu3_work(a, n_ob, true);
  const int h = wk.h, b = wk.b;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (wk.ob * CH >= n) return;
  const HstuLds l = hstu3_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  const bool tgrad = tbias && a.d_time_w != nullptr, pgrad = pbias && a.d_pos_w != nullptr;
  hstu3_load_tables(a, l, tid, true);
  const float inv_l = 1.0f / (float)a.Lw;

  const int q0 = wk.ob * CH + 16 * wave;
  const bool active = q0 < n;
  const int qrow = q0 + i;
  const bool qok = qrow < n;
  const int qsafe = qok ? qrow : n - 1;
  const long long grow = row0 + qsafe;
  f32x4 Qraw[L::NCB], Draw[L::NCB];
  if (active) {
    load_owner_raw<HD>(a.q + grow * a.ldq + h * HD, g, Qraw);
    load_owner_raw<HD>(a.dout + grow * a.lddo + h * HD, g, Draw);
  }
  const long long t_q1 = (tbias && active) ? tsb[qsafe + 1] : 0;
  P3 Qp[L::NS], Dp[L::NS];
  f32x4 dqT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) dqT[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  BucketRun run;
  run.init();

  for (int c = 0; c <= wk.ob; ++c) {
    const int c0 = c * CH, len = min(CH, n - c0);
    if (c > 0) __syncthreads();
    stage_chunk2<HD>(a.k + (row0 + c0) * a.ldk + h * HD, a.ldk, 1.f, l.img0, a.v + (row0 + c0) * a.ldv + h * HD, a.ldv, 1.f, l.img1, len, tid);
    if (tbias && tid < CH) l.ts_p[tid] = tsb[c0 + min(tid, len - 1)];
    const int pw_base = a.Lw - 1 + c0 - (wk.ob * CH + CH - 1);
    if (pbias) hstu3_load_pos(a, l, pw_base, tid);
    __syncthreads();
    if (!active) continue;
    if (c == 0) { split_owner_raw<HD>(Qraw, 1.f, Qp); split_owner_raw<HD>(Draw, qok ? 1.f : 0.f, Dp); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t0 = 32 * s;
      if (c0 + t0 > q0 + 15 || t0 >= len) break;
      f32x4 sT[2], dpT[2];
      rows_times_owner<HD>(l.img0, t0, CH, Qp, i, g, sT);
      rows_times_owner<HD>(l.img1, t0, CH, Dp, i, g, dpT);
      float ds[8];
      int bk8[8];
      if (tbias) hstu_buckets8(l.thr, l.ts_p, t_q1, true, t0, g, len, qok && t0 + 19 + 4 * g < len && c0 + t0 + 19 + 4 * g <= qrow, bk8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kl = t0 + 16 * (e >> 2) + 4 * g + (e & 3), key = c0 + kl;
        const bool valid = qok && kl < len && key <= qrow;
        float bias = 0.f;
        int bk = 0;
        if (tbias) { bk = bk8[e]; bias += l.tw[bk]; }
        if (pbias) bias += l.pw[a.Lw - 1 + key - qrow - pw_base];
        const float z = sT[e >> 2][e & 3] + bias;
        ds[e] = valid ? dpT[e >> 2][e & 3] * inv_l * hstu_silu_d(z) : 0.f;
        if (valid && tgrad) run.add(l.dtw, bk, ds[e]);
      }
      if (pgrad) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {      // (invalid elements are zeros: a diagonal is valid or invalid as a whole up to the window's edges)
          float dmain, dwrap;
          diagonal_sums(ds + 4 * hb, i, dmain, dwrap);
          const int base = a.Lw - 1 + c0 + t0 + 16 * hb + 4 * g - q0;
          if (dmain != 0.f) atomicAdd(l.dpw + min(max(base + 3 - i, 0), 2 * a.Lw - 2), dmain);
          if (dwrap != 0.f) atomicAdd(l.dpw + min(max(base - 13 - i, 0), 2 * a.Lw - 2), dwrap);
        }
      }
      const P3 Sp = split8(ds);
      cols_times_slots<HD>(l.img0, t0, CH, Sp, i, g, dqT);       // dQ^T[c][q] += sum_j K[j][c] dS^T[j][q]
    }
  }
  if (active) {
    if (tgrad) run.flush(l.dtw);
    if (qok) {
      float* dqp = a.dq + grow * a.lddq + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(dqp + 16 * cb + 4 * g) = dqT[cb];
    }
  }
  __syncthreads();
  if (tgrad) {
    const int nw = (int)a.time_thr[NBUCK];
    for (int j = tid; j < NBUCK; j += NT) { const float v = l.dtw[j]; if (v != 0.f) atomicAdd(a.d_time_w + (j < nw ? j : nw - 1), v); }
  }
  if (pgrad) for (int j = tid; j < 2 * a.Lw - 1; j += NT) { const float v = l.dpw[j]; if (v != 0.f) atomicAdd(a.d_pos_w + j, v); }
}

// ---- backward, pass 2: dK, dV.  A lane owns a key; the queries (Q, dO images, their next-action timestamps) stream ------------------------
template <int HD>
__global__ __launch_bounds__(NT, 2) void v3_hstu_bwd_dkv_kernel(HstuV2Args a, int n_ob) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const Work wk = hstu3_work(a, n_ob, false);             // key block 0 sees every query chunk: heaviest first as it is
  const int h = wk.h, b = wk.b;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (wk.ob * CH >= n) return;
  const HstuLds l = hstu3_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  hstu3_load_tables(a, l, tid, false);
  const float inv_l = 1.0f / (float)a.Lw;

  const int k0 = wk.ob * CH + 16 * wave;
  const bool active = k0 < n;
  const int krow = k0 + i;
  const bool kok = krow < n;
  const int ksafe = kok ? krow : n - 1;
  const long long grow = row0 + ksafe;
  f32x4 Kraw[L::NCB], Vraw[L::NCB];
  if (active) {
    load_owner_raw<HD>(a.k + grow * a.ldk + h * HD, g, Kraw);
    load_owner_raw<HD>(a.v + grow * a.ldv + h * HD, g, Vraw);
  }
  const long long t_k = (tbias && active) ? tsb[ksafe] : 0;
  P3 Kp[L::NS], Vp[L::NS];
  f32x4 dkT[L::NCB], dvT[L::NCB];
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) { dkT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int n_chunks = (n + CH - 1) / CH;
  for (int c = wk.ob; c < n_chunks; ++c) {               // query chunks at or behind the key block
    const int c0 = c * CH, len = min(CH, n - c0);
    if (c > wk.ob) __syncthreads();
    stage_chunk2<HD>(a.q + (row0 + c0) * a.ldq + h * HD, a.ldq, 1.f, l.img0, a.dout + (row0 + c0) * a.lddo + h * HD, a.lddo, 1.f, l.img1, len, tid);
    if (tbias && tid < CH) l.ts_p[tid] = tsb[c0 + min(tid, len - 1) + 1];     // a query's time is its NEXT stamp (hstu.py:96-104)
    const int pw_base = a.Lw - 1 + wk.ob * CH - (c0 + CH - 1);
    if (pbias) hstu3_load_pos(a, l, pw_base, tid);
    __syncthreads();
    if (!active) continue;
    if (c == wk.ob) { split_owner_raw<HD>(Kraw, 1.f, Kp); split_owner_raw<HD>(Vraw, 1.f, Vp); }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t0 = 32 * s;                         // local first query of the step
      if (t0 >= len) break;
      if (c0 + t0 + 31 < k0) continue;               // every query of the step lies before every key of the tile
      f32x4 sm[2], dpm[2];                           // S[q][key], dP[q][key]: register (qb, r) = query c0 + t0 + 16 qb + 4 g + r
      rows_times_owner<HD>(l.img0, t0, CH, Kp, i, g, sm);
      rows_times_owner<HD>(l.img1, t0, CH, Vp, i, g, dpm);
      float pd[8], ds[8];
      int bk8[8];
      if (tbias) hstu_buckets8(l.thr, l.ts_p, t_k, false, t0, g, len, kok && t0 + 19 + 4 * g < len && krow <= c0 + t0 + 4 * g, bk8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ql = t0 + 16 * (e >> 2) + 4 * g + (e & 3), q = c0 + ql;
        const bool valid = kok && ql < len && krow <= q;
        float bias = 0.f;
        if (tbias) bias += l.tw[bk8[e]];
        if (pbias) bias += l.pw[a.Lw - 1 + krow - q - pw_base];
        const float z = sm[e >> 2][e & 3] + bias;
        float pz, dz;
        hstu_silu_both(z, pz, dz);
        pd[e] = valid ? pz * inv_l : 0.f;
        ds[e] = valid ? dpm[e >> 2][e & 3] * inv_l * dz : 0.f;
      }
      const P3 Pp = split8(pd);
      cols_times_slots<HD>(l.img1, t0, CH, Pp, i, g, dvT);       // dV^T[c][key] += sum_q dO[q][c] P[q][key]
      const P3 Sp = split8(ds);
      cols_times_slots<HD>(l.img0, t0, CH, Sp, i, g, dkT);       // dK^T[c][key] += sum_q Q[q][c] dS[q][key]
    }
  }
  if (active && kok) {
    float* dkp = a.dk + grow * a.lddk + h * HD;
    float* dvp = a.dv + grow * a.lddv + h * HD;
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) {
      *reinterpret_cast<f32x4*>(dkp + 16 * cb + 4 * g) = dkT[cb];
      *reinterpret_cast<f32x4*>(dvp + 16 * cb + 4 * g) = dvT[cb];
    }
  }
}

template <int HD>
int launch_hstu_fwd(const HstuV2Args& a, hipStream_t stream) {
  const int n_ob = owner_blocks(a.Lw);
  v3_hstu_fwd_kernel<HD><<<a.B * a.H * n_ob, NT, hstu3_lds_bytes<HD>(a.Lw, false), stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
template <int HD>
int launch_hstu_bwd(const HstuV2Args& a, hipStream_t stream) {
  const int n_ob = owner_blocks(a.Lw);
  const size_t lds_q = hstu3_lds_bytes<HD>(a.Lw, true);
  if (lds_q > 64 * 1024) {
    if (lds_q > 160 * 1024) return RT_ERR_UNSUPPORTED;
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&v3_hstu_bwd_dq_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
  }
  v3_hstu_bwd_dq_kernel<HD><<<a.B * a.H * n_ob, NT, lds_q, stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  v3_hstu_bwd_dkv_kernel<HD><<<a.B * a.H * n_ob, NT, hstu3_lds_bytes<HD>(a.Lw, false), stream>>>(a, n_ob);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace

// hd 32 / 64 / 128, any session length (the chunks stream): RT_ERR_UNSUPPORTED otherwise — the caller then takes the earlier kernels
int rt_v3_varlen_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream) {
  if (a.hd == 128) return train ? launch_fwd<128, true, true>(a, max_len, stream) : launch_fwd<128, false, true>(a, max_len, stream);
  if (a.hd == 64) return train ? launch_fwd<64, true, true>(a, max_len, stream) : launch_fwd<64, false, true>(a, max_len, stream);
  if (a.hd == 32) return train ? launch_fwd<32, true, true>(a, max_len, stream) : launch_fwd<32, false, true>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}
int rt_v3_varlen_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream) {
  if (a.hd == 128) return launch_bwd<128, true>(a, max_len, stream);
  if (a.hd == 64) return launch_bwd<64, true>(a, max_len, stream);
  if (a.hd == 32) return launch_bwd<32, true>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}
// bidirectional (no causal mask, no pad keys): BERT4Rec's key-padding-masked window on packed rows
int rt_v3_bidir_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream) {
  if (a.hd == 128) return train ? launch_fwd<128, true, false>(a, max_len, stream) : launch_fwd<128, false, false>(a, max_len, stream);
  if (a.hd == 64) return train ? launch_fwd<64, true, false>(a, max_len, stream) : launch_fwd<64, false, false>(a, max_len, stream);
  if (a.hd == 32) return train ? launch_fwd<32, true, false>(a, max_len, stream) : launch_fwd<32, false, false>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}
int rt_v3_bidir_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream) {
  if (a.hd == 128) return launch_bwd<128, false>(a, max_len, stream);
  if (a.hd == 64) return launch_bwd<64, false>(a, max_len, stream);
  if (a.hd == 32) return launch_bwd<32, false>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}

// K6v3: hd 32 / 64, a session never longer than the window Lw (rt_attention.hip's packed entry points guarantee it)
int rt_v3_hstu_fwd(const rt_varlen::HstuV2Args& a, hipStream_t stream) {
  if (a.hd == 64) return launch_hstu_fwd<64>(a, stream);
  if (a.hd == 32) return launch_hstu_fwd<32>(a, stream);
  return RT_ERR_UNSUPPORTED;
}
int rt_v3_hstu_bwd(const rt_varlen::HstuV2Args& a, hipStream_t stream) {
  if (a.hd == 64) return launch_hstu_bwd<64>(a, stream);
  if (a.hd == 32) return launch_hstu_bwd<32>(a, stream);
  return RT_ERR_UNSUPPORTED;
}

#ifdef RT_ABLATION_BUILD
// ablation builds only: the forward kernel's phase stamps (RT_V2_ABLATE bit 1024); zero-fills the buffer afterwards
extern "C" int rt_v3_trace_read(void* dst, size_t bytes) {
  const size_t n = bytes < sizeof(unsigned long long) * 8192 * 16 ? bytes : sizeof(unsigned long long) * 8192 * 16;
  if (hipDeviceSynchronize() != hipSuccess) return RT_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(rt_v3_trace_buf), n) != hipSuccess) return RT_ERR_LAUNCH;
  static unsigned long long zeros[8192 * 16];
  if (hipMemcpyToSymbol(HIP_SYMBOL(rt_v3_trace_buf), zeros, sizeof(zeros)) != hipSuccess) return RT_ERR_LAUNCH;
  return RT_OK;
}
#endif

size_t rt_v3_prefix_workspace_floats(int window, int H, int hd) { return (size_t)PREFIX_GROUPS * window * 2 * H * hd; }
