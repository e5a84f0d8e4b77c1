// K4v — softmax self-attention over PACKED (padding-free) sessions: the C entry points (docs/kernels/packed_path.md,
// docs/kernels/attention_streamed.md).
//
// The reference runs every block on the left-padded [B, L] window (sasrec.py:186-231, :300; torch_backbone.py:245-260).  Here the
// rows of a batch are the REAL positions only: session b owns rows cu[b] .. cu[b+1]-1 of q / k / v / o (n_b = cu[b+1] - cu[b] items,
// oldest first).  What the padded window adds for a causal SASRec block is closed-form: a pad key row is b_k / b_v in every
// session and block (the block input is masked to 0) and every real query sees all `window - n_b` of them, so ONE virtual key per
// query — logit q.b_k / sqrt(hd), value b_v, multiplicity n_pad — reproduces the padded softmax (tests/test_packed_equivalence.py
// pins this on the CPU oracle, values and gradients).  With key-padding masks (BERT4Rec) pass bk = bv = null: pads do not exist.
//
//   rt_mha_varlen_fwd / _train_fwd / _bwd, rt_mha_varlen_bidir_fwd / _bwd   every query of every session: the bf16-plane kernels of
//       rt_attention_v3.hip (streamed chunks, hd 32 / 64 / 128; RT_VARLEN_IMPL=v2: the whole-session-image kernels of rt_attention_v2.hip,
//       hd 32 / 64, kept as the A/B baseline).  The first form of these kernels (f32-input MFMA, rounds 2-3) lived here and was deleted in
//       round 6: two generations of kernels serve every shape it served.
//   rt_mha_varlen_last_fwd  the LAST query of every session (the final block of recommend()), one workgroup per (session, head): below.
#include "rt_common.h"
#include "rt_varlen.h"
#include <stdlib.h>
#include <string.h>

namespace {
using namespace rt_varlen;

constexpr int VT = 256;   // threads per workgroup (4 waves)


// The last query of every session (row cu[b+1]-1) against all of its keys + the virtual pad key.  q: ONE row per session
// ([B, ldq]); k, v: packed rows.  Phase 1 thread = key, phase 2 thread = (16-byte column, key phase), as the padded kernel
// attn_last_query_kernel (rt_attention.hip).  K and V are read once.
__global__ __launch_bounds__(VT) void attn_varlen_last_kernel(VarlenArgs a, int max_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [max_n rounded to 4] probabilities | 8 reduction slots
  float* prob = smem;
  float* red = smem + ((max_n + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  float* op = a.o + (long long)b * a.ldo + h * a.hd;
  if (n <= 0) {   // a session without items has no query: defined output (zeros), never consumed
    if (tid < a.hd) op[tid] = 0.f;
    return;
  }
  const float* qv = a.q + (long long)b * a.ldq + h * a.hd;
  const float* kb = a.k + row0 * a.ldk + h * a.hd;
  const float* vb = a.v + row0 * a.ldv + h * a.hd;
  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  float mx = -INFINITY;
  for (int j = tid; j < n; j += VT) {
    float sdot = 0.f;
    for (int c = 0; c < a.hd; c += 4) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(kb + (long long)j * a.ldk + c);
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(qv + c);
      sdot += k4[0] * q4[0] + k4[1] * q4[1] + k4[2] * q4[2] + k4[3] * q4[3];
    }
    const float sv = sdot * a.scale;
    prob[j] = sv;
    mx = fmaxf(mx, sv);
  }
  float dp = 0.f;
  if (pads) {
    for (int c = 0; c < a.hd; ++c) dp += qv[c] * a.bk[h * a.hd + c];
    dp *= a.scale;
    mx = fmaxf(mx, dp);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float ps = 0.f;
  for (int j = tid; j < n; j += VT) {
    const float e = __expf(prob[j] - mx);
    prob[j] = e;
    ps += e;
  }
  ps = wave_sum(ps);
  if (lane == 0) red[4 + wave] = ps;
  __syncthreads();
  const float w = pads ? (float)n_pad * __expf(dp - mx) : 0.f;
  const float l = (red[4] + red[5]) + (red[6] + red[7]) + w;
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const int ncol4 = a.hd / 4, phases = VT / ncol4;
  const int c4 = tid % ncol4, ph = tid / ncol4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ph < phases)
    for (int j = ph; j < n; j += phases) acc += *reinterpret_cast<const f32x4*>(vb + (long long)j * a.ldv + c4 * 4) * prob[j];
  __syncthreads();                                 // prob is dead: reuse the LDS for the partial sums
  f32x4* part = reinterpret_cast<f32x4*>(smem);
  if (ph < phases) part[ph * ncol4 + c4] = acc;
  __syncthreads();
  if (tid < ncol4) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int p2 = 0; p2 < phases; ++p2) o += part[p2 * ncol4 + tid];
    if (pads) o += *reinterpret_cast<const f32x4*>(a.bv + h * a.hd + tid * 4) * w;
    *reinterpret_cast<f32x4*>(op + tid * 4) = o * inv;
  }
}

constexpr size_t VARLEN_LDS_LIMIT = 160 * 1024;

// ---------------------------------------------------------------------------------------------------------------------------------
// The last query of every session WITHOUT the key / value projections of its rows (round 6: the final block of recommend()).
//   logit_j  = q_h . (W_k,h x_j + b_k,h) / sqrt(hd) = (W_k,h^T q_h) . x_j / sqrt(hd) + const     (the constant cancels in the softmax;
//                                                                                                   a pad key has x = 0: logit 0)
//   output_h = sum_j p_j (W_v,h x_j + b_v,h)        = W_v,h (sum_j p_j x_j) + b_v,h               (sum_j p_j = 1, pads included)
// The caller makes qk [B, H, d] = W_k,h^T q_h (a [B, hd] x [hd, d] product per head) and applies W_v,h to xbar [B, H, d] afterwards: the
// only pass over ALL rows is this kernel's — it reads the block input x once per phase (1 KB per row at d = 256) where the projection
// form wrote and re-read K | V (2 x that, after a [rows, d] x [d, 2d] product: 18 of the 104 ms of GEMM time of a whole-catalog recommend()).
// One workgroup per session, HG heads per pass.  Phase 1: 16 lanes per row, the row's float4 units strided over them, HG dot products
// reduced over the 16 lanes -> scores in LDS; softmax statistics; phase 2: thread = (16-byte column, row phase) as attn_varlen_last_kernel.
// (The two small products INSIDE the kernel were built and measured: 685 us per 4,096-session launch against 159 + ~160 for the kernel and
// the eight products around it — every workgroup then streams both weight matrices out of the L2 for a single session.)
// ---------------------------------------------------------------------------------------------------------------------------------
struct LastXArgs {
  const float* qk; const float* x; long long ldx; const long long* cu; float* xbar;
  int B, H, d, window, pads; float scale;
  long long prefix_row;      // >= 0: the first row of the shared pad prefix (`window` rows: the window's positions as pads, LiGR without
  //                            key-padding masks — rt_attention_v3.hip `prefix_len`): a session of n rows sees its rows [0, window - n)
  //                            as keys in front of its own; -1: none
};

template <int D64, int HG>      // d = 64 D64; HG heads per pass (H % HG == 0)
__global__ __launch_bounds__(VT) void attn_last_x_kernel(LastXArgs a, int max_n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NC4 = 16 * D64;                 // float4 units per row
  constexpr int PH = VT / NC4 > 0 ? VT / NC4 : 1;      // row phases of phase 2 (D64 = 8: 2; 4: 4; 1: 16)
  const int n4 = (max_n + 3) & ~3;
  float* prob = smem;                            // [n4][HG]
  float* red = smem + (size_t)n4 * HG;           // [2][4 waves][HG] max / sum
  f32x4* part = reinterpret_cast<f32x4*>(red + 2 * 4 * HG + ((2 * 4 * HG) & 3 ? 4 - ((2 * 4 * HG) & 3) : 0));   // [PH][HG][NC4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, d = a.d;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  float* ob = a.xbar + (long long)b * a.H * d;
  if (n <= 0) {      // a session without items has no query: defined output (zeros), never consumed
    for (int c = tid; c < a.H * d; c += VT) ob[c] = 0.f;
    return;
  }
  const float* xb = a.x + row0 * a.ldx;
  const int n_pre = (a.prefix_row >= 0 && a.window > n) ? a.window - n : 0;      // keys taken from the shared pad prefix, in front of the own rows
  const float* xpre = a.x + (a.prefix_row >= 0 ? a.prefix_row : 0) * a.ldx;
  const int nk = n_pre + n;
  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = a.pads != 0 && n_pad > 0 && n_pre == 0;
  const int grp = tid >> 4, li = tid & 15;       // phase 1: 16 groups of 16 lanes
  const int c4 = tid % NC4, ph = tid / NC4;      // phase 2
  const int hs = tid % HG;                       // the head whose statistics this thread scans (VT % HG == 0)
  for (int h0 = 0; h0 < a.H; h0 += HG) {
    if (h0 > 0) __syncthreads();                 // the previous pass's partial sums are read
    // ---- phase 1: scores
    f32x4 qr[HG][D64];
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
      for (int k = 0; k < D64; ++k)
        qr[h][k] = *reinterpret_cast<const f32x4*>(a.qk + ((long long)b * a.H + h0 + h) * d + 4 * (li + 16 * k));
    for (int j = grp; j < nk; j += 16) {
      const float* xrow = j < n_pre ? xpre + (long long)j * a.ldx : xb + (long long)(j - n_pre) * a.ldx;
      f32x4 xr[D64];
#pragma unroll
      for (int k = 0; k < D64; ++k) xr[k] = *reinterpret_cast<const f32x4*>(xrow + 4 * (li + 16 * k));
#pragma unroll
      for (int h = 0; h < HG; ++h) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < D64; ++k) s += xr[k][0] * qr[h][k][0] + xr[k][1] * qr[h][k][1] + xr[k][2] * qr[h][k][2] + xr[k][3] * qr[h][k][3];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);      // (offsets < 16 stay inside the 16-lane group)
        if (li == 0) prob[j * HG + h] = s * a.scale;
      }
    }
    __syncthreads();
    // ---- softmax statistics per head: thread tid scans entries tid, tid + VT, ... (all of head tid % HG)
    float mx = pads ? 0.f : -INFINITY;
    for (int e = tid; e < nk * HG; e += VT) mx = fmaxf(mx, prob[e]);
#pragma unroll
    for (int o = 32; o >= HG; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane < HG) red[wave * HG + lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[hs], red[HG + hs]), fmaxf(red[2 * HG + hs], red[3 * HG + hs]));
    float ps = 0.f;
    for (int e = tid; e < nk * HG; e += VT) {
      const float ex = __expf(prob[e] - mx);
      prob[e] = ex;
      ps += ex;
    }
#pragma unroll
    for (int o = 32; o >= HG; o >>= 1) ps += __shfl_xor(ps, o, 64);
    if (lane < HG) red[4 * HG + wave * HG + lane] = ps;
    __syncthreads();
    // ---- phase 2: xbar_h = sum_j p_j x_j
    f32x4 acc[HG];
#pragma unroll
    for (int h = 0; h < HG; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ph < PH)
      for (int j = ph; j < nk; j += PH) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>((j < n_pre ? xpre + (long long)j * a.ldx : xb + (long long)(j - n_pre) * a.ldx) + 4 * c4);
#pragma unroll
        for (int h = 0; h < HG; ++h) acc[h] += x4 * prob[j * HG + h];
      }
    if (ph < PH)
#pragma unroll
      for (int h = 0; h < HG; ++h) part[(ph * HG + h) * NC4 + c4] = acc[h];
    __syncthreads();
    for (int e = tid; e < HG * NC4; e += VT) {
      const int h = e / NC4, c = e % NC4;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      for (int p2 = 0; p2 < PH; ++p2) o += part[(p2 * HG + h) * NC4 + c];
      const float mh = fmaxf(fmaxf(red[h], red[HG + h]), fmaxf(red[2 * HG + h], red[3 * HG + h]));
      float l = (red[4 * HG + h] + red[5 * HG + h]) + (red[6 * HG + h] + red[7 * HG + h]);
      if (pads) l += (float)n_pad * __expf(-mh);            // the pad keys: logit 0, value b_v (applied by the caller through sum p = 1)
      *reinterpret_cast<f32x4*>(ob + (long long)(h0 + h) * d + 4 * c) = o * (l > 0.f ? 1.f / l : 0.f);
    }
  }
}


// Default: the streamed bf16-plane kernels of rt_attention_v3.hip (hd 32 / 64 / 128, any session length).  RT_VARLEN_IMPL (A/B runs): v2 =
// the whole-session-image kernels of rt_attention_v2.hip wherever they serve the shape (hd 32 / 64, two images within 160 KB of LDS; the
// streamed kernels otherwise); v2fwd / v2bwd / v3fwd / v3bwd: only that pass on the named family.
int v2_mode() {   // bit 0 / 1: v2 forward / backward, bit 2 / 3: v3 forward / backward
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("RT_VARLEN_IMPL");
    mode = (e == nullptr || e[0] == 0) ? 15
           : !strcmp(e, "v2") ? 3 : !strcmp(e, "v2fwd") ? 1 | 8 : !strcmp(e, "v2bwd") ? 2 | 4
           : !strcmp(e, "v3fwd") ? 3 | 4 : !strcmp(e, "v3bwd") ? 3 | 8 : 15;
  }
  return mode;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return RT_OK;
}

// E [d, H d]: E[r, h d + c] = W[r, c] if r / hd == h else 0 — a per-head product [B, hd] x [hd, d] for all heads becomes ONE exact-tile
// product [B, d] x [d, H d] (three quarters of it zeros: 2 GFLOP at B = 4,096, d = 256 — 20 us on the six-term loop against 4 x 40 us of
// ragged-shape launches for the four heads' own products).
__global__ __launch_bounds__(256) void expand_heads_kernel(const float* __restrict__ W, int d, int hd, int H, float* __restrict__ E) {
  const int r = blockIdx.x;
  const int hr = r / hd;
  for (int c = threadIdx.x; c < H * d; c += 256) E[(long long)r * H * d + c] = (c / d == hr) ? W[(long long)r * d + (c % d)] : 0.f;
}

template <int D64, int HG>
int launch_last_x(const LastXArgs& a, int max_len, hipStream_t stream) {
  const size_t n4 = (size_t)((max_len + 3) & ~3);
  constexpr int NC4 = 16 * D64, PH = VT / NC4 > 0 ? VT / NC4 : 1;
  const size_t lds = (n4 * HG + 2 * 4 * HG + 4) * sizeof(float) + (size_t)PH * HG * NC4 * sizeof(f32x4);
  if (lds > VARLEN_LDS_LIMIT) return RT_ERR_UNSUPPORTED;
  { const int rc = set_lds(&attn_last_x_kernel<D64, HG>, lds); if (rc != RT_OK) return rc; }
  attn_last_x_kernel<D64, HG><<<a.B, VT, lds, stream>>>(a, max_len);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
template <int D64>
int launch_last_x_h(const LastXArgs& a, int max_len, hipStream_t stream) {
  if (a.H % 4 == 0) return launch_last_x<D64, 4>(a, max_len, stream);
  if (a.H % 2 == 0) return launch_last_x<D64, 2>(a, max_len, stream);
  return launch_last_x<D64, 1>(a, max_len, stream);
}

bool bad_args(const float* q, const float* k, const float* v, const float* o, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
              const int64_t* cu, int32_t B, int32_t H, int32_t hd, int32_t max_len) {
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  return q == nullptr || k == nullptr || v == nullptr || o == nullptr || cu == nullptr || B < 0 || H <= 0 || hd <= 0 || max_len < 0 ||
         (hd & 7) != 0 || (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || mis(q) || mis(k) || mis(v) || mis(o);
}

}  // namespace

extern "C" {

// Causal softmax attention over packed sessions (see the file header).  q / k / v / o: [N, ld*] rows, head h in columns
// [h*hd, (h+1)*hd); cu_seqlens [B+1] (device, int64, ascending, cu[0] = first row); max_len >= the longest session (sizes the grid:
// ceil(max_len / 64) owner blocks per (session, head));
// bk / bv [H*hd]: the key / value projection biases, i.e. the reference's pad key / value row, or null when pad keys are masked;
// window: the reference's session_max_len (n_pad = window - n_b virtual pad keys per query).  hd in {32, 64, 128}.
int rt_mha_varlen_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                      const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float* o,
                      int64_t ldo, hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || (bk == nullptr) != (bv == nullptr))
    return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  if (hd != 32 && hd != 64 && hd != 128) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd);
  if ((v2_mode() & 1) && !(v2_mode() & 4)) {      // (A/B: the whole-session-image kernels where they serve the shape)
    const int rc = rt_v2_varlen_fwd(a, max_len, false, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v3_varlen_fwd(a, max_len, false, stream);
}

// Training forward: as rt_mha_varlen_fwd, plus attention dropout (p_drop, seed: counter-based masks the backward regenerates; the
// pad keys are dropped one by one like real keys) and lse [N, H] (log-sum-exp of every query, kept for the backward).
int rt_mha_varlen_train_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || (bk == nullptr) != (bv == nullptr) || lse == nullptr ||
      !(p_drop >= 0.f && p_drop < 1.f))
    return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  if (hd != 32 && hd != 64 && hd != 128) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = lse;
  if ((v2_mode() & 1) && !(v2_mode() & 4)) {      // (A/B: the whole-session-image kernels where they serve the shape)
    const int rc = rt_v2_varlen_fwd(a, max_len, true, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v3_varlen_fwd(a, max_len, true, stream);
}

// Backward of rt_mha_varlen_train_fwd.  delta [N, H]: workspace (rowsum(dO * O), written by the dQ pass, read by the dK/dV pass).
// dq / dk / dv rows of the sessions are fully overwritten.  dbv_part [B, H*hd] (or NULL): per-session partials of the value-bias
// gradient that comes from the pad keys (sum them over B and add to the bias gradient of the real rows; the key-bias gradient
// of the padded window is identically zero — a shift of every logit of a query — so the caller zeroes it instead).
int rt_mha_varlen_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                      const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, const float* bk, const float* bv,
                      int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float p_drop, uint64_t seed, float* dq,
                      int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta, float* dbv_part,
                      hipStream_t stream) {
  (void)hipGetLastError();
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || (bk == nullptr) != (bv == nullptr) || dout == nullptr ||
      lse == nullptr || dq == nullptr || dk == nullptr || dv == nullptr || delta == nullptr || (lddo & 3) || (lddq & 3) || (lddk & 3) ||
      (lddv & 3) || mis(dout) || mis(dq) || mis(dk) || mis(dv) || !(p_drop >= 0.f && p_drop < 1.f))
    return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  if (hd != 32 && hd != 64 && hd != 128) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = const_cast<float*>(o); a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = const_cast<float*>(lse);
  a.dout = dout; a.lddo = lddo; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.dbv_part = dbv_part;
  if ((v2_mode() & 2) && !(v2_mode() & 8)) {
    const int rc = rt_v2_varlen_bwd(a, max_len, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v3_varlen_bwd(a, max_len, stream);
}

// Causal attention over packed sessions behind a shared pad prefix (include/rectools_hip.h, K4v3p; rt_attention_v3.hip `prefix_len`).
int rt_mha_varlen_prefix_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                             const int64_t* cu_seqlens, int32_t B, int32_t n_prefixed, int32_t H, int32_t hd, int32_t max_len,
                             int32_t window, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse, hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || n_prefixed < 0 || n_prefixed >= B || window <= 0 ||
      max_len < window || !(p_drop >= 0.f && p_drop < 1.f))
    return RT_ERR_INVALID_ARG;
  if (hd != 32 && hd != 64 && hd != 128) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.H = H; a.hd = hd; a.window = window; a.prefix_sessions = n_prefixed;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = lse != nullptr ? p_drop : 0.f; a.seed = seed; a.lse = lse;
  return rt_v3_varlen_fwd(a, max_len, lse != nullptr && p_drop > 0.f, stream);
}
size_t rt_mha_varlen_prefix_bwd_workspace_bytes(int32_t window, int32_t H, int32_t hd) {
  return window > 0 && H > 0 && hd > 0 ? rt_v3_prefix_workspace_floats(window, H, hd) * sizeof(float) : 0;
}
int rt_mha_varlen_prefix_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o,
                             int64_t ldo, const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, int32_t B,
                             int32_t n_prefixed, int32_t H, int32_t hd, int32_t max_len, int32_t window, float p_drop, uint64_t seed,
                             float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv, float* delta, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || n_prefixed < 0 || n_prefixed >= B || window <= 0 ||
      max_len < window || dout == nullptr || lse == nullptr || dq == nullptr || dk == nullptr || dv == nullptr || delta == nullptr ||
      (lddo & 3) || (lddq & 3) || (lddk & 3) || (lddv & 3) || mis(dout) || mis(dq) || mis(dk) || mis(dv) || !(p_drop >= 0.f && p_drop < 1.f))
    return RT_ERR_INVALID_ARG;
  if (hd != 32 && hd != 64 && hd != 128) return RT_ERR_UNSUPPORTED;
  if (workspace == nullptr || mis(workspace) || workspace_bytes < rt_mha_varlen_prefix_bwd_workspace_bytes(window, H, hd)) return RT_ERR_WORKSPACE;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = const_cast<float*>(o); a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.H = H; a.hd = hd; a.window = window; a.prefix_sessions = n_prefixed;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = const_cast<float*>(lse);
  a.dout = dout; a.lddo = lddo; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.prefix_ws = reinterpret_cast<float*>(workspace);
  return rt_v3_varlen_bwd(a, max_len, stream);
}

// Bidirectional attention inside every packed session (no causal mask; BERT4Rec: the reference masks the pad keys of its window,
// torch_backbone.py:254, bert4rec.py:200 — packed rows have none).  lse != NULL: training forward (attention dropout p_drop / seed, lse
// [N, H] kept); lse == NULL: inference.  Served by the bf16-plane kernels only (hd 32 / 64, 2 * (max_len + 1) * 6 * hd bytes of LDS):
// RT_ERR_UNSUPPORTED otherwise — the caller keeps the padded window then.
int rt_mha_varlen_bidir_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                            int32_t B, int32_t H, int32_t hd, int32_t max_len, float p_drop, uint64_t seed, float* o, int64_t ldo, float* lse,
                            hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || !(p_drop >= 0.f && p_drop < 1.f)) return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.H = H; a.hd = hd; a.window = 0;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = lse != nullptr ? p_drop : 0.f; a.seed = seed; a.lse = lse;
  if ((v2_mode() & 1) && !(v2_mode() & 4)) {
    const int rc = rt_v2_bidir_fwd(a, max_len, lse != nullptr, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v3_bidir_fwd(a, max_len, lse != nullptr, stream);
}
int rt_mha_varlen_bidir_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                            const float* dout, int64_t lddo, const float* lse, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t hd,
                            int32_t max_len, float p_drop, uint64_t seed, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                            int64_t lddv, float* delta, hipStream_t stream) {
  (void)hipGetLastError();
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || dout == nullptr || lse == nullptr || dq == nullptr ||
      dk == nullptr || dv == nullptr || delta == nullptr || (lddo & 3) || (lddq & 3) || (lddk & 3) || (lddv & 3) || mis(dout) || mis(dq) ||
      mis(dk) || mis(dv) || !(p_drop >= 0.f && p_drop < 1.f))
    return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = const_cast<float*>(o); a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.B = B; a.H = H; a.hd = hd; a.window = 0;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = const_cast<float*>(lse);
  a.dout = dout; a.lddo = lddo; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  if ((v2_mode() & 2) && !(v2_mode() & 8)) {
    const int rc = rt_v2_bidir_bwd(a, max_len, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v3_bidir_bwd(a, max_len, stream);
}

// The last query of every session: q [B, ldq] (one row per session), k / v packed rows, o [B, ldo].  hd % 8 == 0, hd <= 256.
int rt_mha_varlen_last_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           const int64_t* cu_seqlens, const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd,
                           int32_t max_len, int32_t window, float* o, int64_t ldo, hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || (bk == nullptr) != (bv == nullptr) || hd > 256)
    return RT_ERR_INVALID_ARG;
  if (B == 0) return RT_OK;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd);
  const size_t prob_f = (size_t)((max_len + 3) & ~3) + 8, part_f = (size_t)VT * 4;
  const size_t lds = (prob_f > part_f ? prob_f : part_f) * sizeof(float);
  if (lds > VARLEN_LDS_LIMIT) return RT_ERR_UNSUPPORTED;
  { const int rc = set_lds(&attn_varlen_last_kernel, lds); if (rc != RT_OK) return rc; }
  attn_varlen_last_kernel<<<B * H, VT, lds, stream>>>(a, max_len);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// The last query of every session from the block input itself (see attn_last_x_kernel): qk [B, H, d] = W_k,h^T q_h per session and
// head, x packed rows [*, ldx], xbar [B, H, d] = sum_j softmax_j((qk . x_j) / sqrt(hd)) x_j.  pad_keys != 0: the window's pad keys take
// part with logit 0 (their x is zero).  prefix_row >= 0 (with pad_keys = 0): the batch carries the window's pad rows once, from that row on —
// a session of n rows sees rows [0, window - n) of them as keys in front of its own (LiGR without key-padding masks).  d in {64, 128, 256, 512}.
int rt_mha_varlen_last_x_fwd(const float* qk, const float* x, int64_t ldx, const int64_t* cu_seqlens, int32_t B, int32_t H, int32_t d,
                             int32_t max_len, int32_t window, int32_t pad_keys, int64_t prefix_row, float* xbar, hipStream_t stream) {
  (void)hipGetLastError();
  if (qk == nullptr || x == nullptr || cu_seqlens == nullptr || xbar == nullptr || B < 0 || H <= 0 || d <= 0 || d % H != 0 || (ldx & 3) != 0 ||
      max_len <= 0 || (prefix_row >= 0 && (pad_keys != 0 || max_len < window)))
    return RT_ERR_INVALID_ARG;
  if (d != 64 && d != 128 && d != 256 && d != 512) return RT_ERR_UNSUPPORTED;
  if (B == 0) return RT_OK;
  LastXArgs a{};
  a.prefix_row = prefix_row;
  a.qk = qk; a.x = x; a.ldx = ldx; a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.xbar = xbar;
  a.B = B; a.H = H; a.d = d; a.window = window; a.pads = pad_keys; a.scale = 1.0f / sqrtf((float)(d / H));
  switch (d) {
    case 64: return launch_last_x_h<1>(a, max_len, stream);
    case 128: return launch_last_x_h<2>(a, max_len, stream);
    case 256: return launch_last_x_h<4>(a, max_len, stream);
    default: return launch_last_x_h<8>(a, max_len, stream);
  }
}

// The head-expanded copy of a [d, d] projection weight for the two products around rt_mha_varlen_last_x_fwd (see expand_heads_kernel):
// qk = Q E(W_k) (E as [K = d, N = H d]), attention output = xbar E(W_v)^T + b_v (E as [N = d, K = H d]).
int rt_mha_last_x_expand(const float* W, int32_t d, int32_t H, float* E, hipStream_t stream) {
  (void)hipGetLastError();
  if (W == nullptr || E == nullptr || d <= 0 || H <= 0 || d % H != 0) return RT_ERR_INVALID_ARG;
  expand_heads_kernel<<<d, 256, 0, stream>>>(W, d, d / H, H, E);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
