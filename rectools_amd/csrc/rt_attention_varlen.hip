// K4v — softmax self-attention over PACKED (padding-free) sessions, forward only (the recommend() encoder).
//
// The reference runs every block on the left-padded [B, L] window (sasrec.py:186-231, :300; torch_backbone.py:245-260).  Here the
// rows of a batch are the REAL positions only: session b owns rows cu[b] .. cu[b+1]-1 of q / k / v / o (n_b = cu[b+1] - cu[b] items,
// oldest first).  What the padded window adds for a causal SASRec block is closed-form: a pad key row is b_k / b_v in every
// session and block (the block input is masked to 0) and every real query sees all `window - n_b` of them, so ONE virtual key per
// query — logit q.b_k / sqrt(hd), value b_v, multiplicity n_pad — reproduces the padded softmax (tests/test_packed_equivalence.py
// pins this on the CPU oracle, values and gradients).  With key-padding masks (BERT4Rec) pass bk = bv = null: pads do not exist.
//
//   rt_mha_varlen_fwd       every query of every session (causal), one workgroup per (session, head)
//   rt_mha_varlen_last_fwd  the LAST query of every session (the final block of recommend()), one workgroup per (session, head)
//
// Layout of the full kernel: everything is kept transposed so that a LANE owns a QUERY.  S^T = K Q^T (A = K rows from LDS, B = the
// query fragment in registers) leaves 16 keys of query `lane & 31` in each lane: row maxima / sums are in-lane plus one xor-32
// shuffle, the probabilities are fed back as the B operand of O^T = V^T P^T without leaving the registers (MFMA step r consumes
// the keys row_of(r, 0) and row_of(r, 1), which is exactly what the two lane halves hold in accumulator register r), and the
// rescale factor of the online softmax is a per-lane scalar for S^T, P^T and O^T alike.  v_mfma_f32_32x32x2_f32: exact fp32.
#include "rt_common.h"
#include "rt_varlen.h"
#include <stdlib.h>
#include <string.h>

namespace {
using namespace rt_varlen;

constexpr int VT = 256;   // threads per workgroup (4 waves)

__device__ __forceinline__ int row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int HD>
__global__ __launch_bounds__(VT) void attn_varlen_fwd_kernel(VarlenArgs a) {
  constexpr int KS = HD + 1;            // LDS row stride (odd: lanes that walk rows at a fixed column hit distinct banks)
  constexpr int NS = HD / 2;            // MFMA steps of one S^T tile (k = 2 per step)
  constexpr int NCB = HD / 32;          // 32-column blocks of the head dimension
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const int n32 = (n + 31) & ~31;
  float* Ks = smem;                     // [n32][KS]
  float* Vs = smem + (size_t)n32 * KS;  // [n32][KS]

  // ---- stage this (session, head)'s K and V rows (zero rows behind the session's end)
  for (int idx = tid; idx < n32 * (HD / 4); idx += VT) {
    const int r = idx / (HD / 4), c4 = idx % (HD / 4);
    f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {
      kk = *reinterpret_cast<const f32x4*>(a.k + (row0 + r) * a.ldk + h * HD + c4 * 4);
      vv = *reinterpret_cast<const f32x4*>(a.v + (row0 + r) * a.ldv + h * HD + c4 * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { Ks[r * KS + c4 * 4 + t] = kk[t]; Vs[r * KS + c4 * 4 + t] = vv[t]; }
  }
  __syncthreads();

  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const int n_qt = n32 / 32;
  for (int qt = wave; qt < n_qt; qt += 4) {
    const int qrow = qt * 32 + i;                                   // this lane's query (valid if < n)
    const float* qp = a.q + (row0 + (qrow < n ? qrow : n - 1)) * a.ldq + h * HD;
    float qf[NS];                                                   // B operand of S^T: Q[qrow][2s + half] * scale
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qp[2 * s + half] * a.scale;
    float m = -INFINITY, l = 0.f;
    f32x16 oT[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) oT[cb][r] = 0.f;

    for (int kt = 0; kt <= qt; ++kt) {
      // S^T[j][i] = sum_k K[kt*32 + j][k] * Q[i][k]:  A lane = K row (lane & 31), B lane = query (lane & 31)
      f32x16 sT;
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[r] = 0.f;
      const float* kp = Ks + (kt * 32 + i) * KS + half;
#pragma unroll
      for (int s = 0; s < NS; ++s) sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sT, 0, 0, 0);
      // causal + end-of-session mask, running maximum (16 keys in-lane, the other 16 in the partner lane)
      float mx = m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jr = kt * 32 + row_of(r, half);
        const bool ok = jr <= qrow && jr < n;
        sT[r] = ok ? sT[r] : -INFINITY;
        mx = fmaxf(mx, sT[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mx);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (sT[r] == -INFINITY) ? 0.f : __expf(sT[r] - mx);
        sT[r] = p;
        ps += p;
      }
      ps += __shfl_xor(ps, 32, 64);
      l = l * alpha + ps;
      m = mx;
      if (thr16 != 0u) {   // dropout acts on the normalised probabilities: the row sum above stays undropped
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sT[r] = drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, (unsigned)(kt * 32 + row_of(r, half)), thr16) ? sT[r] * inv_keep : 0.f;
      }
      // O^T[c][i] = alpha * O^T[c][i] + sum_j V[j][c] * P^T[j][i]:  step r consumes the keys row_of(r, 0) / row_of(r, 1)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[cb][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vp = Vs + (kt * 32 + row_of(r, half)) * KS + i;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          oT[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[cb * 32], sT[r], oT[cb], 0, 0, 0);
      }
    }

    if (pads) {   // the window's pad keys: one virtual key, logit q.b_k / sqrt(hd), value b_v, multiplicity n_pad
      float dp = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) dp += qf[s] * a.bk[h * HD + 2 * s + half];
      dp += __shfl_xor(dp, 32, 64);
      const float mx = fmaxf(m, dp);
      const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mx);
      const float e = __expf(dp - mx);
      l = l * alpha + (float)n_pad * e;
      m = mx;
      // value side: the pads that survive the dropout
      const float wv = thr16 != 0u ? (float)pads_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, n, n_pad, thr16) * inv_keep * e
                                   : (float)n_pad * e;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[cb][r] = oT[cb][r] * alpha + wv * a.bv[h * HD + cb * 32 + row_of(r, half)];
    }

    if (qrow < n) {
      if (a.lse != nullptr && half == 0) a.lse[(row0 + qrow) * a.H + h] = m + __logf(l);
      const float inv = l > 0.f ? 1.f / l : 0.f;
      float* op = a.o + (row0 + qrow) * a.ldo + h * HD;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {   // row_of(r..r+3, half) are 4 consecutive columns: one 16-byte store
          f32x4 o4 = {oT[cb][r] * inv, oT[cb][r + 1] * inv, oT[cb][r + 2] * inv, oT[cb][r + 3] * inv};
          *reinterpret_cast<f32x4*>(op + cb * 32 + row_of(r, half)) = o4;
        }
    }
  }
}

// ---- backward, pass 1: dQ (+ delta, + the pad keys' share of the value-bias gradient).  Same "a lane owns a query" layout as the
// forward: S^T and dP^T = V dO^T are recomputed per key tile, dS^T = P (drop * dP - delta) feeds dQ^T = K^T dS^T from registers.
template <int HD>
__global__ __launch_bounds__(VT) void attn_varlen_bwd_dq_kernel(VarlenArgs a) {
  constexpr int KS = HD + 1, NS = HD / 2, NCB = HD / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  float* dbv = a.dbv_part != nullptr ? a.dbv_part + (long long)b * a.H * HD + h * HD : nullptr;
  if (n <= 0) {
    if (dbv != nullptr && tid < HD) dbv[tid] = 0.f;
    return;
  }
  const int n32 = (n + 31) & ~31;
  float* Ks = smem;                        // [n32][KS]
  float* Vs = smem + (size_t)n32 * KS;     // [n32][KS]
  float* red = Vs + (size_t)n32 * KS;      // [4][HD] per-wave partials of the value-bias gradient
  for (int idx = tid; idx < n32 * (HD / 4); idx += VT) {
    const int r = idx / (HD / 4), c4 = idx % (HD / 4);
    f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {
      kk = *reinterpret_cast<const f32x4*>(a.k + (row0 + r) * a.ldk + h * HD + c4 * 4);
      vv = *reinterpret_cast<const f32x4*>(a.v + (row0 + r) * a.ldv + h * HD + c4 * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { Ks[r * KS + c4 * 4 + t] = kk[t]; Vs[r * KS + c4 * 4 + t] = vv[t]; }
  }
  __syncthreads();

  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  float dbv_acc[NCB][16];                  // this lane's queries' share of d_bv, columns cb*32 + row_of(r, half)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) dbv_acc[cb][r] = 0.f;

  const int n_qt = n32 / 32;
  for (int qt = wave; qt < n_qt; qt += 4) {
    const int qrow = qt * 32 + i;
    const bool qok = qrow < n;
    const long long grow = row0 + (qok ? qrow : n - 1);
    const float* qp = a.q + grow * a.ldq + h * HD;
    const float* dop = a.dout + grow * a.lddo + h * HD;
    const float* op = a.o + grow * a.ldo + h * HD;
    float qf[NS], dof[NS];                 // B operands: Q[qrow][2s + half] * scale, dO[qrow][2s + half]
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      qf[s] = qp[2 * s + half] * a.scale;
      dof[s] = qok ? dop[2 * s + half] : 0.f;
      dl += dof[s] * op[2 * s + half];
    }
    dl += __shfl_xor(dl, 32, 64);          // delta = rowsum(dO * O)
    const float lse = a.lse[grow * a.H + h];
    if (qok && half == 0) a.delta[grow * a.H + h] = dl;
    f32x16 dqT[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqT[cb][r] = 0.f;

    for (int kt = 0; kt <= qt; ++kt) {
      f32x16 sT, dpT;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dpT[r] = 0.f; }
      const float* kp = Ks + (kt * 32 + i) * KS + half;
      const float* vp = Vs + (kt * 32 + i) * KS + half;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        sT = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], qf[s], sT, 0, 0, 0);
        dpT = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * s], dof[s], dpT, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jr = kt * 32 + row_of(r, half);
        const bool ok = jr <= qrow && jr < n;
        const float p = ok ? __expf(sT[r] - lse) : 0.f;
        float dp = dpT[r];
        if (thr16 != 0u) dp = drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, (unsigned)jr, thr16) ? dp * inv_keep : 0.f;
        sT[r] = p * (dp - dl);             // dS^T
      }
      // dQ^T[c][i] += sum_j K[j][c] * dS^T[j][i]  (scale applied at the store)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* kr = Ks + (kt * 32 + row_of(r, half)) * KS + i;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          dqT[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[cb * 32], sT[r], dqT[cb], 0, 0, 0);
      }
    }

    if (pads) {   // the virtual pad key: dS_p = P_p (drop * dO.b_v - delta), dq += dS_p b_k, d_b_v += drop * P_p * dO
      float sp = 0.f, dpp = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        sp += qf[s] * a.bk[h * HD + 2 * s + half];
        dpp += dof[s] * a.bv[h * HD + 2 * s + half];
      }
      sp += __shfl_xor(sp, 32, 64);
      dpp += __shfl_xor(dpp, 32, 64);
      const float e = __expf(sp - lse);                                         // one pad key's probability
      const float kept = thr16 != 0u ? (float)pads_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, n, n_pad, thr16) * inv_keep
                                     : (float)n_pad;
      const float dsp = e * (kept * dpp - (float)n_pad * dl);
      const float wv = qok ? e * kept : 0.f;                                    // dropped pad mass that multiplied b_v
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqT[cb][r] += dsp * a.bk[h * HD + cb * 32 + row_of(r, half)];
      if (dbv != nullptr) {
        // d_bv[c] += wv * dO[qrow][c] for the columns this lane stores: dO is re-read in the dQ^T layout
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) dbv_acc[cb][r] += wv * dop[cb * 32 + row_of(r, half)];
      }
    }

    if (qok) {
      float* dqp = a.dq + grow * a.lddq + h * HD;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          f32x4 g = {dqT[cb][r] * a.scale, dqT[cb][r + 1] * a.scale, dqT[cb][r + 2] * a.scale, dqT[cb][r + 3] * a.scale};
          *reinterpret_cast<f32x4*>(dqp + cb * 32 + row_of(r, half)) = g;
        }
    }
  }

  if (dbv != nullptr) {   // reduce d_bv over the queries: the 32 lanes of a half, then the 4 waves through LDS
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = dbv_acc[cb][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (i == 0) red[wave * HD + cb * 32 + row_of(r, half)] = v;
      }
    __syncthreads();
    if (tid < HD) dbv[tid] = (red[tid] + red[HD + tid]) + (red[2 * HD + tid] + red[3 * HD + tid]);
  }
}

// ---- backward, pass 2: dK, dV.  A lane owns a KEY: S = Q K^T and dP = dO V^T (A = Q / dO rows from LDS, B = the key's K / V
// fragment in registers) leave 16 queries of key `lane & 31` per lane; P~ and dS feed dV^T = dO^T P~ and dK^T = Q^T dS from registers.
template <int HD>
__global__ __launch_bounds__(VT) void attn_varlen_bwd_dkv_kernel(VarlenArgs a) {
  constexpr int KS = HD + 1, NS = HD / 2, NCB = HD / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const int n32 = (n + 31) & ~31;
  float* Qs = smem;                         // [n32][KS] (already scaled by 1/sqrt(hd))
  float* Ds = smem + (size_t)n32 * KS;      // [n32][KS] dO
  float* Ls = Ds + (size_t)n32 * KS;        // [n32] lse
  float* Dl = Ls + n32;                     // [n32] delta
  for (int idx = tid; idx < n32 * (HD / 4); idx += VT) {
    const int r = idx / (HD / 4), c4 = idx % (HD / 4);
    f32x4 qq = {0.f, 0.f, 0.f, 0.f}, dd = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {
      qq = *reinterpret_cast<const f32x4*>(a.q + (row0 + r) * a.ldq + h * HD + c4 * 4) * a.scale;
      dd = *reinterpret_cast<const f32x4*>(a.dout + (row0 + r) * a.lddo + h * HD + c4 * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { Qs[r * KS + c4 * 4 + t] = qq[t]; Ds[r * KS + c4 * 4 + t] = dd[t]; }
  }
  for (int r = tid; r < n32; r += VT) {
    Ls[r] = r < n ? a.lse[(row0 + r) * a.H + h] : 0.f;
    Dl[r] = r < n ? a.delta[(row0 + r) * a.H + h] : 0.f;
  }
  __syncthreads();

  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const int n_kt = n32 / 32;
  for (int kt = wave; kt < n_kt; kt += 4) {
    const int krow = kt * 32 + i;            // this lane's key (valid if < n)
    const long long grow = row0 + (krow < n ? krow : n - 1);
    const float* kp = a.k + grow * a.ldk + h * HD;
    const float* vp = a.v + grow * a.ldv + h * HD;
    float kf[NS], vf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { kf[s] = kp[2 * s + half]; vf[s] = vp[2 * s + half]; }
    f32x16 dkT[NCB], dvT[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkT[cb][r] = 0.f; dvT[cb][r] = 0.f; }

    for (int qt = kt; qt < n_kt; ++qt) {     // causal: queries at or behind the key
      f32x16 sm, dpm;                        // S[q][key], dP[q][key]: lane = key, register r = query row_of(r, half)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sm[r] = 0.f; dpm[r] = 0.f; }
      const float* qa = Qs + (qt * 32 + i) * KS + half;
      const float* da = Ds + (qt * 32 + i) * KS + half;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        sm = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * s], kf[s], sm, 0, 0, 0);
        dpm = __builtin_amdgcn_mfma_f32_32x32x2f32(da[2 * s], vf[s], dpm, 0, 0, 0);
      }
      f32x16 pd;                             // dropped probabilities (for dV)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = qt * 32 + row_of(r, half);
        const bool ok = krow <= qr && qr < n && krow < n;
        const float p = ok ? __expf(sm[r] - Ls[qr]) : 0.f;
        float keepf = 1.f;
        if (thr16 != 0u) keepf = drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qr, (unsigned)krow, thr16) ? inv_keep : 0.f;
        pd[r] = p * keepf;
        sm[r] = p * (dpm[r] * keepf - Dl[qr]);   // dS
      }
      // dV^T[c][key] += sum_q dO[q][c] * P~[q][key];  dK^T[c][key] += sum_q Q[q][c] * dS[q][key]   (Q is pre-scaled)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* dr = Ds + (qt * 32 + row_of(r, half)) * KS + i;
        const float* qr2 = Qs + (qt * 32 + row_of(r, half)) * KS + i;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          dvT[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[cb * 32], pd[r], dvT[cb], 0, 0, 0);
          dkT[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(qr2[cb * 32], sm[r], dkT[cb], 0, 0, 0);
        }
      }
    }

    if (krow < n) {
      float* dkp = a.dk + grow * a.lddk + h * HD;
      float* dvp = a.dv + grow * a.lddv + h * HD;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          f32x4 gk = {dkT[cb][r], dkT[cb][r + 1], dkT[cb][r + 2], dkT[cb][r + 3]};
          f32x4 gv = {dvT[cb][r], dvT[cb][r + 1], dvT[cb][r + 2], dvT[cb][r + 3]};
          *reinterpret_cast<f32x4*>(dkp + cb * 32 + row_of(r, half)) = gk;
          *reinterpret_cast<f32x4*>(dvp + cb * 32 + row_of(r, half)) = gv;
        }
    }
  }
}

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

// Default: the streamed bf16-plane kernels of rt_attention_v3.hip (hd 32 / 64, any session length).  RT_VARLEN_IMPL (A/B runs): v2 = the
// whole-session-image kernels of rt_attention_v2.hip wherever they serve the shape (they answer RT_ERR_UNSUPPORTED otherwise), v1 = the
// first-form kernels of this file (f32-input MFMA); v2fwd / v2bwd / v3fwd / v3bwd: only that pass on the named family, v1 / v2 for the other.
int v2_mode() {   // bit 0 / 1: v2 forward / backward, bit 2 / 3: v3 forward / backward
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("RT_VARLEN_IMPL");
    mode = (e == nullptr || e[0] == 0) ? 15
           : !strcmp(e, "v1") ? 0 : !strcmp(e, "v2") ? 3 : !strcmp(e, "v2fwd") ? 1 : !strcmp(e, "v2bwd") ? 2
           : !strcmp(e, "v3fwd") ? 3 | 4 : !strcmp(e, "v3bwd") ? 3 | 8 : 15;
  }
  return mode;
}

template <typename K>
int set_lds(K kernel, size_t lds) {
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return RT_OK;
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
// [h*hd, (h+1)*hd); cu_seqlens [B+1] (device, int64, ascending, cu[0] = first row); max_len >= the longest session (sizes the LDS
// image: 2 * roundup32(max_len) * (hd + 1) floats — RT_ERR_UNSUPPORTED beyond 160 KB, the caller then takes the padded path);
// bk / bv [H*hd]: the key / value projection biases, i.e. the reference's pad key / value row, or null when pad keys are masked;
// window: the reference's session_max_len (n_pad = window - n_b virtual pad keys per query).  hd in {32, 64}.
int rt_mha_varlen_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const int64_t* cu_seqlens,
                      const float* bk, const float* bv, int32_t B, int32_t H, int32_t hd, int32_t max_len, int32_t window, float* o,
                      int64_t ldo, hipStream_t stream) {
  (void)hipGetLastError();
  if (bad_args(q, k, v, o, ldq, ldk, ldv, ldo, cu_seqlens, B, H, hd, max_len) || (bk == nullptr) != (bv == nullptr))
    return RT_ERR_INVALID_ARG;
  if (B == 0 || max_len == 0) return RT_OK;
  if (hd != 32 && hd != 64) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd);
  if (v2_mode() & 4) {
    const int rc = rt_v3_varlen_fwd(a, max_len, false, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  if (v2_mode() & 1) {
    const int rc = rt_v2_varlen_fwd(a, max_len, false, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  const size_t lds = (size_t)2 * ((max_len + 31) & ~31) * (hd + 1) * sizeof(float);
  if (lds > VARLEN_LDS_LIMIT) return RT_ERR_UNSUPPORTED;
  if (hd == 64) {
    const int rc = set_lds(&attn_varlen_fwd_kernel<64>, lds); if (rc != RT_OK) return rc;
    attn_varlen_fwd_kernel<64><<<B * H, VT, lds, stream>>>(a);
  } else {
    const int rc = set_lds(&attn_varlen_fwd_kernel<32>, lds); if (rc != RT_OK) return rc;
    attn_varlen_fwd_kernel<32><<<B * H, VT, lds, stream>>>(a);
  }
  RT_CHECK_LAUNCH();
  return RT_OK;
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
  if (hd != 32 && hd != 64) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = lse;
  if (v2_mode() & 4) {
    const int rc = rt_v3_varlen_fwd(a, max_len, true, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  if (v2_mode() & 1) {
    const int rc = rt_v2_varlen_fwd(a, max_len, true, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  const size_t lds = (size_t)2 * ((max_len + 31) & ~31) * (hd + 1) * sizeof(float);
  if (lds > VARLEN_LDS_LIMIT) return RT_ERR_UNSUPPORTED;
  if (hd == 64) {
    const int rc = set_lds(&attn_varlen_fwd_kernel<64>, lds); if (rc != RT_OK) return rc;
    attn_varlen_fwd_kernel<64><<<B * H, VT, lds, stream>>>(a);
  } else {
    const int rc = set_lds(&attn_varlen_fwd_kernel<32>, lds); if (rc != RT_OK) return rc;
    attn_varlen_fwd_kernel<32><<<B * H, VT, lds, stream>>>(a);
  }
  RT_CHECK_LAUNCH();
  return RT_OK;
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
  if (hd != 32 && hd != 64) return RT_ERR_UNSUPPORTED;
  VarlenArgs a{};
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = const_cast<float*>(o); a.ldo = ldo;
  a.cu = reinterpret_cast<const long long*>(cu_seqlens); a.bk = bk; a.bv = bv; a.B = B; a.H = H; a.hd = hd; a.window = window;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.lse = const_cast<float*>(lse);
  a.dout = dout; a.lddo = lddo; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.dbv_part = dbv_part;
  if (v2_mode() & 8) {
    const int rc = rt_v3_varlen_bwd(a, max_len, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  if (v2_mode() & 2) {
    const int rc = rt_v2_varlen_bwd(a, max_len, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  const size_t n32 = (size_t)((max_len + 31) & ~31);
  const size_t lds_dq = (2 * n32 * (hd + 1) + 4 * (size_t)hd) * sizeof(float);
  const size_t lds_kv = (2 * n32 * (hd + 1) + 2 * n32) * sizeof(float);
  if (lds_dq > VARLEN_LDS_LIMIT || lds_kv > VARLEN_LDS_LIMIT) return RT_ERR_UNSUPPORTED;
  if (hd == 64) {
    { const int rc = set_lds(&attn_varlen_bwd_dq_kernel<64>, lds_dq); if (rc != RT_OK) return rc; }
    attn_varlen_bwd_dq_kernel<64><<<B * H, VT, lds_dq, stream>>>(a);
    RT_CHECK_LAUNCH();
    { const int rc = set_lds(&attn_varlen_bwd_dkv_kernel<64>, lds_kv); if (rc != RT_OK) return rc; }
    attn_varlen_bwd_dkv_kernel<64><<<B * H, VT, lds_kv, stream>>>(a);
  } else {
    { const int rc = set_lds(&attn_varlen_bwd_dq_kernel<32>, lds_dq); if (rc != RT_OK) return rc; }
    attn_varlen_bwd_dq_kernel<32><<<B * H, VT, lds_dq, stream>>>(a);
    RT_CHECK_LAUNCH();
    { const int rc = set_lds(&attn_varlen_bwd_dkv_kernel<32>, lds_kv); if (rc != RT_OK) return rc; }
    attn_varlen_bwd_dkv_kernel<32><<<B * H, VT, lds_kv, stream>>>(a);
  }
  RT_CHECK_LAUNCH();
  return RT_OK;
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
  if (v2_mode() & 4) {
    const int rc = rt_v3_bidir_fwd(a, max_len, lse != nullptr, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v2_bidir_fwd(a, max_len, lse != nullptr, stream);
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
  if (v2_mode() & 8) {
    const int rc = rt_v3_bidir_bwd(a, max_len, stream);
    if (rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return rt_v2_bidir_bwd(a, max_len, stream);
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

}  // extern "C"
