// K4v2 — softmax self-attention over PACKED sessions on the bf16 matrix pipe, fp32-accurate.
//
// Same contract as rt_attention_varlen.hip (causal attention inside every session of a packed batch, the reference's left-pad keys as
// ONE virtual key per query: sasrec.py:186-231, torch_backbone.py:245-260), different arithmetic and geometry:
//
//  * every fp32 product is six v_mfma_f32_16x16x32_bf16 products of an EXACT three-way bf16 split (x = h + m + l, the GEMM's scheme,
//    rt_gemm.hip): terms hh, hm, mh, mm, hl, lh accumulate in fp32; the dropped terms are <= 2 * 2^-24 |a||b| — one fp32 rounding.
//    The f32-input MFMA the first kernels used runs at the fp32 VECTOR rate on the FMA lanes (MI355X_MICROARCH.md): its time ADDS to the
//    softmax arithmetic; the bf16 pipe is 16x faster per flop (6/16 after the split) and runs beside the VALU.
//  * operands are split ONCE, when a (session, head)'s rows are staged into LDS, into three bf16 planes per row (h | m | l, row-interleaved:
//    row r, plane p at r * 3 * ROWB + p * ROWB): no split arithmetic inside the tile loop except for the 8 probabilities a lane produces.
//  * an LDS image serves BOTH operand orientations: rows as MFMA rows through ds_read_b128 (S = K Q^T, dP = V dO^T), and rows as the
//    reduction index through ds_read_b64_tr_b16, the gfx950 transpose read (O^T = V^T P^T, dQ^T = K^T dS^T, dK^T = Q^T dS, dV^T = dO^T P):
//    lane i of a 16-lane group receives column i of the [4 rows][16 columns] block whose 16 eight-byte chunks the group's lanes address
//    (result element e of lane i = element (i & 3) of the chunk supplied by lane 4e + (i >> 2); probed on hardware,
//    scripts/microbench/tr16_probe.hip).  An XOR swizzle of the 16-byte unit index by row bits 1-2 makes both patterns bank-conflict
//    free (scripts/attn/swizzle_search.py, bank model of MI355X_MICROARCH.md §LDS).
//  * tiles: 16 OWNER rows per wave-tile (a lane owns one query — or, in the dK/dV pass, one key — four lanes share it and hold
//    different reduction slots), 32 partner rows per step.  v_mfma_f32_16x16x32_bf16 leaves the 4 x 2 scores of a lane in registers in
//    exactly the slot order the next product's B operand wants, so probabilities / dS never leave the registers.  16-row owner tiles
//    deal a causal triangle to 8 waves within 15 % of even (13 tiles of weight 1..7 at 200 rows) without merging partial results.
//  * rows behind a session's end read a ZERO row kept behind every image (index n): a partner tile never needs a bounds branch.
#include "rt_attn_planes.h"

namespace {
using namespace rt_varlen;
using namespace rt_planes;

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: a lane owns a query (4 lanes per query hold different keys / different head-dim columns)
// ---------------------------------------------------------------------------------------------------------------------------------
// CAUSAL = false: every query sees every key of its session (BERT4Rec: the reference masks the window's pad keys, bert4rec.py:200 /
// torch_backbone.py:254 — on packed rows they simply do not exist); no virtual pad key then.
template <int HD, int NW, bool TRAIN, bool CAUSAL = true>
__global__ __launch_bounds__(NW * 64) void v2_fwd_kernel(VarlenArgs a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + (size_t)(n + 1) * L::ROW3;
  if (!RT_ABL(a, 1))
    stage_images2<HD>(a.k + row0 * a.ldk + h * HD, a.ldk, 1.f, Kimg, a.v + row0 * a.ldv + h * HD, a.ldv, 1.f, Vimg, n, tid, NW * 64);
  __syncthreads();
  if (RT_ABL(a, 2)) return;

  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = CAUSAL && a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = TRAIN ? drop_thr16(a.p_drop) : 0u;
  const float inv_keep = (TRAIN && a.p_drop > 0.f) ? 1.f / (1.f - a.p_drop) : 1.f;
  const float qscale = a.scale * LOG2E;          // scores live in the base-2 domain: p = exp2(s' - m')

  for_my_tiles<NW, CAUSAL>(wave, (n + 15) >> 4, [&](int qt) {
    const int qrow = qt * 16 + i;
    const bool qok = qrow < n;
    const long long grow = row0 + (qok ? qrow : n - 1);
    P3 Qp[L::NS];
    if (!RT_ABL(a, 64)) load_owner_planes<HD>(a.q + grow * a.ldq + h * HD, g, qscale, Qp);
    float m = -INFINITY, lsum = 0.f;             // lsum: this lane's share of the row sum (its own keys), reduced at the end
    f32x4 oT[L::NCB];
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) oT[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_last = CAUSAL ? (qt * 16 + 15) >> 5 : (n - 1) >> 5;

    for (int kt = 0; kt <= kt_last; ++kt) {
      f32x4 sT[2];
      rows_times_owner<HD>(Kimg, kt * 32, n, Qp, i, g, sT, RT_ABLV(a));      // sT[kb][r]: key kt*32 + 16 kb + 4 g + r
      float sc[8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[4 * kb + r] = sT[kb][r];
      if (!RT_ABL(a, 4)) {
      if (kt == kt_last) {                                       // the causal edge (keys behind the session's end lie behind it too)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = kt * 32 + 16 * (e >> 2) + 4 * g + (e & 3);
          sc[e] = (CAUSAL ? key <= qrow : key < n) ? sc[e] : -INFINITY;
        }
      }
      float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
      mx = fmaxf(m, quad_max(mx));                               // finite: every query sees key kt*32 of every tile it visits
      const float alpha = __builtin_amdgcn_exp2f(m - mx);
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = __builtin_amdgcn_exp2f(sc[e] - mx); ps += sc[e]; }
      lsum = lsum * alpha + ps;
      m = mx;
      if (TRAIN && thr16 != 0u) {      // dropout acts on the normalised probabilities: the row sum above stays undropped
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const unsigned key = (unsigned)(kt * 32 + 16 * (e >> 2) + 4 * g + (e & 3));
          const unsigned hsh = drop_hash(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, key >> 1);
          sc[e] = (hsh & 0xFFFFu) >= thr16 ? sc[e] * inv_keep : 0.f;
          sc[e + 1] = (hsh >> 16) >= thr16 ? sc[e + 1] * inv_keep : 0.f;
        }
      }
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) oT[cb] *= alpha;
      }
      const P3 Pp = RT_SPLIT8(a, sc);
      cols_times_slots<HD>(Vimg, kt * 32, n, Pp, i, g, oT, RT_ABLV(a));      // oT[cb][r]: column 16 cb + 4 g + r of query `qrow`
    }

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
        int kept = 0;
        for (int kk = n + g; kk < n + n_pad; kk += 4) kept += drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, (unsigned)kk, thr16) ? 1 : 0;
        kept += __shfl_xor(kept, 16, 64);
        kept += __shfl_xor(kept, 32, 64);
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
  });
}


// ---------------------------------------------------------------------------------------------------------------------------------
// backward, pass 1: dQ, delta = rowsum(dO * O), and the pad keys' share of the value-bias gradient.  A lane owns a query, as in the
// forward: S^T and dP^T = V dO^T are recomputed per key tile (K and V rows as MFMA rows), dS^T = P (drop * dP - delta) stays in
// registers and feeds dQ^T = K^T dS^T (K rows as the reduction index: transpose reads of the SAME image).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, int NW, bool CAUSAL = true>
__global__ __launch_bounds__(NW * 64) void v2_bwd_dq_kernel(VarlenArgs a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  float* dbv = a.dbv_part != nullptr ? a.dbv_part + (long long)b * a.H * HD + h * HD : nullptr;
  if (n <= 0) {
    if (dbv != nullptr && tid < HD) dbv[tid] = 0.f;
    return;
  }
  unsigned char* Kimg = smem;
  unsigned char* Vimg = smem + (size_t)(n + 1) * L::ROW3;
  if (!RT_ABL(a, 1))
    stage_images2<HD>(a.k + row0 * a.ldk + h * HD, a.ldk, 1.f, Kimg, a.v + row0 * a.ldv + h * HD, a.ldv, 1.f, Vimg, n, tid, NW * 64);
  __syncthreads();
  if (RT_ABL(a, 2)) return;

  const int n_pad = a.window > n ? a.window - n : 0;
  const bool pads = CAUSAL && a.bk != nullptr && a.bv != nullptr && n_pad > 0;
  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float qscale = a.scale * LOG2E;
  f32x4 dbv_acc[L::NCB];                  // this lane's queries' share of d_bv, columns 16 cb + 4 g + r
#pragma unroll
  for (int cb = 0; cb < L::NCB; ++cb) dbv_acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for_my_tiles<NW, CAUSAL>(wave, (n + 15) >> 4, [&](int qt) {
    const int qrow = qt * 16 + i;
    const bool qok = qrow < n;
    const long long grow = row0 + (qok ? qrow : n - 1);
    const float* qp = a.q + grow * a.ldq + h * HD;
    const float* dop = a.dout + grow * a.lddo + h * HD;
    const float* op = a.o + grow * a.ldo + h * HD;
    P3 Qp[L::NS], Dp[L::NS];
    float dl = 0.f, lse2 = 0.f;             // delta = rowsum(dO * O): 16 of the HD columns per lane
    if (!RT_ABL(a, 64)) {
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
    const int kt_last = CAUSAL ? (qt * 16 + 15) >> 5 : (n - 1) >> 5;

    for (int kt = 0; kt <= kt_last; ++kt) {
      f32x4 sT[2], dpT[2];
      rows_times_owner<HD>(Kimg, kt * 32, n, Qp, i, g, sT, RT_ABLV(a));
      rows_times_owner<HD>(Vimg, kt * 32, n, Dp, i, g, dpT, RT_ABLV(a));
      float ds[8];
      if (RT_ABL(a, 4)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ds[e] = sT[e >> 2][e & 3] + dpT[e >> 2][e & 3];
      } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = kt * 32 + 16 * (e >> 2) + 4 * g + (e & 3);
        const float pr = (kt < kt_last || (CAUSAL ? key <= qrow : key < n)) ? __builtin_amdgcn_exp2f(sT[e >> 2][e & 3] - lse2) : 0.f;
        ds[e] = pr;
      }
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float d0 = dpT[e >> 2][e & 3], d1 = dpT[e >> 2][(e & 3) + 1];
        if (thr16 != 0u) {
          const unsigned key = (unsigned)(kt * 32 + 16 * (e >> 2) + 4 * g + (e & 3));
          const unsigned hsh = drop_hash(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, key >> 1);
          d0 = (hsh & 0xFFFFu) >= thr16 ? d0 * inv_keep : 0.f;
          d1 = (hsh >> 16) >= thr16 ? d1 * inv_keep : 0.f;
        }
        ds[e] *= d0 - dl;                   // dS^T
        ds[e + 1] *= d1 - dl;
      }
      }
      const P3 Sp = RT_SPLIT8(a, ds);
      cols_times_slots<HD>(Kimg, kt * 32, n, Sp, i, g, dqT, RT_ABLV(a));   // dQ^T[c][q] += sum_j K[j][c] dS^T[j][q]  (scale at the store)
    }

    if (pads) {   // the virtual pad key: dS_p = P_p (drop * dO.b_v - delta), dq += dS_p b_k, d_b_v += drop * P_p * dO
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
        int kc = 0;
        for (int kk = n + g; kk < n + n_pad; kk += 4) kc += drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qrow, (unsigned)kk, thr16) ? 1 : 0;
        kc += __shfl_xor(kc, 16, 64);
        kc += __shfl_xor(kc, 32, 64);
        kept = (float)kc * inv_keep;
      }
      const float dsp = e1 * (kept * dpp - (float)n_pad * dl);
      const float wv = qok ? e1 * kept : 0.f;                                    // dropped pad mass that multiplied b_v
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) {
        const f32x4 bk4 = *reinterpret_cast<const f32x4*>(a.bk + h * HD + 16 * cb + 4 * g);
        dqT[cb] += bk4 * dsp;
        if (dbv != nullptr) dbv_acc[cb] += *reinterpret_cast<const f32x4*>(dop + 16 * cb + 4 * g) * wv;
      }
    }

    if (qok && !RT_ABL(a, 128)) {
      float* dqp = a.dq + grow * a.lddq + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(dqp + 16 * cb + 4 * g) = dqT[cb] * a.scale;
    }
  });

  if (dbv != nullptr) {   // reduce d_bv over the queries: the 16 lanes of a group, then the waves through LDS (the images are dead)
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = dbv_acc[cb][r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        dbv_acc[cb][r] = v;
      }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [NW][HD]
    if (i == 0)
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(red + wave * HD + 16 * cb + 4 * g) = dbv_acc[cb];
    __syncthreads();
    if (tid < HD) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * HD + tid];
      dbv[tid] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV.  A lane owns a KEY: S = Q K^T and dP = dO V^T (Q / dO rows as MFMA rows, the key's K / V fragment in
// registers) leave 2 x 4 queries of key `lane & 15` per lane; P~ and dS feed dV^T = dO^T P~ and dK^T = Q^T dS through transpose reads of
// the same two images.  Q is staged pre-scaled by log2(e) / sqrt(hd): dK comes out times log2(e) and is scaled back at the store.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int HD, int NW, bool CAUSAL = true>
__global__ __launch_bounds__(NW * 64) void v2_bwd_dkv_kernel(VarlenArgs a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const int n32 = (n + 31) & ~31;
  unsigned char* Qimg = smem;
  unsigned char* Dimg = smem + (size_t)(n + 1) * L::ROW3;
  float* Ls = reinterpret_cast<float*>(smem + 2 * (size_t)(n + 1) * L::ROW3);   // [n32] lse * log2(e)   (16-byte aligned: ROW3 % 16 == 0)
  float* Dl = Ls + n32;                                                         // [n32] delta
  const float qscale = a.scale * LOG2E;
  if (!RT_ABL(a, 1)) {
  stage_images2<HD>(a.q + row0 * a.ldq + h * HD, a.ldq, qscale, Qimg, a.dout + row0 * a.lddo + h * HD, a.lddo, 1.f, Dimg, n, tid, NW * 64);
  for (int r = tid; r < n32; r += NW * 64) {
    Ls[r] = r < n ? a.lse[(row0 + r) * a.H + h] * LOG2E : 0.f;
    Dl[r] = r < n ? a.delta[(row0 + r) * a.H + h] : 0.f;
  }
  }
  __syncthreads();
  if (RT_ABL(a, 2)) return;

  const unsigned thr16 = drop_thr16(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const int qt_last = (n - 1) >> 5;

  for_my_tiles<NW, false>(wave, (n + 15) >> 4, [&](int kt) {   // (bidirectional: every key tile weighs the same, any order is even)
    const int krow = kt * 16 + i;            // this lane's key (valid if < n)
    const bool kok = krow < n;
    const long long grow = row0 + (kok ? krow : n - 1);
    P3 Kp[L::NS], Vp[L::NS];
    if (!RT_ABL(a, 64)) {
    load_owner_planes<HD>(a.k + grow * a.ldk + h * HD, g, 1.f, Kp);
    load_owner_planes<HD>(a.v + grow * a.ldv + h * HD, g, 1.f, Vp);
    }
    f32x4 dkT[L::NCB], dvT[L::NCB];
#pragma unroll
    for (int cb = 0; cb < L::NCB; ++cb) { dkT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int qt_first = CAUSAL ? (kt * 16) >> 5 : 0;

    for (int qt = qt_first; qt <= qt_last; ++qt) {     // causal: query tiles at or behind the key tile
      f32x4 sm[2], dpm[2];                             // S[q][key], dP[q][key]: register (qb, r) = query qt*32 + 16 qb + 4 g + r
      rows_times_owner<HD>(Qimg, qt * 32, n, Kp, i, g, sm, RT_ABLV(a));
      rows_times_owner<HD>(Dimg, qt * 32, n, Vp, i, g, dpm, RT_ABLV(a));
      const bool edge = qt == qt_first || qt == qt_last;
      float pd[8], ds[8];
      if (RT_ABL(a, 4)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { pd[e] = sm[e >> 2][e & 3]; ds[e] = dpm[e >> 2][e & 3]; }
      } else
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const f32x4 ls4 = *reinterpret_cast<const f32x4*>(Ls + qt * 32 + 16 * qb + 4 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(Dl + qt * 32 + 16 * qb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qr = qt * 32 + 16 * qb + 4 * g + r;
          float pr = __builtin_amdgcn_exp2f(sm[qb][r] - ls4[r]);
          if (edge) pr = ((!CAUSAL || krow <= qr) && qr < n && kok) ? pr : 0.f;
          float keepf = 1.f;
          if (thr16 != 0u)
            keepf = drop_kept(a.seed, (unsigned)blockIdx.x, (unsigned)qr, (unsigned)krow, thr16) ? inv_keep : 0.f;
          pd[4 * qb + r] = pr * keepf;                               // dropped probabilities (for dV)
          ds[4 * qb + r] = pr * (dpm[qb][r] * keepf - dl4[r]);       // dS
        }
      }
      const P3 Pp = RT_SPLIT8(a, pd);
      cols_times_slots<HD>(Dimg, qt * 32, n, Pp, i, g, dvT, RT_ABLV(a));    // dV^T[c][key] += sum_q dO[q][c] P~[q][key]
      const P3 Sp = RT_SPLIT8(a, ds);
      cols_times_slots<HD>(Qimg, qt * 32, n, Sp, i, g, dkT, RT_ABLV(a));    // dK^T[c][key] += sum_q Q'[q][c] dS[q][key]
    }

    if (kok && !RT_ABL(a, 128)) {
      float* dkp = a.dk + grow * a.lddk + h * HD;
      float* dvp = a.dv + grow * a.lddv + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) {
        *reinterpret_cast<f32x4*>(dkp + 16 * cb + 4 * g) = dkT[cb] * LN2;     // Q' = Q * scale * log2(e): scale is in, log2(e) comes out
        *reinterpret_cast<f32x4*>(dvp + 16 * cb + 4 * g) = dvT[cb];
      }
    }
  });
}

#ifdef RT_ABLATION_BUILD
inline int v2_ablate_env() { const char* e = getenv("RT_V2_ABLATE"); return e != nullptr ? atoi(e) : 0; }
#endif

template <int HD, int NW, bool CAUSAL = true>
int launch_bwd(const VarlenArgs& a, int max_len, hipStream_t stream) {
  const size_t img = 2 * Lay<HD>::image_bytes(max_len);
  const size_t lds_dq = img > (size_t)NW * HD * 4 ? img : (size_t)NW * HD * 4;
  const size_t lds_kv = img + 2 * (size_t)((max_len + 31) & ~31) * sizeof(float);
  if (lds_dq > 160 * 1024 || lds_kv > 160 * 1024) return RT_ERR_UNSUPPORTED;
  auto kq = &v2_bwd_dq_kernel<HD, NW, CAUSAL>;
  auto kkv = &v2_bwd_dkv_kernel<HD, NW, CAUSAL>;
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dq));
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
#ifdef RT_ABLATION_BUILD
  VarlenArgs b = a;
  b.ablate = v2_ablate_env();
  if (!(b.ablate & 256)) kq<<<a.B * a.H, NW * 64, lds_dq, stream>>>(b);
  if (!(b.ablate & 512)) kkv<<<a.B * a.H, NW * 64, lds_kv, stream>>>(b);
  RT_CHECK_LAUNCH();
  return RT_OK;
#endif
  kq<<<a.B * a.H, NW * 64, lds_dq, stream>>>(a);
  RT_CHECK_LAUNCH();
  kkv<<<a.B * a.H, NW * 64, lds_kv, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

template <int HD, int NW, bool TRAIN, bool CAUSAL = true>
int launch_fwd(const VarlenArgs& a, int max_len, hipStream_t stream) {
  const size_t lds = 2 * Lay<HD>::image_bytes(max_len);
  if (lds > 160 * 1024) return RT_ERR_UNSUPPORTED;
  auto kern = &v2_fwd_kernel<HD, NW, TRAIN, CAUSAL>;
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef RT_ABLATION_BUILD
  VarlenArgs b = a;
  b.ablate = v2_ablate_env();
  kern<<<a.B * a.H, NW * 64, lds, stream>>>(b);
  RT_CHECK_LAUNCH();
  return RT_OK;
#endif
  kern<<<a.B * a.H, NW * 64, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}


// =================================================================================================================================
// K6v2 — HSTU's pointwise attention (hstu.py:270-288: silu(q k^T + rab) / L, causal, the relative time / position bias rab of
// hstu.py:84-128 computed in-kernel) on the geometry above.  No softmax: a query's output is a plain SUM over its keys, so a session
// longer than one LDS image is walked in CHUNKS of HCH partner rows — the owner rows' accumulators are carried through memory between
// chunks (each lane re-reads what it wrote itself), nothing has to be renormalised.  The f32-input ring kernels (rt_attention.hip) spend
// 64 quarter-rate matrix instructions per 32 x 32 tile pair and split nothing; here the K / V (Q / dO) rows are split once per chunk
// and every product is six bf16 instructions.  First measured at the very end of round 4: the C4-shaped HSTU step 6.81 -> 5.93 ms
// (18.8 -> 21.6 k seqs/s; forward 0.27 -> 0.18 ms per layer), parity against the padded ring kernels and the oracle on the first run
// (tests/test_packed_hstu_gpu.py parametrised over both; RT_HSTU_ATTN=ring keeps the ring kernels).  Untuned: chunk size, the bias
// arithmetic per element (~40 VALU) and the dK/dV pass's 220 registers are where the next factor is.
// =================================================================================================================================
constexpr int HCH = 192;         // partner rows per chunk (6 tiles of 32): two hd-64 images = 148 KB
// LDS behind the two images: the chunk's partner timestamps, the three tables, (backward) the two bias-gradient accumulators
struct HstuLdsV2 {
  unsigned char* img0; unsigned char* img1;
  long long* ts_p;       // [HCH] timestamps of the chunk's partner rows (keys: ts[k]; dK/dV pass: queries' ts[q + 1])
  long long* thr;        // [NBUCK]
  float* tw;             // [NBUCK + 3]
  float* pw;             // [2 Lw - 1 (+ pad)]
  float* dtw; float* dpw;
};
template <int HD>
__device__ __forceinline__ HstuLdsV2 hstu_carve(unsigned char* smem, int Lw) {
  using L = Lay<HD>;
  HstuLdsV2 l;
  l.img0 = smem; l.img1 = smem + (size_t)(HCH + 1) * L::ROW3;
  unsigned char* p = smem + 2 * (size_t)(HCH + 1) * L::ROW3;           // ROW3 is a multiple of 64: 8-byte aligned
  l.ts_p = reinterpret_cast<long long*>(p); p += HCH * 8;
  l.thr = reinterpret_cast<long long*>(p); p += NBUCK * 8;
  l.tw = reinterpret_cast<float*>(p); p += (NBUCK + 3) * 4;
  l.pw = reinterpret_cast<float*>(p); p += (size_t)(2 * Lw) * 4;
  l.dtw = reinterpret_cast<float*>(p); p += (NBUCK + 3) * 4;
  l.dpw = reinterpret_cast<float*>(p);
  return l;
}
template <int HD> inline size_t hstu_lds_bytes(int Lw, bool grads) {
  return 2 * (size_t)(HCH + 1) * Lay<HD>::ROW3 + HCH * 8 + NBUCK * 8 + (NBUCK + 3) * 4 + (size_t)(2 * Lw) * 4 +
         (grads ? (NBUCK + 3) * 4 + (size_t)(2 * Lw) * 4 : 0);
}
__device__ __forceinline__ void hstu_load_tables(const HstuV2Args& a, const HstuLdsV2& l, int tid, int nthreads, bool grads) {
  const int nw = a.time_thr != nullptr ? (int)a.time_thr[NBUCK] : 1;     // entries of time_w; later buckets read the last one (hstu_fill)
  for (int j = tid; j < NBUCK; j += nthreads) {
    l.thr[j] = a.time_thr != nullptr ? a.time_thr[j] : 0;
    l.tw[j] = a.time_w != nullptr ? a.time_w[j < nw ? j : nw - 1] : 0.f;
    if (grads) l.dtw[j] = 0.f;
  }
  for (int j = tid; j < 2 * a.Lw - 1; j += nthreads) {
    l.pw[j] = a.pos_w != nullptr ? a.pos_w[j] : 0.f;
    if (grads) l.dpw[j] = 0.f;
  }
}

// ---- forward: a lane owns a query; chunks of keys ------------------------------------------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void v2_hstu_fwd_kernel(HstuV2Args a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const HstuLdsV2 l = hstu_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  hstu_load_tables(a, l, tid, NW * 64, false);
  const float inv_l = 1.0f / (float)a.Lw;
  const int n_tiles = (n + 15) >> 4;

  for (int c0 = 0; c0 < n; c0 += HCH) {
    const int len = min(HCH, n - c0);
    __syncthreads();                                  // the previous chunk's readers are done (first chunk: nothing to wait for)
    stage_images2<HD>(a.k + (row0 + c0) * a.ldk + h * HD, a.ldk, 1.f, l.img0, a.v + (row0 + c0) * a.ldv + h * HD, a.ldv, 1.f, l.img1, len, tid, NW * 64);
    if (tbias) for (int j = tid; j < len; j += NW * 64) l.ts_p[j] = tsb[c0 + j];
    __syncthreads();

    for_my_tiles<NW, true>(wave, n_tiles, [&](int qt) {
      if (qt * 16 + 15 < c0) return;                  // every query of the tile lies before the chunk's keys
      const int qrow = qt * 16 + i;
      const bool qok = qrow < n;
      const int qsafe = qok ? qrow : n - 1;
      const long long grow = row0 + qsafe;
      P3 Qp[L::NS];
      load_owner_planes<HD>(a.q + grow * a.ldq + h * HD, g, 1.f, Qp);
      const long long t_q1 = tbias ? tsb[qsafe + 1] : 0;
      f32x4 oT[L::NCB];
      float* op = a.o + grow * a.ldo + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb)
        oT[cb] = (c0 > 0 && qok) ? *reinterpret_cast<const f32x4*>(op + 16 * cb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      const int t_last = min((qt * 16 + 15 - c0) >> 5, (len - 1) >> 5);        // last 32-key tile of the chunk this query tile sees
      for (int t = 0; t <= t_last; ++t) {
        f32x4 sT[2];
        rows_times_owner<HD>(l.img0, t * 32, len, Qp, i, g, sT);
        float pr[8];
        int bk8[8];
        if (tbias) hstu_buckets8(l.thr, l.ts_p, t_q1, true, t * 32, g, len, qok && t * 32 + 19 + 4 * g < len && c0 + t * 32 + 19 + 4 * g <= qrow, bk8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kl = t * 32 + 16 * (e >> 2) + 4 * g + (e & 3), key = c0 + kl;
          const bool valid = qok && kl < len && key <= qrow;
          float bias = 0.f;
          if (tbias) bias += l.tw[bk8[e]];
          if (pbias) bias += l.pw[max(a.Lw - 1 + key - qrow, 0)];
          pr[e] = valid ? hstu_silu(sT[e >> 2][e & 3] + bias) * inv_l : 0.f;
        }
        const P3 Pp = split8(pr);
        cols_times_slots<HD>(l.img1, t * 32, len, Pp, i, g, oT);
      }
      if (qok) {
#pragma unroll
        for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(op + 16 * cb + 4 * g) = oT[cb];
      }
    });
  }
}

// ---- backward, pass 1: dQ and the bias gradients.  A lane owns a query; chunks of keys ------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void v2_hstu_bwd_dq_kernel(HstuV2Args a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const HstuLdsV2 l = hstu_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  const bool tgrad = tbias && a.d_time_w != nullptr, pgrad = pbias && a.d_pos_w != nullptr;
  hstu_load_tables(a, l, tid, NW * 64, true);
  const float inv_l = 1.0f / (float)a.Lw;
  const int n_tiles = (n + 15) >> 4;

  for (int c0 = 0; c0 < n; c0 += HCH) {
    const int len = min(HCH, n - c0);
    __syncthreads();
    stage_images2<HD>(a.k + (row0 + c0) * a.ldk + h * HD, a.ldk, 1.f, l.img0, a.v + (row0 + c0) * a.ldv + h * HD, a.ldv, 1.f, l.img1, len, tid, NW * 64);
    if (tbias) for (int j = tid; j < len; j += NW * 64) l.ts_p[j] = tsb[c0 + j];
    __syncthreads();

    for_my_tiles<NW, true>(wave, n_tiles, [&](int qt) {
      if (qt * 16 + 15 < c0) return;
      const int qrow = qt * 16 + i;
      const bool qok = qrow < n;
      const int qsafe = qok ? qrow : n - 1;
      const long long grow = row0 + qsafe;
      P3 Qp[L::NS], Dp[L::NS];
      load_owner_planes<HD>(a.q + grow * a.ldq + h * HD, g, 1.f, Qp);
      load_owner_planes<HD>(a.dout + grow * a.lddo + h * HD, g, qok ? 1.f : 0.f, Dp);
      const long long t_q1 = tbias ? tsb[qsafe + 1] : 0;
      f32x4 dqT[L::NCB];
      float* dqp = a.dq + grow * a.lddq + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb)
        dqT[cb] = (c0 > 0 && qok) ? *reinterpret_cast<const f32x4*>(dqp + 16 * cb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      BucketRun run;
      run.init();
      const int t_last = min((qt * 16 + 15 - c0) >> 5, (len - 1) >> 5);
      for (int t = 0; t <= t_last; ++t) {
        f32x4 sT[2], dpT[2];
        rows_times_owner<HD>(l.img0, t * 32, len, Qp, i, g, sT);
        rows_times_owner<HD>(l.img1, t * 32, len, Dp, i, g, dpT);
        float ds[8];
        int bk8[8];
        if (tbias) hstu_buckets8(l.thr, l.ts_p, t_q1, true, t * 32, g, len, qok && t * 32 + 19 + 4 * g < len && c0 + t * 32 + 19 + 4 * g <= qrow, bk8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kl = t * 32 + 16 * (e >> 2) + 4 * g + (e & 3), key = c0 + kl;
          const bool valid = qok && kl < len && key <= qrow;
          float bias = 0.f;
          int bk = 0;
          const int pidx = max(a.Lw - 1 + key - qrow, 0);
          if (tbias) { bk = bk8[e]; bias += l.tw[bk]; }
          if (pbias) bias += l.pw[pidx];
          const float z = sT[e >> 2][e & 3] + bias;
          ds[e] = valid ? dpT[e >> 2][e & 3] * inv_l * hstu_silu_d(z) : 0.f;
          if (valid && tgrad) run.add(l.dtw, bk, ds[e]);
        }
        if (pgrad) {
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {      // (invalid elements are zeros: a diagonal is valid or invalid as a whole up to the window's edges)
            float dmain, dwrap;
            diagonal_sums(ds + 4 * hb, i, dmain, dwrap);
            const int base = a.Lw - 1 + c0 + t * 32 + 16 * hb + 4 * g - qt * 16;
            if (dmain != 0.f) atomicAdd(l.dpw + min(max(base + 3 - i, 0), 2 * a.Lw - 2), dmain);
            if (dwrap != 0.f) atomicAdd(l.dpw + min(max(base - 13 - i, 0), 2 * a.Lw - 2), dwrap);
          }
        }
        const P3 Sp = split8(ds);
        cols_times_slots<HD>(l.img0, t * 32, len, Sp, i, g, dqT);       // dQ^T[c][q] += sum_j K[j][c] dS^T[j][q]
      }
      if (tgrad) run.flush(l.dtw);
      if (qok) {
#pragma unroll
        for (int cb = 0; cb < L::NCB; ++cb) *reinterpret_cast<f32x4*>(dqp + 16 * cb + 4 * g) = dqT[cb];
      }
    });
  }
  __syncthreads();
  if (tgrad) {
    const int nw = (int)a.time_thr[NBUCK];
    for (int j = tid; j < NBUCK; j += NW * 64) { const float v = l.dtw[j]; if (v != 0.f) atomicAdd(a.d_time_w + (j < nw ? j : nw - 1), v); }
  }
  if (pgrad) for (int j = tid; j < 2 * a.Lw - 1; j += NW * 64) { const float v = l.dpw[j]; if (v != 0.f) atomicAdd(a.d_pos_w + j, v); }
}

// ---- backward, pass 2: dK, dV.  A lane owns a key; chunks of QUERIES (images of Q and dO) ---------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void v2_hstu_bwd_dkv_kernel(HstuV2Args a) {
  using L = Lay<HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long long row0 = a.cu[b];
  const int n = (int)(a.cu[b + 1] - row0);
  if (n <= 0) return;
  const HstuLdsV2 l = hstu_carve<HD>(smem, a.Lw);
  const long long* tsb = a.ts != nullptr ? a.ts + row0 + b : nullptr;
  const bool tbias = tsb != nullptr && a.time_w != nullptr, pbias = a.pos_w != nullptr;
  hstu_load_tables(a, l, tid, NW * 64, false);
  const float inv_l = 1.0f / (float)a.Lw;
  const int n_tiles = (n + 15) >> 4;

  for (int c0 = 0; c0 < n; c0 += HCH) {                 // queries [c0, c0 + len)
    const int len = min(HCH, n - c0);
    __syncthreads();
    stage_images2<HD>(a.q + (row0 + c0) * a.ldq + h * HD, a.ldq, 1.f, l.img0, a.dout + (row0 + c0) * a.lddo + h * HD, a.lddo, 1.f, l.img1, len, tid, NW * 64);
    if (tbias) for (int j = tid; j < len; j += NW * 64) l.ts_p[j] = tsb[c0 + j + 1];     // a query's time is its NEXT stamp (hstu.py:96-104)
    __syncthreads();

    for_my_tiles<NW, false>(wave, n_tiles, [&](int kt) {
      if (kt * 16 > c0 + len - 1) return;               // every key of the tile lies behind the chunk's queries
      const int krow = kt * 16 + i;
      const bool kok = krow < n;
      const int ksafe = kok ? krow : n - 1;
      const long long grow = row0 + ksafe;
      P3 Kp[L::NS], Vp[L::NS];
      load_owner_planes<HD>(a.k + grow * a.ldk + h * HD, g, 1.f, Kp);
      load_owner_planes<HD>(a.v + grow * a.ldv + h * HD, g, 1.f, Vp);
      const long long t_k = tbias ? tsb[ksafe] : 0;
      // the first chunk that holds a query >= this tile's first key starts the accumulation; later chunks continue it
      const bool first = c0 <= kt * 16;
      f32x4 dkT[L::NCB], dvT[L::NCB];
      float* dkp = a.dk + grow * a.lddk + h * HD;
      float* dvp = a.dv + grow * a.lddv + h * HD;
#pragma unroll
      for (int cb = 0; cb < L::NCB; ++cb) {
        dkT[cb] = (!first && kok) ? *reinterpret_cast<const f32x4*>(dkp + 16 * cb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        dvT[cb] = (!first && kok) ? *reinterpret_cast<const f32x4*>(dvp + 16 * cb + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int t_first = max(kt * 16 - c0, 0) >> 5;    // first 32-query tile of the chunk that holds a query >= the tile's first key
      const int t_end = (len - 1) >> 5;
      for (int t = t_first; t <= t_end; ++t) {
        f32x4 sm[2], dpm[2];                            // S[q][key], dP[q][key]: register (qb, r) = query c0 + 32 t + 16 qb + 4 g + r
        rows_times_owner<HD>(l.img0, t * 32, len, Kp, i, g, sm);
        rows_times_owner<HD>(l.img1, t * 32, len, Vp, i, g, dpm);
        float pd[8], ds[8];
        int bk8[8];
        if (tbias) hstu_buckets8(l.thr, l.ts_p, t_k, false, t * 32, g, len, kok && t * 32 + 19 + 4 * g < len && krow <= c0 + t * 32 + 4 * g, bk8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ql = t * 32 + 16 * (e >> 2) + 4 * g + (e & 3), q = c0 + ql;
          const bool valid = kok && ql < len && krow <= q;
          float bias = 0.f;
          if (tbias) bias += l.tw[bk8[e]];
          if (pbias) bias += l.pw[max(a.Lw - 1 + krow - q, 0)];
          const float z = sm[e >> 2][e & 3] + bias;
          float pz, dz;
          hstu_silu_both(z, pz, dz);
          pd[e] = valid ? pz * inv_l : 0.f;
          ds[e] = valid ? dpm[e >> 2][e & 3] * inv_l * dz : 0.f;
        }
        const P3 Pp = split8(pd);
        cols_times_slots<HD>(l.img1, t * 32, len, Pp, i, g, dvT);       // dV^T[c][key] += sum_q dO[q][c] P[q][key]
        const P3 Sp = split8(ds);
        cols_times_slots<HD>(l.img0, t * 32, len, Sp, i, g, dkT);       // dK^T[c][key] += sum_q Q[q][c] dS[q][key]
      }
      if (kok) {
#pragma unroll
        for (int cb = 0; cb < L::NCB; ++cb) {
          *reinterpret_cast<f32x4*>(dkp + 16 * cb + 4 * g) = dkT[cb];
          *reinterpret_cast<f32x4*>(dvp + 16 * cb + 4 * g) = dvT[cb];
        }
      }
    });
  }
}

template <int HD>
int launch_hstu_fwd(const HstuV2Args& a, hipStream_t stream) {
  constexpr int NW = 8;
  const size_t lds = hstu_lds_bytes<HD>(a.Lw, false);
  if (lds > 160 * 1024) return RT_ERR_UNSUPPORTED;
  auto kern = &v2_hstu_fwd_kernel<HD, NW>;
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<a.B * a.H, NW * 64, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
template <int HD>
int launch_hstu_bwd(const HstuV2Args& a, hipStream_t stream) {
  constexpr int NW = 8;
  const size_t lds_q = hstu_lds_bytes<HD>(a.Lw, true), lds_kv = hstu_lds_bytes<HD>(a.Lw, false);
  if (lds_q > 160 * 1024) return RT_ERR_UNSUPPORTED;
  auto kq = &v2_hstu_bwd_dq_kernel<HD, NW>;
  auto kkv = &v2_hstu_bwd_dkv_kernel<HD, NW>;
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
  kq<<<a.B * a.H, NW * 64, lds_q, stream>>>(a);
  RT_CHECK_LAUNCH();
  kkv<<<a.B * a.H, NW * 64, lds_kv, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace

int rt_v2_hstu_fwd(const rt_varlen::HstuV2Args& a, hipStream_t stream) {
  if (a.hd == 64) return launch_hstu_fwd<64>(a, stream);
  if (a.hd == 32) return launch_hstu_fwd<32>(a, stream);
  return RT_ERR_UNSUPPORTED;
}
int rt_v2_hstu_bwd(const rt_varlen::HstuV2Args& a, hipStream_t stream) {
  if (a.hd == 64) return launch_hstu_bwd<64>(a, stream);
  if (a.hd == 32) return launch_hstu_bwd<32>(a, stream);
  return RT_ERR_UNSUPPORTED;
}

int rt_v2_varlen_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream) {
  if (a.hd == 64) return train ? launch_fwd<64, 8, true>(a, max_len, stream) : launch_fwd<64, 8, false>(a, max_len, stream);
  if (a.hd == 32) return train ? launch_fwd<32, 8, true>(a, max_len, stream) : launch_fwd<32, 8, false>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}

int rt_v2_varlen_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream) {
  if (a.hd == 64) return launch_bwd<64, 8>(a, max_len, stream);
  if (a.hd == 32) return launch_bwd<32, 8>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}

// bidirectional (no causal mask, no pad keys): BERT4Rec's key-padding-masked window on packed rows
int rt_v2_bidir_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream) {
  if (a.hd == 64) return train ? launch_fwd<64, 8, true, false>(a, max_len, stream) : launch_fwd<64, 8, false, false>(a, max_len, stream);
  if (a.hd == 32) return train ? launch_fwd<32, 8, true, false>(a, max_len, stream) : launch_fwd<32, 8, false, false>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}
int rt_v2_bidir_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream) {
  if (a.hd == 64) return launch_bwd<64, 8, false>(a, max_len, stream);
  if (a.hd == 32) return launch_bwd<32, 8, false>(a, max_len, stream);
  return RT_ERR_UNSUPPORTED;
}
