// K8 / K9 / K10 / K11 — similarity logits and losses of the training step (fp32; gBCE transform in fp64).
//
//  sampled losses   `DistanceSimilarityModule._get_pos_neg_logits` (similarity.py:88-95) materialises the gather
//                   item_embs[candidates] as a [B, L, 1+N, d] tensor (3.4 GB at B128 L200 N128 d256) before a batched
//                   mat-vec; `_calc_bce_loss` / `_calc_gbce_loss` / `_calc_sampled_softmax_loss`
//                   (lightning.py:164-212) then reduce it.  Here one wave per position keeps the session row in
//                   registers, streams the 1+N candidate rows (16 lanes per row, 4 rows per step), and writes only
//                   the [M, 1+N] logits and one loss value per position; in training the same pass also yields
//                   d(session) (online softmax) and dL/dlogits; the backward groups the (position, candidate) pairs by
//                   item id (counting sort) and reduces every row of the dense table gradient exactly once.
//  full softmax     logits come from the MFMA GEMM (rt_gemm) over the ACTIVE positions only (y != 0; the others
//                   have zero loss and zero gradient, ignore_index=0 at lightning.py:157); `softmax_ce_rows`
//                   converts each logits row in place into (softmax - onehot) * weight / norm and the two gradient
//                   GEMMs consume it.
//  cosine           L2 row normalisation with the reference's max(||x||, 1e-8) denominator (similarity.py:97-100),
//                   forward and backward; the sampled kernels normalise only the rows they read (K11).
//  loss reduction   sum(loss) / sum(loss > 0) for the softmax family (lightning.py:159-161), sum(loss) / sum(y != 0)
//                   for BCE / gBCE (lightning.py:197-198).
#include "rt_common.h"
#include <cstdlib>
#include "rt_scan.h"

namespace {

enum { LOSS_BCE = 0, LOSS_GBCE = 1, LOSS_SAMPLED_SOFTMAX = 2 };
constexpr float EPS_COS = 1e-8f;

struct SampledArgs {
  const float* sess; long long ld_sess;   // [M, d]
  const float* table;                     // [V, d]
  const long long* y;                     // [M] positives (0 = inactive position)
  const long long* neg;                   // [M, N]
  const float* w;                         // [M] weights
  int M, N, d;
  int loss, cosine;
  float inv_t;                            // 1 / logits_t
  double gbce_beta;
  float* logits;                          // [M, 1+N] saved logits (already / logits_t)
  float* loss_pos;                        // [M] weighted loss per position
  // backward
  const float* norm;                      // [1] normaliser produced by the reduction kernel
  float gscale;                           // upstream dL/dloss (host side factor)
  const float* upstream;                  // [1] upstream dL/dloss on the device (nullable): multiplies gscale — no host read, no divide launch
  bool narrow_rows;                       // the table half was asked for on its own (see launch_sampled)
  float* d_sess; long long ld_dsess;      // [M, d] overwritten (training forward: UNIT gradient, upstream = norm = 1)
  float* d_table;                         // [V, d] overwritten (every row written exactly once)
  int V;
  float* glog;                            // [M, 1+N] dL/d(raw similarity) of every (position, candidate)
  float* inv_ns;                          // [M] 1 / max(||session||, eps)   (cosine)
  int* count; int* offsets; int* cursor;  // [V+1] counting-sort state over candidate ids
  int2* pairs;                            // [M*(1+N)] (position m, unit gradient bits) records grouped by candidate id
  float* bterm;                           // [M*(1+N)] cosine only: the pair's share of sum_j g_j cos_j (chain rule of e/|e|)
  int* blocksum;                          // scan scratch
  int* rank;                              // [M, 1+N] rank of every pair inside its candidate id
  int prepared;                           // the counting sort of the candidate ids ran ahead (rt_sampled_loss_prepare): ranks and segment
                                          // offsets exist before the training forward, which then writes the pair records itself
  int* heavy_count; int* heavy_ids; int* heavy_chunk;   // chunk list of the rows with > HEAVY_T pairs (rt_scan.h)
  float* slab; float* slab_bsum;          // [chunks][d] partial rows of those chunks, [chunks] partial cosine sums
};

__device__ __forceinline__ float group16_sum(float v) {  // sum over the 16 lanes of a quarter-wave
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double softplus_d(double z) { return z > 0 ? z + log1p(exp(-z)) : log1p(exp(z)); }
__device__ __forceinline__ double sigmoid_d(double z) { return 1.0 / (1.0 + exp(-z)); }

// gBCE transform of the positive logit and its derivative (lightning.py:164-186), fp64.
__device__ __forceinline__ void gbce_transform(double z, double beta, double& f, double& df) {
  const double eps = 1e-10, fmax_d = 1.7976931348623157e308;
  const double sg = sigmoid_d(z);
  double p = sg, dp = sg * (1.0 - sg);
  if (p < eps) { p = eps; dp = 0; } else if (p > 1 - eps) { p = 1 - eps; dp = 0; }
  double a = pow(p, -beta), da = -beta * pow(p, -beta - 1.0) * dp;
  if (a < 1 + eps) { a = 1 + eps; da = 0; } else if (a > fmax_d) { a = fmax_d; da = 0; }
  double bq = 1.0 / (a - 1.0), db = -da / ((a - 1.0) * (a - 1.0));
  if (bq < eps) { bq = eps; db = 0; } else if (bq > fmax_d) { bq = fmax_d; db = 0; }
  f = log(bq); df = db / bq;
}

// One wave per position; D4 = number of float4 per lane for a d-wide row split over 16 lanes (d <= 64*D4); the 4
// quarter-waves stream 4 candidate rows per step.
//
// TRAIN = true is the training forward: the SAME pass over the candidate rows also produces the position-side half of
// the backward — the reference's gather (3.4 GB at the C2 shape) is the whole cost of these kernels, and a separate
// backward kernel for d_sess would read every row a second time.  For the sampled softmax the session gradient
// sum_j softmax_j e_j is accumulated with an online softmax (running max / sum / rescaled accumulator per
// quarter-wave, merged at the end — the flash-attention recurrence applied to the loss); BCE / gBCE gradients depend
// on their own logit only and are accumulated directly.  Gradients are "unit" (upstream = 1, normaliser = 1): the
// backward scales by gscale / norm once both exist.  Also takes the counting-sort ranks of the negatives.
// SM = sampled softmax (fp32 only); !SM = BCE / gBCE (fp64 transforms) — separate instances keep the fp64 temporaries out
// of the softmax kernel's register budget.
// One table row slice (D4 float4 per lane) through inline asm: hipcc's own waitcnt bookkeeping turns every attempt at software
// pipelining of these gathers into `s_waitcnt vmcnt(0)` (one row per quarter-wave in flight); asm loads are invisible to it
// and `gather_wait<N>` — which names the destination registers as in/out operands, so no use can be scheduled above it —
// waits until at most N younger loads are outstanding (loads return in order).
template <int D4>
__device__ __forceinline__ void gather_issue(const float* row, int sub, f32x4 (&e)[D4]) {
#pragma unroll
  for (int i = 0; i < D4; ++i)
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(e[i]) : "v"(row + (sub + 16 * i) * 4) : "memory");
}
template <int N, int D4>
__device__ __forceinline__ void gather_wait(f32x4 (&e)[D4]) {
  if constexpr (D4 == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(e[0]) : "n"(N) : "memory");
  else if constexpr (D4 == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(e[0]), "+v"(e[1]) : "n"(N) : "memory");
  else if constexpr (D4 == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : "n"(N) : "memory");
  else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "n"(N) : "memory");
}

// FAST (d == 64 D4 and 1 + N <= 260, decided on the host): the pipelined gather loop below; otherwise the generic loop.
template <int D4, bool TRAIN, bool SM, bool FAST = false>
__global__ __launch_bounds__(256) void sampled_fwd_kernel(SampledArgs a) {
  __shared__ float s_z[4][260];
  __shared__ int s_cid[4][260];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  const int C = a.N + 1;
  const long long yy = a.y[m];
  float* zrow = a.logits + (long long)m * C;
  const int sub = lane & 15, grp = lane >> 4;
  if (yy == 0) {  // inactive: zero loss and gradient, no pairs; logits are never read for it
    if (lane == 0) { a.loss_pos[m] = 0.f; if (TRAIN) a.inv_ns[m] = 1.f; }
    if (TRAIN && grp == 0) {
#pragma unroll
      for (int i = 0; i < D4; ++i) {
        const int c = (sub + 16 * i) * 4;
        if (c < a.d) *reinterpret_cast<f32x4*>(a.d_sess + (long long)m * a.ld_dsess + c) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    return;
  }
  // candidate ids of this position -> LDS (coalesced), so the row gathers below do not wait on an id load each
  for (int j = lane; j < C && j < 260; j += 64) s_cid[wave][j] = (j == 0) ? (int)yy : (int)a.neg[(long long)m * a.N + (j - 1)];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // session slice of this lane: float4 index sub + 16*i
  f32x4 sv[D4];
  float ss = 0.f;
  const float* srow = a.sess + (long long)m * a.ld_sess;
#pragma unroll
  for (int i = 0; i < D4; ++i) {
    const int c = (sub + 16 * i) * 4;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    sv[i] = (c < a.d) ? *reinterpret_cast<const f32x4*>(srow + c) : z;
    ss += sv[i][0] * sv[i][0] + sv[i][1] * sv[i][1] + sv[i][2] * sv[i][2] + sv[i][3] * sv[i][3];
  }
  ss = group16_sum(ss);
  const float ns = sqrtf(ss);
  const float inv_ns = a.cosine ? 1.0f / fmaxf(ns, EPS_COS) : 1.0f;

  // training state of this quarter-wave: accumulator of (weight_j * e_hat_j), online-softmax max / sum
  f32x4 acc[D4];
  float m_run = -INFINITY, l_run = 0.f;
  if (TRAIN) {
#pragma unroll
    for (int i = 0; i < D4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if constexpr (FAST) {
    // ---- pipelined gather.  One iteration = 4 candidate rows (one per quarter-wave).  The generic loop below keeps ONE row
    // per quarter-wave in flight and pays a full memory round trip per iteration (it used to pay a second one for a returning
    // rank atomic): 33 dependent round trips per position made the C2 forward latency-bound (281 us for 3.4 GB of gather,
    // 1.6 GB of it from the fabric).  Here the rows of PF iterations are in flight; slots past the last candidate (and the
    // PF - 1 look-ahead iterations past the end) read the PAD row with weight 0, so the in-flight count is constant and the
    // wait is a compile-time vmcnt; logits go to LDS and leave in one coalesced store after the loop.
    constexpr int PF = D4 <= 4 ? 3 : 2;
    const int n_it = (C + 3) >> 2;
    auto row_of_it = [&](int it) -> const float* {
      const int j = it * 4 + grp;
      const long long cid = (j < C) ? (long long)s_cid[wave][j] : 0;
      return a.table + cid * (long long)a.d;
    };
    auto consume = [&](int it, const f32x4 (&ev)[D4]) {
      const int j = it * 4 + grp;
      const bool valid = j < C;
      float dot = 0.f, ee = 0.f;
#pragma unroll
      for (int i = 0; i < D4; ++i) {
        dot += ev[i][0] * sv[i][0] + ev[i][1] * sv[i][1] + ev[i][2] * sv[i][2] + ev[i][3] * sv[i][3];
        ee += ev[i][0] * ev[i][0] + ev[i][1] * ev[i][1] + ev[i][2] * ev[i][2] + ev[i][3] * ev[i][3];
      }
      dot = group16_sum(dot);
      ee = group16_sum(ee);
      const float einv = a.cosine ? 1.0f / fmaxf(sqrtf(ee), EPS_COS) : 1.0f;
      float z = dot;
      if (a.cosine) z = z * inv_ns * einv;
      z *= a.inv_t;
      if (valid && sub == 0) s_z[wave][j] = z;
      if (TRAIN) {
        float wj;   // weight of e_hat_j in d s_hat, up to the factors applied after the loop
        if (SM) {
          const float m_new = valid ? fmaxf(m_run, z) : m_run;
          const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
          wj = valid ? __expf(z - m_new) : 0.f;
          l_run = l_run * alpha + wj;
          m_run = m_new;
#pragma unroll
          for (int i = 0; i < D4; ++i) acc[i] *= alpha;
        } else {
          double zd = (double)z, gd;
          if (j == 0) {
            double df = 1.0;
            if (a.loss == LOSS_GBCE) { double f; gbce_transform(zd, a.gbce_beta, f, df); zd = f; }
            gd = (sigmoid_d(zd) - 1.0) * df;
          } else {
            gd = sigmoid_d(zd);
          }
          wj = valid ? (float)(gd / (double)C) : 0.f;
        }
        const float we = wj * einv;
#pragma unroll
        for (int i = 0; i < D4; ++i) acc[i] += ev[i] * we;
      }
    };
    // everything the compiler tracks (session row, ids) has landed before the first asm load: vmcnt starts at 0
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    f32x4 evb[PF][D4];
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) gather_issue<D4>(row_of_it(u), sub, evb[u]);
#pragma unroll 1
    for (int it0 = 0; it0 < n_it; it0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int it = it0 + u;
        gather_issue<D4>(row_of_it(it + PF - 1), sub, evb[(u + PF - 1) % PF]);   // look-ahead (PAD row past the end)
        gather_wait<(PF - 1) * D4, D4>(evb[u]);
        if (it < n_it) consume(it, evb[u]);
      }
    }
    // Drain the look-ahead loads.  The wait must NAME the buffers: once the last `consume` is done the compiler considers their
    // registers dead and would hand them to the address arithmetic below while PF - 1 loads are still on their way into them
    // (seen on hardware as HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: a table row landed in a pointer).
#pragma unroll
    for (int u = 0; u < PF; ++u) gather_wait<0, D4>(evb[u]);
  } else {
#pragma unroll 2
  for (int j0 = 0; j0 < C; j0 += 4) {
    const int j = j0 + grp;
    const bool valid = j < C;
    float dot = 0.f, ee = 0.f;
    f32x4 ev[D4];
    long long cid = 0;
    if (valid) cid = (j < 260) ? (long long)s_cid[wave][j] : ((j == 0) ? yy : a.neg[(long long)m * a.N + (j - 1)]);
    {
      const float* er = a.table + cid * (long long)a.d;   // invalid slots read the PAD row: a valid address, weight 0
#pragma unroll
      for (int i = 0; i < D4; ++i) {
        const int c = (sub + 16 * i) * 4;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ev[i] = (valid && c < a.d) ? *reinterpret_cast<const f32x4*>(er + c) : z;
        dot += ev[i][0] * sv[i][0] + ev[i][1] * sv[i][1] + ev[i][2] * sv[i][2] + ev[i][3] * sv[i][3];
        ee += ev[i][0] * ev[i][0] + ev[i][1] * ev[i][1] + ev[i][2] * ev[i][2] + ev[i][3] * ev[i][3];
      }
    }
    dot = group16_sum(dot);
    ee = group16_sum(ee);
    const float einv = a.cosine ? 1.0f / fmaxf(sqrtf(ee), EPS_COS) : 1.0f;
    float z = dot;
    if (a.cosine) z = z * inv_ns * einv;
    z *= a.inv_t;
    if (valid && sub == 0) {
      if (j < 260) s_z[wave][j] = z;
      zrow[j] = z;
    }
    if (TRAIN) {
      float wj;   // weight of e_hat_j in d s_hat, up to the factors applied after the loop
      if (SM) {
        const float m_new = valid ? fmaxf(m_run, z) : m_run;
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
        wj = valid ? __expf(z - m_new) : 0.f;
        l_run = l_run * alpha + wj;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < D4; ++i) acc[i] *= alpha;
      } else {
        double zd = (double)z, gd;
        if (j == 0) {
          double df = 1.0;
          if (a.loss == LOSS_GBCE) { double f; gbce_transform(zd, a.gbce_beta, f, df); zd = f; }
          gd = (sigmoid_d(zd) - 1.0) * df;
        } else {
          gd = sigmoid_d(zd);
        }
        wj = valid ? (float)(gd / (double)C) : 0.f;
      }
      const float we = wj * einv;
#pragma unroll
      for (int i = 0; i < D4; ++i) acc[i] += ev[i] * we;
    }
  }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  if constexpr (FAST) { for (int j = lane; j < C; j += 64) zrow[j] = s_z[wave][j]; }   // logits row: one coalesced store
  if (TRAIN && !a.prepared) {   // counting-sort ranks of the negatives (targets are ranked by agg_rank_kernel): independent atomics, all in flight
    for (int j = 1 + lane; j < C; j += 64) {
      const long long cid = (j < 260) ? (long long)s_cid[wave][j] : a.neg[(long long)m * a.N + (j - 1)];
      if (cid != 0) a.rank[m * C + j] = atomicAdd(a.count + cid, 1);
    }
  }
  const float wgt = a.w[m];
  auto zat = [&](int j) -> float { return j < 260 ? s_z[wave][j] : zrow[j]; };
  float out, mx = 0.f, se = 1.f;
  if (SM) {
    mx = -INFINITY;
    for (int j = lane; j < C; j += 64) mx = fmaxf(mx, zat(j));
    mx = wave_max(mx);
    se = 0.f;
    for (int j = lane; j < C; j += 64) se += __expf(zat(j) - mx);
    se = wave_sum(se);
    out = (mx + __logf(se) - zat(0)) * wgt;
  } else {
    double accd = 0.0;
    for (int j = lane; j < C; j += 64) {
      double z = (double)zat(j);
      if (j == 0) {
        if (a.loss == LOSS_GBCE) { double f, df; gbce_transform(z, a.gbce_beta, f, df); z = f; }
        accd += softplus_d(-z);
      } else {
        accd += softplus_d(z);
      }
    }
    accd = wave_sum_d(accd);
    out = (float)(accd / (double)C) * wgt;
  }
  if (lane == 0) a.loss_pos[m] = out;
  if (!TRAIN) return;

  // ---- unit dL/d(raw similarity) of every candidate (consumed by the table-row reduction of the backward) ----
  const float gs = wgt * a.inv_t;   // includes d z / d(raw similarity) = 1 / logits_t
  float* grow = a.glog + (long long)m * C;
  for (int j = lane; j < C; j += 64) {
    float g;
    if (SM) {
      // positions whose weighted loss is not > 0 are outside the normaliser but still carry gradient in the
      // reference (sum(loss) / sum(loss > 0)); keep that.
      g = (__expf(zat(j) - mx) / se - (j == 0 ? 1.f : 0.f)) * gs;
    } else {
      double z = (double)zat(j), gd;
      if (j == 0) {
        double df = 1.0;
        if (a.loss == LOSS_GBCE) { double f; gbce_transform(z, a.gbce_beta, f, df); z = f; }
        gd = (sigmoid_d(z) - 1.0) * df;
      } else {
        gd = sigmoid_d(z);
      }
      g = (float)(gd / (double)C) * gs;
    }
    grow[j] = g;
    if (a.prepared) {   // the record pairs_scatter_kernel would write: the segment offsets and this pair's rank are known already
      const long long cid = (j == 0) ? yy : ((j < 260) ? (long long)s_cid[wave][j] : a.neg[(long long)m * a.N + (j - 1)]);
      if (cid != 0) {
        const int slot = a.offsets[cid] + a.rank[m * C + j];
        a.pairs[slot] = make_int2(m, __float_as_int(g * inv_ns));       // (inv_ns == 1 unless cosine)
        if (a.cosine) a.bterm[slot] = g * (zat(j) / a.inv_t);            // logits = cos / t
      }
    }
  }
  if (lane == 0) a.inv_ns[m] = inv_ns;

  // ---- session gradient: merge the 4 quarter-wave streams (lanes with equal `sub` hold the same columns) ----
  float scale_g = 1.f;
  if (SM) {
    float mall = fmaxf(m_run, __shfl_xor(m_run, 16, 64));
    mall = fmaxf(mall, __shfl_xor(mall, 32, 64));
    scale_g = (m_run == -INFINITY) ? 0.f : __expf(m_run - mall);
    float l = l_run * scale_g;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    scale_g /= l;   // softmax_j = exp(z_j - mall) / l
  }
  f32x4 ds[D4];
#pragma unroll
  for (int i = 0; i < D4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[i][e] * scale_g;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      ds[i][e] = v;
    }
  if (SM) {   // minus the positive's own row: d s_hat = sum_j softmax_j e_hat_j - e_hat_0
    const float* er = a.table + yy * (long long)a.d;
    f32x4 e0[D4];
    float ee = 0.f;
#pragma unroll
    for (int i = 0; i < D4; ++i) {
      const int c = (sub + 16 * i) * 4;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      e0[i] = (c < a.d) ? *reinterpret_cast<const f32x4*>(er + c) : z;
      ee += e0[i][0] * e0[i][0] + e0[i][1] * e0[i][1] + e0[i][2] * e0[i][2] + e0[i][3] * e0[i][3];
    }
    ee = group16_sum(ee);
    const float einv0 = a.cosine ? 1.0f / fmaxf(sqrtf(ee), EPS_COS) : 1.0f;
#pragma unroll
    for (int i = 0; i < D4; ++i) ds[i] -= e0[i] * einv0;
  }
#pragma unroll
  for (int i = 0; i < D4; ++i) ds[i] *= gs;
  if (a.cosine) {  // d s = (d s_hat - s_hat (s_hat . d s_hat)) / ns
    float proj = 0.f;
#pragma unroll
    for (int i = 0; i < D4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) proj += ds[i][e] * sv[i][e] * inv_ns;
    proj = group16_sum(proj);
#pragma unroll
    for (int i = 0; i < D4; ++i) {
      f32x4 sh = sv[i] * inv_ns;
      ds[i] = (ns > EPS_COS) ? (ds[i] - sh * proj) * inv_ns : ds[i] * inv_ns;
    }
  }
  if (grp == 0) {
    float* drow = a.d_sess + (long long)m * a.ld_dsess;
#pragma unroll
    for (int i = 0; i < D4; ++i) {
      const int c = (sub + 16 * i) * 4;
      if (c < a.d) *reinterpret_cast<f32x4*>(drow + c) = ds[i];
    }
  }
}

// (An XCD-sliced forward — the table cut into eight id ranges, one per XCD's L2, a position's softmax merged from eight partial states —
// was built and measured in round 4: correct, and slower (340 vs 264 us at the C2 step: bound by its own fixed cost, eight waves per
// position; profiles/r4_loss_sliced_probe.txt).  Removed in round 5; the figures stay in DESIGN.md K8/K9.)

// d_sess = unit gradient * gscale / norm
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ src, long long ld_src, float* __restrict__ dst,
                                                         long long ld_dst, int M, int d, const float* __restrict__ norm,
                                                         float gscale, const float* __restrict__ upstream) {
  const float sc = gscale * (upstream != nullptr ? upstream[0] : 1.f) / norm[0];
  const int per_row = d >> 2;
  const long long n4 = (long long)M * per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / per_row; const int c = (int)(i - r * per_row) * 4;
    *reinterpret_cast<f32x4*>(dst + r * ld_dst + c) = *reinterpret_cast<const f32x4*>(src + r * ld_src + c) * sc;
  }
}

// ---- counting sort, ahead of the forward pass (rt_sampled_loss_prepare): the ranks of the negatives, as the training forward takes them ----
__global__ __launch_bounds__(256) void sampled_rank_kernel(SampledArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.M || a.y[m] == 0) return;          // inactive positions have no pairs
  const int C = a.N + 1;
  for (int j = 1 + lane; j < C; j += 64) {
    const long long cid = a.neg[(long long)m * a.N + (j - 1)];
    if (cid != 0) a.rank[m * C + j] = atomicAdd(a.count + cid, 1);
  }
}

// ---- counting sort over candidate ids: exclusive scan of count[0..V] (rt_scan.h), then scatter of the pairs ----
__global__ __launch_bounds__(256) void pairs_scatter_kernel(SampledArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.M) return;
  const long long yy = a.y[m];
  if (yy == 0) return;
  const int C = a.N + 1;
  // slot = start of the id's segment + the rank taken when it was counted.  The record carries everything the row reduction
  // needs — the position and the (cosine: 1/|s| scaled) unit gradient — so that reduction is a TWO-deep chain
  // (record -> session row) of wave-uniform scalar loads and row gathers instead of three dependent vector loads.
  const float inv_ns_m = a.cosine ? a.inv_ns[m] : 1.f;
  for (int j = lane; j < C; j += 64) {
    const long long cid = (j == 0) ? yy : a.neg[(long long)m * a.N + (j - 1)];
    if (cid != 0) {
      const int slot = a.offsets[cid] + a.rank[m * C + j];
      const float g = a.glog[m * C + j];
      a.pairs[slot] = make_int2(m, __float_as_int(g * inv_ns_m));
      if (a.cosine) a.bterm[slot] = g * (a.logits[m * C + j] / a.inv_t);   // logits = cos / t
    }
  }
}

// Backward, part 2: d_table[id] = sum over the pairs of that id — written once, no float atomics.
// A wave accumulates the pairs [beg, end) with stride-free contiguous access, 4 gather chains in flight.
constexpr int HEAVY_T = 128;     // ids with more pairs than this go to the workgroup-per-id kernel (popularity skew:
                                 // a Zipf catalog gives its top item ~9% of all positives -> one wave would serialise them)
constexpr int HEAVY_CH = 128;    // pairs per chunk of a popular row: one 4-wave workgroup, two rounds of 16 gathers per wave
constexpr int HEAVY_WAVES = 4;

template <int D4, int U, bool COS>
__device__ __forceinline__ void accumulate_pairs_u(const SampledArgs& a, int& k, int end, int lane,
                                                   f32x4 (&acc)[(D4 + 3) / 4], float& bsum) {
  constexpr int NA = (D4 + 3) / 4;
  for (; k + U <= end; k += U) {   // U independent (record -> session row) chains in flight; k is wave-uniform: scalar loads
    float g[U]; const float* sr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int2 rec = a.pairs[k + u];
      g[u] = __int_as_float(rec.y);
      if (COS) bsum += a.bterm[k + u];
      sr[u] = a.sess + (long long)rec.x * a.ld_sess;
    }
    f32x4 v[U][NA];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int c = lane * 4 + 256 * i;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[u][i] = (c < a.d) ? *reinterpret_cast<const f32x4*>(sr[u] + c) : z;
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] += v[u][i] * g[u];
  }
}
// a wave's serial chain is ~1-2 us per round of gathers: 16 pairs per round, then 4, then 1
template <int D4>
__device__ __forceinline__ void accumulate_pairs(const SampledArgs& a, int beg, int end, int lane,
                                                 f32x4 (&acc)[(D4 + 3) / 4], float& bsum) {
  int k = beg;
  if (a.cosine) {
    accumulate_pairs_u<D4, (D4 <= 4 ? 16 : 8), true>(a, k, end, lane, acc, bsum);
    accumulate_pairs_u<D4, 4, true>(a, k, end, lane, acc, bsum);
    accumulate_pairs_u<D4, 1, true>(a, k, end, lane, acc, bsum);
  } else {
    accumulate_pairs_u<D4, (D4 <= 4 ? 16 : 8), false>(a, k, end, lane, acc, bsum);
    accumulate_pairs_u<D4, 4, false>(a, k, end, lane, acc, bsum);
    accumulate_pairs_u<D4, 1, false>(a, k, end, lane, acc, bsum);
  }
}

// cosine: e -> e/|e| chain rule, then the single store of the row
template <int D4>
__device__ __forceinline__ void finish_table_row(const SampledArgs& a, int id, int lane, bool any,
                                                 f32x4 (&acc)[(D4 + 3) / 4], float bsum) {
  constexpr int NA = (D4 + 3) / 4;
  float* dr = a.d_table + (long long)id * a.d;
  const float sc = a.gscale * (a.upstream != nullptr ? a.upstream[0] : 1.f) / a.norm[0];   // glog holds unit gradients (training forward)
  if (a.cosine && any) {
    const float* er = a.table + (long long)id * a.d;
    float ee = 0.f;
    f32x4 ev[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = lane * 4 + 256 * i;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      ev[i] = (c < a.d) ? *reinterpret_cast<const f32x4*>(er + c) : z;
      ee += ev[i][0] * ev[i][0] + ev[i][1] * ev[i][1] + ev[i][2] * ev[i][2] + ev[i][3] * ev[i][3];
    }
    const float ne = sqrtf(wave_sum(ee));
    const float inv_ne = 1.0f / fmaxf(ne, EPS_COS);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      acc[i] = (ne > EPS_COS) ? (acc[i] - ev[i] * (inv_ne * bsum)) * inv_ne : acc[i] * inv_ne;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < a.d) *reinterpret_cast<f32x4*>(dr + c) = acc[i] * sc;
  }
}

// one wave per table row; a popular row (cursor >= 0) only combines the partial rows its chunks left in the slab
template <int D4>
__global__ __launch_bounds__(256) void sampled_bwd_rows_kernel(SampledArgs a) {
  constexpr int NA = (D4 + 3) / 4;
  const int lane = threadIdx.x & 63;
  for (int blk = blockIdx.x; blk * 4 < a.V; blk += gridDim.x) {      // (a grid smaller than V / 4: persistent workgroups, see launch_sampled)
  const int id = __builtin_amdgcn_readfirstlane(blk * 4 + (threadIdx.x >> 6));   // wave-uniform: offsets / records are scalar loads
  if (id >= a.V) return;
  const int beg = a.offsets[id], end = a.offsets[id + 1];
  f32x4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int slot = a.cursor[id];
  if (slot >= 0) {
    const int n_ch = (end - beg + HEAVY_CH - 1) / HEAVY_CH;
    for (int c = 0; c < n_ch; ++c) {   // fixed order: deterministic given the pair order
      bsum += a.slab_bsum[slot + c];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int col = lane * 4 + 256 * i;
        if (col < a.d) acc[i] += *reinterpret_cast<const f32x4*>(a.slab + (long long)(slot + c) * a.d + col);
      }
    }
  } else {
    accumulate_pairs<D4>(a, beg, end, lane, acc, bsum);
  }
  finish_table_row<D4>(a, id, lane, end > beg, acc, bsum);
  }
}

// one 4-wave workgroup per chunk of a popular row: 32 pairs per wave, combined through LDS, partial row -> slab
template <int D4>
__global__ __launch_bounds__(HEAVY_WAVES * 64) void sampled_bwd_heavy_kernel(SampledArgs a) {
  constexpr int NA = (D4 + 3) / 4;
  __shared__ f32x4 s_part[HEAVY_WAVES][NA][64];
  __shared__ float s_bsum[HEAVY_WAVES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_chunks = *a.heavy_count;
  for (int h = blockIdx.x; h < n_chunks; h += gridDim.x) {
    const int id = a.heavy_ids[h];
    const int beg = a.offsets[id] + a.heavy_chunk[h] * HEAVY_CH;
    const int end = min(beg + HEAVY_CH, a.offsets[id + 1]);
    const int per = HEAVY_CH / HEAVY_WAVES;
    const int wb = min(beg + wave * per, end), we = min(wb + per, end);
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    accumulate_pairs<D4>(a, wb, we, lane, acc, bsum);
#pragma unroll
    for (int i = 0; i < NA; ++i) s_part[wave][i][lane] = acc[i];
    if (lane == 0) s_bsum[wave] = bsum;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int col = lane * 4 + 256 * i;
        if (col < a.d)
          *reinterpret_cast<f32x4*>(a.slab + (long long)h * a.d + col) =
              (s_part[0][i][lane] + s_part[1][i][lane]) + (s_part[2][i][lane] + s_part[3][i][lane]);
      }
      if (lane == 0) a.slab_bsum[h] = (s_bsum[0] + s_bsum[1]) + (s_bsum[2] + s_bsum[3]);
    }
    __syncthreads();
  }
}

// ---- loss reduction: out[0] = loss, out[1] = normaliser -------------------------------------------
// mode 0: sum(l) / count(l > 0)  (softmax family)     mode 1: sum(l) / count(y != 0)  (BCE family)
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const float* __restrict__ loss_pos, const long long* __restrict__ y,
                                                           int M, int mode, float* __restrict__ out) {
  __shared__ double s_sum[16]; __shared__ double s_cnt[16];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < M; i += 1024) {
    const float l = loss_pos[i];
    s += (double)l;
    if (mode == 0) c += (l > 0.f) ? 1.0 : 0.0; else c += (y[i] != 0) ? 1.0 : 0.0;
  }
  s = wave_sum_d(s); c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = s; s_cnt[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0, tc = 0;
    for (int i = 0; i < 16; ++i) { ts += s_sum[i]; tc += s_cnt[i]; }
    out[0] = (float)(ts / tc);
    out[1] = (float)tc;
  }
}

// ---- full-softmax rows: logits row -> loss and (softmax - onehot) * w * gscale / norm in place ------------
// pass A (grad == 0): loss_pos[r] = (lse - z_y) * w ; pass B (grad == 1): row := (softmax - onehot) * coef
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(float* __restrict__ logits, long long ld, int R, int V,
                                                              const long long* __restrict__ y_act, const float* __restrict__ w_act,
                                                              float inv_t, int grad, const float* __restrict__ norm, float gscale,
                                                              const float* __restrict__ upstream,
                                                              float* __restrict__ loss_pos, float* __restrict__ lse_out) {
  const int r = blockIdx.x;
  if (r >= R) return;
  float* row = logits + (long long)r * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float red[4];
  __shared__ float s_b;
  const long long yy = y_act[r];
  if (!grad) {
    float mx = -INFINITY;
    for (int j = tid; j < V; j += 256) mx = fmaxf(mx, row[j] * inv_t);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) s_b = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = s_b;
    float se = 0.f;
    for (int j = tid; j < V; j += 256) se += __expf(row[j] * inv_t - mx);
    se = wave_sum(se);
    __syncthreads();
    if (lane == 0) red[wave] = se;
    __syncthreads();
    if (tid == 0) {
      const float lse = mx + __logf(red[0] + red[1] + red[2] + red[3]);
      lse_out[r] = lse;
      loss_pos[r] = (lse - row[yy] * inv_t) * w_act[r];
    }
  } else {
    const float lse = lse_out[r];
    const float coef = w_act[r] * (gscale * (upstream != nullptr ? upstream[0] : 1.f)) / norm[0] * inv_t;
    for (int j = tid; j < V; j += 256) {
      float p = __expf(row[j] * inv_t - lse);
      row[j] = (p - (j == yy ? 1.f : 0.f)) * coef;
    }
  }
}

// ---- L2 row normalisation (cosine), forward / backward ----------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, long long ldx, int M, int d,
                                                         float* __restrict__ y, long long ldy, float* __restrict__ nrm) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (long long)m * ldx;
  float s = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  const float n = sqrtf(wave_sum(s));
  const float inv = 1.0f / fmaxf(n, EPS_COS);
  for (int c = lane * 4; c < d; c += 256)
    *reinterpret_cast<f32x4*>(y + (long long)m * ldy + c) = *reinterpret_cast<const f32x4*>(xr + c) * inv;
  if (lane == 0 && nrm) nrm[m] = n;
}
// dx = (dy - xhat (xhat . dy)) / n   (n > eps);   dy / eps otherwise.  `accumulate`: dx += (table gradients)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                         long long ldx, int M, int d, int accumulate,
                                                         float* __restrict__ dx, long long lddx) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* xr = x + (long long)m * ldx;
  const float* gr = dy + (long long)m * lddy;
  float s = 0.f, t = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), g = *reinterpret_cast<const f32x4*>(gr + c);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    t += v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
  }
  s = wave_sum(s); t = wave_sum(t);
  const float n = sqrtf(s);
  const float inv = 1.0f / fmaxf(n, EPS_COS);
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), g = *reinterpret_cast<const f32x4*>(gr + c);
    f32x4 o = (n > EPS_COS) ? (g - v * (t * inv * inv)) * inv : g * inv;
    float* dst = dx + (long long)m * lddx + c;
    if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = o;
  }
}

// ---- row gather / scatter-add (active positions of the full-softmax loss) ---------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, long long lds_, const long long* __restrict__ idx,
                                                          int R, int d, float* __restrict__ dst, long long ldd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* s = src + idx[r] * lds_;
  for (int c = lane * 4; c < d; c += 256)
    *reinterpret_cast<f32x4*>(dst + (long long)r * ldd + c) = *reinterpret_cast<const f32x4*>(s + c);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, long long lds_, const long long* __restrict__ idx,
                                                           int R, int d, float* __restrict__ dst, long long ldd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float* o = dst + idx[r] * ldd;   // indices are unique: plain stores
  for (int c = lane * 4; c < d; c += 256)
    *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(src + (long long)r * lds_ + c);
}

// upper bound on the chunks of popular rows among `n_pairs` pairs: every chunk but the last of a row is full, and a row
// needs more than HEAVY_T pairs to have chunks at all
inline long long heavy_chunk_cap(long long n_pairs) { return n_pairs / HEAVY_CH + n_pairs / HEAVY_T + 2; }

constexpr bool fwd_fast_allowed() { return true; }      // (the generic gather loop stays for shapes the fast one does not tile)

// stage: 0 = inference forward, 1 = training forward (logits, loss, unit gradients, ranks), 2 = backward
template <int D4>
int launch_sampled(const SampledArgs& a, int stage, hipStream_t stream) {
  const int blocks = (a.M + 3) / 4;
  const bool fast = a.d == D4 * 64 && a.N + 1 <= 260 && fwd_fast_allowed();
  if (stage == 0) {
    if (a.loss == LOSS_SAMPLED_SOFTMAX) {
      if (fast) sampled_fwd_kernel<D4, false, true, true><<<blocks, 256, 0, stream>>>(a);
      else sampled_fwd_kernel<D4, false, true><<<blocks, 256, 0, stream>>>(a);
    } else {
      if (fast) sampled_fwd_kernel<D4, false, false, true><<<blocks, 256, 0, stream>>>(a);
      else sampled_fwd_kernel<D4, false, false><<<blocks, 256, 0, stream>>>(a);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
  }
  const int n = a.V + 1;
  if (stage == 3) {   // the counting sort of the candidate ids, ahead of the forward pass: counts + ranks, segment offsets, popular-id chunks
    RT_CHECK_HIP(hipMemsetAsync(a.count, 0, sizeof(int) * (((size_t)n + 1 + 15) & ~(size_t)15), stream));   // + heavy_count (+ the pad)
    sampled_rank_kernel<<<blocks, 256, 0, stream>>>(a);
    RT_CHECK_LAUNCH();
    agg_rank_kernel<<<(a.M + AGG_T - 1) / AGG_T, AGG_T, 0, stream>>>(a.y, a.M, a.count, a.rank, a.N + 1);
    RT_CHECK_LAUNCH();
    return exclusive_scan_counts(a.count, n, a.offsets, a.cursor, a.blocksum, stream, HEAVY_T, HEAVY_CH, a.heavy_count, a.heavy_ids,
                                 a.heavy_chunk);
  }
  if (stage == 1) {
    if (!a.prepared) RT_CHECK_HIP(hipMemsetAsync(a.count, 0, sizeof(int) * (((size_t)n + 1 + 15) & ~(size_t)15), stream));   // + heavy_count (+ the pad)
    if (a.loss == LOSS_SAMPLED_SOFTMAX) {
      if (fast) sampled_fwd_kernel<D4, true, true, true><<<blocks, 256, 0, stream>>>(a);
      else sampled_fwd_kernel<D4, true, true><<<blocks, 256, 0, stream>>>(a);
    } else {
      if (fast) sampled_fwd_kernel<D4, true, false, true><<<blocks, 256, 0, stream>>>(a);
      else sampled_fwd_kernel<D4, true, false><<<blocks, 256, 0, stream>>>(a);
    }
    RT_CHECK_LAUNCH();
    return RT_OK;
  }
  if (!a.prepared) {   // (prepared: the sort ran ahead of the forward pass, which wrote the pair records itself)
    agg_rank_kernel<<<(a.M + AGG_T - 1) / AGG_T, AGG_T, 0, stream>>>(a.y, a.M, a.count, a.rank, a.N + 1);
    RT_CHECK_LAUNCH();
    const int rc = exclusive_scan_counts(a.count, n, a.offsets, a.cursor, a.blocksum, stream, HEAVY_T, HEAVY_CH, a.heavy_count,
                                         a.heavy_ids, a.heavy_chunk);
    if (rc != RT_OK) return rc;
    pairs_scatter_kernel<<<blocks, 256, 0, stream>>>(a);
    RT_CHECK_LAUNCH();
  }
  const long long max_chunks = heavy_chunk_cap((long long)a.M * (a.N + 1));
  sampled_bwd_heavy_kernel<D4><<<(int)min(max_chunks, (long long)rt_num_cus() * 8), HEAVY_WAVES * 64, 0, stream>>>(a);
  RT_CHECK_LAUNCH();
  {
    // A long catalog's row reducer asked for on its own (d_sess == NULL: the optimiser's half, issued on a side stream) runs beside the
    // backward pass for most of its length: as one workgroup per 4 rows it fills every SIMD's wave slots and the attention backward's
    // workgroups queue for them (HSTU C4: v3_hstu_bwd_dq 1,010 us beside it against 190 alone).  TWO persistent workgroups per CU leave
    // the slots free: HSTU 23.1 -> 23.5 k, eSASRec 13.97 -> 14.16 k seqs/s on one box (gpurun_out/r6_rows3); short catalogs (C2: 6,687
    // workgroups, 194 us) keep the wide grid — narrowed it stretches past the backward pass (86.2 -> 83.2 k).
    int grid = (a.V + 3) / 4;
    if (a.narrow_rows && grid >= 64 * rt_num_cus()) grid = 2 * rt_num_cus();
    sampled_bwd_rows_kernel<D4><<<grid, 256, 0, stream>>>(a);
  }
  RT_CHECK_LAUNCH();
  return RT_OK;
}
int dispatch_sampled(const SampledArgs& a, int stage, hipStream_t stream) {
  if (a.d <= 64) return launch_sampled<1>(a, stage, stream);
  if (a.d <= 128) return launch_sampled<2>(a, stage, stream);
  if (a.d <= 256) return launch_sampled<4>(a, stage, stream);
  if (a.d <= 512) return launch_sampled<8>(a, stage, stream);
  return RT_ERR_UNSUPPORTED;
}

// workspace shared by the training forward and the backward
void carve_workspace(SampledArgs& a, void* workspace, int M, int N, int V, int d) {
  const size_t C = (size_t)N + 1, n = (size_t)V + 1;
  const size_t cap = (size_t)heavy_chunk_cap((long long)M * (long long)C);
  float* f = reinterpret_cast<float*>(workspace);
  a.slab = f; f += cap * (size_t)d;   // first: 16-byte aligned rows
  a.slab_bsum = f; f += cap;
  a.glog = f; f += (size_t)M * C;
  a.inv_ns = f; f += M;
  a.bterm = f; f += (size_t)M * C;
  if (reinterpret_cast<uintptr_t>(f) & 7) f += 1;                   // the records are 8-byte loads
  int* ip = reinterpret_cast<int*>(f);
  a.pairs = reinterpret_cast<int2*>(ip); ip += 2 * (size_t)M * C;
  a.rank = ip; ip += (size_t)M * C;
  ip = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(ip) + 255) & ~(uintptr_t)255);   // the cleared region starts on a 256-byte line and
  a.count = ip; ip += n;                                                                    // is a multiple of 64 bytes long: ONE fill kernel
  a.heavy_count = ip; ip += 1;   // directly behind count: one memset clears both          // (an unaligned memset is three: head, body, tail)
  ip += (16 - ((n + 1) & 15)) & 15;
  a.heavy_ids = ip; ip += cap;
  a.heavy_chunk = ip; ip += cap;
  a.offsets = ip; ip += n;
  a.cursor = ip; ip += n;
  a.blocksum = ip;
}

}  // namespace

extern "C" {

// sess [M,d], table [V,d], y [M], neg [M,N], w [M]  ->  logits [M,1+N], loss_pos [M]
int rt_sampled_loss_fwd(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                        const float* w, int32_t M, int32_t N, int32_t d, int32_t loss, int32_t cosine, float logits_t,
                        double gbce_beta, float* logits, float* loss_pos, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || N < 0 || loss < LOSS_BCE || loss > LOSS_SAMPLED_SOFTMAX || (ld_sess & 3)) return RT_ERR_INVALID_ARG;
  SampledArgs a{};
  a.sess = sess; a.ld_sess = ld_sess; a.table = table; a.y = reinterpret_cast<const long long*>(y);
  a.neg = reinterpret_cast<const long long*>(neg); a.w = w; a.M = M; a.N = N; a.d = d; a.loss = loss; a.cosine = cosine;
  a.inv_t = 1.0f / logits_t; a.gbce_beta = gbce_beta; a.logits = logits; a.loss_pos = loss_pos;
  return dispatch_sampled(a, 0, stream);
}

// Host arithmetic: bytes of the scratch that rt_sampled_loss_fwd_train fills and rt_sampled_loss_bwd consumes.
size_t rt_sampled_loss_bwd_workspace_bytes(int32_t M, int32_t N, int32_t V, int32_t d) {
  const size_t C = (size_t)N + 1, n = (size_t)V + 1;
  const size_t nb = (n + SCAN_T * SCAN_E - 1) / (SCAN_T * SCAN_E);
  const size_t cap = (size_t)heavy_chunk_cap((long long)M * (long long)C);
  return 4 * ((size_t)M * C * 5 + (size_t)M + 3 * n + nb + 64 + 4 + cap * ((size_t)d + 3)) + 256 + 64;   // (+ the aligned cleared region)
}

// The counting sort of the (position, candidate) pairs by candidate id depends on the ids alone: called ahead of the forward pass (on
// another stream: a handful of small launches that would otherwise sit between the forward and the backward kernels), it leaves the
// ranks, the segment offsets and the popular ids' chunks in `workspace`; rt_sampled_loss_fwd_train / _bwd called with prepared = 1 on
// the SAME workspace then skip their share of it (the forward writes the pair records, the backward starts at the row reductions).
int rt_sampled_loss_prepare(const int64_t* y, const int64_t* neg, int32_t M, int32_t N, int32_t d, int32_t V, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || N < 0 || y == nullptr || (N > 0 && neg == nullptr)) return RT_ERR_INVALID_ARG;
  if ((long long)M * (N + 1) >= (1LL << 31)) return RT_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < rt_sampled_loss_bwd_workspace_bytes(M, N, V, d)) return RT_ERR_WORKSPACE;
  SampledArgs a{};
  a.y = reinterpret_cast<const long long*>(y); a.neg = reinterpret_cast<const long long*>(neg); a.M = M; a.N = N; a.d = d; a.V = V;
  carve_workspace(a, workspace, M, N, V, d);
  return dispatch_sampled(a, 3, stream);
}

// Training forward: everything rt_sampled_loss_fwd writes, plus — in `workspace` — the unit gradient of every logit,
// the counting-sort ranks of the candidate ids and 1/|session| (cosine), and d_sess_unit [M,d] (upstream = norm = 1).
// One pass over the candidate rows for any 1+N (online softmax for the sampled softmax gradient).
int rt_sampled_loss_fwd_train(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                              const float* w, int32_t M, int32_t N, int32_t d, int32_t V, int32_t loss, int32_t cosine,
                              float logits_t, double gbce_beta, float* logits, float* loss_pos, float* d_sess_unit,
                              int64_t ld_du, void* workspace, size_t workspace_bytes, int32_t prepared, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || N < 0 || loss < LOSS_BCE || loss > LOSS_SAMPLED_SOFTMAX || (ld_sess & 3) || (ld_du & 3)) return RT_ERR_INVALID_ARG;
  if ((long long)M * (N + 1) >= (1LL << 31)) return RT_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < rt_sampled_loss_bwd_workspace_bytes(M, N, V, d)) return RT_ERR_WORKSPACE;
  SampledArgs a{};
  a.sess = sess; a.ld_sess = ld_sess; a.table = table; a.y = reinterpret_cast<const long long*>(y);
  a.neg = reinterpret_cast<const long long*>(neg); a.w = w; a.M = M; a.N = N; a.d = d; a.V = V; a.loss = loss; a.cosine = cosine;
  a.inv_t = 1.0f / logits_t; a.gbce_beta = gbce_beta; a.logits = logits; a.loss_pos = loss_pos;
  a.norm = nullptr; a.gscale = 1.f; a.d_sess = d_sess_unit; a.ld_dsess = ld_du; a.prepared = prepared != 0;
  carve_workspace(a, workspace, M, N, V, d);
  return dispatch_sampled(a, 1, stream);
}

// Backward of the training forward (call once per forward: it consumes the ranks in `workspace`).  d_sess [M,d] =
// d_sess_unit * gscale / norm; d_table [V,d] is fully overwritten — the (position, candidate) pairs are counting-sorted
// by candidate id and each table row is reduced by one wave (a 16-wave workgroup for popular ids): no float atomics.
int rt_sampled_loss_bwd(const float* sess, int64_t ld_sess, const float* table, const int64_t* y, const int64_t* neg,
                        int32_t M, int32_t N, int32_t d, int32_t V, int32_t cosine, float logits_t, const float* logits,
                        const float* norm, float gscale, const float* upstream, const float* d_sess_unit, int64_t ld_du, float* d_sess,
                        int64_t ld_dsess, float* d_table, void* workspace, size_t workspace_bytes, int32_t prepared, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || N < 0 || (ld_sess & 3) || (ld_dsess & 3) || (ld_du & 3)) return RT_ERR_INVALID_ARG;
  if ((long long)M * (N + 1) >= (1LL << 31)) return RT_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < rt_sampled_loss_bwd_workspace_bytes(M, N, V, d)) return RT_ERR_WORKSPACE;
  SampledArgs a{};
  a.sess = sess; a.ld_sess = ld_sess; a.table = table; a.y = reinterpret_cast<const long long*>(y);
  a.neg = reinterpret_cast<const long long*>(neg); a.M = M; a.N = N; a.d = d; a.V = V; a.cosine = cosine;
  a.inv_t = 1.0f / logits_t; a.logits = const_cast<float*>(logits); a.norm = norm; a.gscale = gscale; a.upstream = upstream;
  a.d_table = d_table; a.prepared = prepared != 0; a.narrow_rows = d_sess == nullptr;
  carve_workspace(a, workspace, M, N, V, d);
  if (d_sess != nullptr) {   // position side: a scaled copy of what the training forward accumulated
    scale_rows_kernel<<<rt_num_cus() * 4, 256, 0, stream>>>(d_sess_unit, ld_du, d_sess, ld_dsess, M, d, norm, gscale, upstream);
    RT_CHECK_LAUNCH();
  }
  if (d_table == nullptr) return RT_OK;   // table side asked for separately (e.g. on another stream: it is needed only by Adam)
  return dispatch_sampled(a, 2, stream);
}

// out[0] = sum(loss_pos) / normaliser, out[1] = normaliser;  mode 0: count(loss_pos > 0), mode 1: count(y != 0)
int rt_loss_reduce(const float* loss_pos, const int64_t* y, int32_t M, int32_t mode, float* out, hipStream_t stream) {
  (void)hipGetLastError();
  loss_reduce_kernel<<<1, 1024, 0, stream>>>(loss_pos, reinterpret_cast<const long long*>(y), M, mode, out);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// logits [R,V] (raw similarities of the R active rows) -> loss_pos [R], lse [R]  (grad = 0)
//                                                      -> logits := (softmax - onehot) * w * gscale / (norm * t)  (grad = 1)
int rt_softmax_ce_rows(float* logits, int64_t ld, int32_t R, int32_t V, const int64_t* y_act, const float* w_act,
                       float logits_t, int32_t grad, const float* norm, float gscale, const float* upstream, float* loss_pos, float* lse,
                       hipStream_t stream) {
  (void)hipGetLastError();
  if (R <= 0) return RT_OK;
  softmax_ce_rows_kernel<<<R, 256, 0, stream>>>(logits, ld, R, V, reinterpret_cast<const long long*>(y_act), w_act,
                                                1.0f / logits_t, grad, norm, gscale, upstream, loss_pos, lse);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_l2norm_fwd(const float* x, int64_t ldx, int32_t M, int32_t d, float* y, int64_t ldy, float* nrm, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || (ldx & 3) || (ldy & 3)) return RT_ERR_INVALID_ARG;
  l2norm_fwd_kernel<<<(M + 3) / 4, 256, 0, stream>>>(x, ldx, M, d, y, ldy, nrm);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_l2norm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int32_t M, int32_t d, int32_t accumulate,
                  float* dx, int64_t lddx, hipStream_t stream) {
  (void)hipGetLastError();
  if (M <= 0) return RT_OK;
  if ((d & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3)) return RT_ERR_INVALID_ARG;
  l2norm_bwd_kernel<<<(M + 3) / 4, 256, 0, stream>>>(dy, lddy, x, ldx, M, d, accumulate, dx, lddx);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_gather_rows(const float* src, int64_t ld_src, const int64_t* idx, int32_t R, int32_t d, float* dst, int64_t ld_dst,
                   hipStream_t stream) {
  (void)hipGetLastError();
  if (R <= 0) return RT_OK;
  if ((d & 3) || (ld_src & 3) || (ld_dst & 3)) return RT_ERR_INVALID_ARG;
  gather_rows_kernel<<<(R + 3) / 4, 256, 0, stream>>>(src, ld_src, reinterpret_cast<const long long*>(idx), R, d, dst, ld_dst);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// dst[idx[r]] = src[r]  (idx unique; other rows of dst untouched — the caller zero-fills them)
int rt_scatter_rows(const float* src, int64_t ld_src, const int64_t* idx, int32_t R, int32_t d, float* dst, int64_t ld_dst,
                    hipStream_t stream) {
  (void)hipGetLastError();
  if (R <= 0) return RT_OK;
  if ((d & 3) || (ld_src & 3) || (ld_dst & 3)) return RT_ERR_INVALID_ARG;
  scatter_rows_kernel<<<(R + 3) / 4, 256, 0, stream>>>(src, ld_src, reinterpret_cast<const long long*>(idx), R, d, dst, ld_dst);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
