// K12 — exact full-catalog top-k scorer (recommend() hot loop).
//
// Replaces the per-128-user loop of the reference `TorchRanker.rank`
// (rectools/models/rank/rank_torch.py:122-155): `U @ I.T` -> masked_fill(-inf) from a dense
// `csr.toarray()` -> `torch.topk` -> D2H.  Here the catalog is streamed from HBM exactly once per
// user batch, scores live only in MFMA accumulators, and per-lane top-k lists absorb them.
//
// Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain), so scores match an
// fp32 reference to rounding of the summation order only.
//
// Structure (gfx950):
//   * workgroup = 4 waves; item block = 128 catalog rows (32 per wave); user batch UB = 32*TU rows.
//   * k-chunks of 32 floats are staged HBM -> VGPR -> LDS with full 128-byte lines per row and a
//     +4-float row pad (conflict-free ds_read_b128 fragment reads); double-buffered, one barrier
//     per chunk, next chunk's global loads in flight under the current chunk's MFMAs.
//   * MFMA operand roles: A = items (rows i), B = users (cols j); lane l feeds A[item l&31][k] and
//     B[k][user l&31] for k = 8s + 4(l>>5) + t, t = 0..3 — a fixed permutation of the reduction index
//     that lets every lane fetch its 4 MFMA steps with ONE ds_read_b128 per operand.
//   * D layout (col = lane&31 = user, row = (r&3)+8(r>>2)+4(lane>>5) = item) puts 16 items of ONE user
//     in each lane: selection is lane-local, no cross-lane traffic.
//   * selection: `score >= thr` fast path in registers; rare slow path does the viewed-items check
//     (binary search in the user's CSR row) and a replace-worst insert into the lane's list (global
//     workspace, L2 resident).  A per-user global threshold (atomicMax of each full list's worst
//     score) is shared by all workgroups, and can be seeded from a catalog prefix (two-phase launch).
//   * merge kernel: one workgroup per user compacts all lists and extracts the k best in order
//     (score desc, position asc — the tie rule of oracle/ranker_oracle.py).
#include "rt_common.h"

namespace {

constexpr int IB = 128;       // catalog rows per item block
constexpr int KC = 32;        // floats per k-chunk
constexpr int LDK = KC + 4;   // padded LDS row stride (floats)
constexpr int NTHREADS = 256;
constexpr int LISTS_PER_WG = 8;  // 4 waves x 2 half-waves

enum { DIST_DOT = 0, DIST_COSINE = 1, DIST_EUCLID = 2 };

struct TopkArgs {
  const float* users; long long user_stride; const long long* user_rows; int n_users;
  const float* items; long long item_stride; const long long* whitelist;
  long long item_begin, item_end;  // candidate positions handled by this launch
  int d; int distance; int k;
  const long long* filt_indptr; const int* filt_indices;
  float* list_scores; int* list_pos; int* list_counts;  // [n_lists][UB][k], [n_lists][UB]
  int list_base;  // first list id of this launch
  int ub;         // user slots per list (== 32*TU)
  unsigned* gthr; // [ub] ordered keys
};

__device__ __forceinline__ bool better(float s, long long p, float s2, long long p2) {
  return (s > s2) || (s == s2 && p < p2);
}

// Is candidate id `cid` among the (ascending) filter indices of this user?
__device__ __forceinline__ bool is_filtered(const TopkArgs& a, int u, long long cid) {
  if (a.filt_indptr == nullptr) return false;
  long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    long long v = (long long)a.filt_indices[mid];
    if (v < cid) lo = mid + 1; else hi = mid;
  }
  return lo < a.filt_indptr[u + 1] && (long long)a.filt_indices[lo] == cid;
}

template <int TU>
__global__ __launch_bounds__(NTHREADS) void topk_partial_kernel(TopkArgs a) {
  constexpr int UB = 32 * TU;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][IB][LDK]
  float* Us = smem + 2 * IB * LDK;       // [2][UB][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;   // user column inside a 32-user tile / item row for the A operand
  const int half = lane >> 5;

  const long long n_cand = a.item_end - a.item_begin;
  const long long n_blocks = (n_cand + IB - 1) / IB;
  const int n_chunks = (a.d + KC - 1) / KC;

  // ---- per-lane list state (one list per (workgroup, wave, half, user)) ----
  const int list_id = a.list_base + blockIdx.x * LISTS_PER_WG + wave * 2 + half;
  float worst_s[TU]; long long worst_p[TU]; int worst_slot[TU]; int cnt[TU]; float thr[TU];
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    worst_s[tu] = -INFINITY; worst_p[tu] = -1; worst_slot[tu] = 0; cnt[tu] = 0; thr[tu] = -INFINITY;
  }

  // ---- user-row staging assignment: TU float4 per thread per chunk ----
  const float* urow[TU]; int ur_r[TU];
  const int c4 = tid & 7;  // float4 column inside the chunk
#pragma unroll
  for (int j = 0; j < TU; ++j) {
    int r = (tid >> 3) + 32 * j;
    ur_r[j] = r;
    if (r < a.n_users) {
      long long src = a.user_rows ? a.user_rows[r] : (long long)r;
      urow[j] = a.users + src * a.user_stride;
    } else {
      urow[j] = nullptr;
    }
  }

  for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const long long pos0 = a.item_begin + blk * IB;
    // item-row staging assignment: 4 float4 per thread per chunk
    const float* irow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long p = pos0 + (tid >> 3) + 32 * j;
      if (p < a.item_end) {
        long long src = a.whitelist ? a.whitelist[p] : p;
        irow[j] = a.items + src * a.item_stride;
      } else {
        irow[j] = nullptr;
      }
    }

    f32x16 acc[TU];
#pragma unroll
    for (int tu = 0; tu < TU; ++tu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tu][r] = 0.f;
    float nrm_i = 0.f; float nrm_u[TU];
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) nrm_u[tu] = 0.f;

    f32x4 ri[4]; f32x4 ru[TU];
    auto gload = [&](int c) {
      const int kofs = c * KC + c4 * 4;
      const bool kin = kofs < a.d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ri[j] = (irow[j] != nullptr && kin) ? *reinterpret_cast<const f32x4*>(irow[j] + kofs) : z;
      }
#pragma unroll
      for (int j = 0; j < TU; ++j) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ru[j] = (urow[j] != nullptr && kin) ? *reinterpret_cast<const f32x4*>(urow[j] + kofs) : z;
      }
    };
    auto lstore = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r = (tid >> 3) + 32 * j;
        *reinterpret_cast<f32x4*>(As + (buf * IB + r) * LDK + c4 * 4) = ri[j];
      }
#pragma unroll
      for (int j = 0; j < TU; ++j) {
        *reinterpret_cast<f32x4*>(Us + (buf * UB + ur_r[j]) * LDK + c4 * 4) = ru[j];
      }
    };

    __syncthreads();  // previous block's last chunk fully consumed before buffer 0 is overwritten
    gload(0);
    lstore(0);
    __syncthreads();

    for (int c = 0; c < n_chunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < n_chunks) gload(c + 1);
      const float* Ab = As + (buf * IB + wave * 32 + col) * LDK + 4 * half;
      const float* Ub = Us + (buf * UB + col) * LDK + 4 * half;
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        f32x4 av = *reinterpret_cast<const f32x4*>(Ab + 8 * s);
        nrm_i += av[0] * av[0] + av[1] * av[1] + av[2] * av[2] + av[3] * av[3];
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) {
          f32x4 bv = *reinterpret_cast<const f32x4*>(Ub + tu * 32 * LDK + 8 * s);
          nrm_u[tu] += bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2] + bv[3] * bv[3];
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[tu] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[tu], 0, 0, 0);
        }
      }
      if (c + 1 < n_chunks) {
        lstore(buf ^ 1);
        __syncthreads();
      }
    }

    // ---- distance epilogue inputs ----
    float ni_full = nrm_i + __shfl_xor(nrm_i, 32, 64);  // lanes r and r+32 hold item row r of this wave
    // ---- refresh thresholds from the shared per-user bound ----
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      const int u = tu * 32 + col;
      if (u < a.n_users) {
        float g = key_to_f32(__hip_atomic_load(a.gthr + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        thr[tu] = fmaxf(thr[tu], g);
      }
    }

#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      const int u = tu * 32 + col;
      const bool uvalid = u < a.n_users;
      float nu_full = nrm_u[tu] + __shfl_xor(nrm_u[tu], 32, 64);
      float inv_u = 1.0f / fmaxf(sqrtf(nu_full), 1e-8f);
      float sc[16];
      unsigned cmask = 0;
      const bool need_norm = a.distance != DIST_DOT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float s = acc[tu][r];
        if (need_norm) {
          float ni = __shfl(ni_full, row, 64);
          if (a.distance == DIST_COSINE) {
            s = s * inv_u * (1.0f / fmaxf(sqrtf(ni), 1e-8f));
          } else {
            s = -sqrtf(fmaxf(nu_full + ni - 2.0f * s, 0.f));
          }
        }
        sc[r] = s;
        const long long p = pos0 + wave * 32 + row;
        if (uvalid && p < a.item_end && s >= thr[tu]) cmask |= (1u << r);
      }
      if (__any(cmask != 0)) {
        // rare slow path: viewed-items check + replace-worst insert into this lane's list
        const long long lbase = ((long long)list_id * a.ub + u) * a.k;
        while (cmask != 0) {
          const int r = __ffs(cmask) - 1;
          cmask &= cmask - 1;
          float s = sc[0];
#pragma unroll
          for (int i = 1; i < 16; ++i) s = (r == i) ? sc[i] : s;  // static-index select: no scratch
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const long long p = pos0 + wave * 32 + row;
          if (!(s >= thr[tu])) continue;  // threshold may have risen inside this loop
          bool take = (cnt[tu] < a.k) || better(s, p, worst_s[tu], worst_p[tu]);
          if (!take) continue;
          const long long cid = a.whitelist ? a.whitelist[p] : p;
          if (is_filtered(a, u, cid)) continue;
          const long long rel = p - a.item_begin;  // < 2^31 enforced by the host
          if (cnt[tu] < a.k) {
            a.list_scores[lbase + cnt[tu]] = s;
            a.list_pos[lbase + cnt[tu]] = (int)rel;
            cnt[tu] += 1;
          } else {
            a.list_scores[lbase + worst_slot[tu]] = s;
            a.list_pos[lbase + worst_slot[tu]] = (int)rel;
          }
          if (cnt[tu] == a.k) {  // list full: (re)locate its worst entry and publish the bound
            float ws = a.list_scores[lbase]; long long wp = a.item_begin + a.list_pos[lbase]; int wslot = 0;
            for (int e = 1; e < a.k; ++e) {
              float es = a.list_scores[lbase + e]; long long ep = a.item_begin + a.list_pos[lbase + e];
              if (better(ws, wp, es, ep)) { ws = es; wp = ep; wslot = e; }
            }
            worst_s[tu] = ws; worst_p[tu] = wp; worst_slot[tu] = wslot;
            thr[tu] = fmaxf(thr[tu], ws);
            atomicMax(a.gthr + u, f32_to_key(ws));
          }
        }
      }
    }
  }

  // ---- publish list lengths ----
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int u = tu * 32 + col;
    if (u < a.n_users) a.list_counts[(long long)list_id * a.ub + u] = cnt[tu];
  }
}

struct MergeArgs {
  const float* list_scores; const int* list_pos; const int* list_counts;
  int n_lists; int ub; int k; int n_users;
  long long item_begin;
  // optional extra sorted list per user (result of an earlier phase), absolute positions
  const float* extra_scores; const long long* extra_pos; const int* extra_counts;
  float* compact_scores; long long* compact_pos; long long compact_cap;  // per-user scratch
  // outputs
  const long long* whitelist; int distance;
  long long* out_ids; float* out_scores; int* out_counts;      // final outputs (nullable)
  float* mid_scores; long long* mid_pos; int* mid_counts;      // phase outputs (nullable), positions
  unsigned* gthr;                                              // seed threshold (nullable)
};

__global__ __launch_bounds__(NTHREADS) void topk_merge_kernel(MergeArgs m) {
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ int s_scan[NTHREADS];
  __shared__ float s_ws[4]; __shared__ long long s_wp[4];
  __shared__ float s_bs; __shared__ long long s_bp;

  // ---- pass 1: per-thread candidate counts -> exclusive scan -> compaction ----
  int my = 0;
  for (int l = tid; l < m.n_lists; l += NTHREADS) my += m.list_counts[(long long)l * m.ub + u];
  int extra_n = (m.extra_counts != nullptr) ? m.extra_counts[u] : 0;
  if (tid == 0) my += extra_n;
  s_scan[tid] = my;
  __syncthreads();
  for (int o = 1; o < NTHREADS; o <<= 1) {
    int v = (tid >= o) ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const int total = s_scan[NTHREADS - 1];
  int ofs = s_scan[tid] - my;
  float* cs = m.compact_scores + (long long)u * m.compact_cap;
  long long* cp = m.compact_pos + (long long)u * m.compact_cap;
  if (tid == 0) {
    for (int e = 0; e < extra_n; ++e) {
      cs[ofs] = m.extra_scores[(long long)u * m.k + e];
      cp[ofs] = m.extra_pos[(long long)u * m.k + e];
      ++ofs;
    }
  }
  for (int l = tid; l < m.n_lists; l += NTHREADS) {
    const int c = m.list_counts[(long long)l * m.ub + u];
    const long long lb = ((long long)l * m.ub + u) * m.k;
    for (int e = 0; e < c; ++e) {
      cs[ofs] = m.list_scores[lb + e];
      cp[ofs] = m.item_begin + (long long)m.list_pos[lb + e];
      ++ofs;
    }
  }
  __syncthreads();

  // ---- pass 2: k rounds of block-wide arg-best below the previous winner ----
  const int n_out = total < m.k ? total : m.k;
  float prev_s = INFINITY; long long prev_p = -1;
  for (int r = 0; r < n_out; ++r) {
    float bs = -INFINITY; long long bp = 0x7fffffffffffffffLL; bool have = false;
    for (int e = tid; e < total; e += NTHREADS) {
      float s = cs[e]; long long p = cp[e];
      bool below = (r == 0) || better(prev_s, prev_p, s, p);
      if (below && (!have || better(s, p, bs, bp))) { bs = s; bp = p; have = true; }
    }
    if (!have) { bs = -INFINITY; bp = 0x7fffffffffffffffLL; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float os = __shfl_xor(bs, o, 64); long long op = __shfl_xor(bp, o, 64);
      if (better(os, op, bs, bp)) { bs = os; bp = op; }
    }
    if (lane == 0) { s_ws[wave] = bs; s_wp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = s_ws[0]; long long fp = s_wp[0];
      for (int w = 1; w < 4; ++w)
        if (better(s_ws[w], s_wp[w], fs, fp)) { fs = s_ws[w]; fp = s_wp[w]; }
      s_bs = fs; s_bp = fp;
      if (m.out_ids != nullptr) {
        m.out_ids[(long long)u * m.k + r] = m.whitelist ? m.whitelist[fp] : fp;
        m.out_scores[(long long)u * m.k + r] = (m.distance == DIST_EUCLID) ? -fs : fs;
      }
      if (m.mid_scores != nullptr) {
        m.mid_scores[(long long)u * m.k + r] = fs;
        m.mid_pos[(long long)u * m.k + r] = fp;
      }
    }
    __syncthreads();
    prev_s = s_bs; prev_p = s_bp;
  }
  if (tid == 0) {
    if (m.out_counts != nullptr) m.out_counts[u] = n_out;
    if (m.mid_counts != nullptr) m.mid_counts[u] = n_out;
    if (m.gthr != nullptr && n_out == m.k && m.k > 0) atomicMax(m.gthr + u, f32_to_key(prev_s));
  }
}

__global__ void fill_u32_kernel(unsigned* p, unsigned v, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct WsLayout {
  size_t gthr, list_scores, list_pos, list_counts, compact_scores, compact_pos, mid_scores, mid_pos, mid_counts, total;
  int n_lists_a, n_lists_b;
};

// Grid/workspace plan shared by rt_topk_workspace_bytes and rt_topk_score.
inline void plan(int ub, int k, long long n_cand, int* grid_a, int* grid_b, long long* split, WsLayout* L) {
  const int max_wg = 2 * rt_num_cus();  // 2 workgroups / CU (55 KB LDS each at TU=2)
  const long long n_blocks = (n_cand + IB - 1) / IB;
  long long ga, gb, sp;
  if (n_blocks <= (long long)max_wg * 2) {  // small catalog: single phase
    ga = n_blocks < max_wg ? (n_blocks > 0 ? n_blocks : 1) : max_wg;
    gb = 0; sp = n_cand;
  } else {  // seed the shared threshold from a prefix of one block per workgroup, then stream the rest
    ga = max_wg; sp = (long long)max_wg * IB;
    gb = max_wg;
    long long rem_blocks = (n_cand - sp + IB - 1) / IB;
    if (rem_blocks < gb) gb = rem_blocks;
  }
  *grid_a = (int)ga; *grid_b = (int)gb; *split = sp;
  L->n_lists_a = (int)ga * LISTS_PER_WG; L->n_lists_b = (int)gb * LISTS_PER_WG;
  const size_t n_lists = (size_t)(L->n_lists_a > L->n_lists_b ? L->n_lists_a : L->n_lists_b);
  size_t o = 0;
  L->gthr = o; o = align_up(o + (size_t)ub * 4, 256);
  L->list_scores = o; o = align_up(o + n_lists * ub * (size_t)k * 4, 256);
  L->list_pos = o; o = align_up(o + n_lists * ub * (size_t)k * 4, 256);
  L->list_counts = o; o = align_up(o + n_lists * ub * 4, 256);
  const size_t cap = n_lists * (size_t)k + (size_t)k;
  L->compact_scores = o; o = align_up(o + (size_t)ub * cap * 4, 256);
  L->compact_pos = o; o = align_up(o + (size_t)ub * cap * 8, 256);
  L->mid_scores = o; o = align_up(o + (size_t)ub * k * 4, 256);
  L->mid_pos = o; o = align_up(o + (size_t)ub * k * 8, 256);
  L->mid_counts = o; o = align_up(o + (size_t)ub * 4, 256);
  L->total = o;
}

inline int pick_tu(int users_per_pass, int n_users) {
  int upp = users_per_pass;
  if (upp <= 0) upp = 64;  // fp32 MFMA vs HBM balance point on gfx950
  if (upp > 128) upp = 128;
  int tu = upp <= 32 ? 1 : (upp <= 64 ? 2 : 4);
  // do not pay for empty user tiles
  if (n_users <= 32) tu = 1; else if (n_users <= 64 && tu > 2) tu = 2;
  return tu;
}

template <int TU>
int launch_batch(TopkArgs a, MergeArgs m, int grid_a, int grid_b, long long split, long long n_cand,
                 const WsLayout& L, char* ws, hipStream_t stream) {
  constexpr int UB = 32 * TU;
  const size_t lds = (size_t)(2 * IB * LDK + 2 * UB * LDK) * sizeof(float);
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_partial_kernel<TU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RT_ERR_LAUNCH;
      attr_set = true;
    }
  }
  unsigned* gthr = reinterpret_cast<unsigned*>(ws + L.gthr);
  // gthr := key(-inf)
  fill_u32_kernel<<<1, 128, 0, stream>>>(gthr, 0x007FFFFFu, UB);
  a.ub = UB; a.gthr = gthr; a.list_base = 0;
  a.list_scores = reinterpret_cast<float*>(ws + L.list_scores);
  a.list_pos = reinterpret_cast<int*>(ws + L.list_pos);
  a.list_counts = reinterpret_cast<int*>(ws + L.list_counts);
  m.list_scores = a.list_scores; m.list_pos = a.list_pos; m.list_counts = a.list_counts;
  m.ub = UB;
  m.compact_scores = reinterpret_cast<float*>(ws + L.compact_scores);
  m.compact_pos = reinterpret_cast<long long*>(ws + L.compact_pos);
  m.compact_cap = (long long)((L.n_lists_a > L.n_lists_b ? L.n_lists_a : L.n_lists_b)) * a.k + a.k;

  // phase A: [0, split)
  a.item_begin = 0; a.item_end = split;
  topk_partial_kernel<TU><<<grid_a, NTHREADS, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  MergeArgs ma = m;
  ma.n_lists = L.n_lists_a; ma.item_begin = 0;
  ma.extra_scores = nullptr; ma.extra_pos = nullptr; ma.extra_counts = nullptr;
  if (grid_b == 0) {
    ma.mid_scores = nullptr; ma.mid_pos = nullptr; ma.mid_counts = nullptr; ma.gthr = nullptr;
    topk_merge_kernel<<<a.n_users, NTHREADS, 0, stream>>>(ma);
    RT_CHECK_LAUNCH();
    return RT_OK;
  }
  ma.out_ids = nullptr; ma.out_scores = nullptr; ma.out_counts = nullptr;
  ma.mid_scores = reinterpret_cast<float*>(ws + L.mid_scores);
  ma.mid_pos = reinterpret_cast<long long*>(ws + L.mid_pos);
  ma.mid_counts = reinterpret_cast<int*>(ws + L.mid_counts);
  ma.gthr = gthr;
  topk_merge_kernel<<<a.n_users, NTHREADS, 0, stream>>>(ma);
  RT_CHECK_LAUNCH();
  // phase B: [split, n_cand) with the seeded threshold
  a.item_begin = split; a.item_end = n_cand;
  topk_partial_kernel<TU><<<grid_b, NTHREADS, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  MergeArgs mb = m;
  mb.n_lists = L.n_lists_b; mb.item_begin = split;
  mb.extra_scores = ma.mid_scores; mb.extra_pos = ma.mid_pos; mb.extra_counts = ma.mid_counts;
  mb.mid_scores = nullptr; mb.mid_pos = nullptr; mb.mid_counts = nullptr; mb.gthr = nullptr;
  topk_merge_kernel<<<a.n_users, NTHREADS, 0, stream>>>(mb);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace

extern "C" {

size_t rt_topk_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t users_per_pass) {
  if (n_users <= 0 || n_candidates <= 0 || k <= 0) return 256;
  int tu = pick_tu(users_per_pass, n_users);
  int ga, gb; long long split; WsLayout L;
  long long kk = k < n_candidates ? k : n_candidates;
  plan(32 * tu, (int)kk, n_candidates, &ga, &gb, &split, &L);
  return L.total;
}

int rt_topk_score(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users,
                  const float* items, int64_t item_stride, const int64_t* whitelist, int64_t n_candidates,
                  int32_t d, int32_t distance, int32_t k,
                  const int64_t* filt_indptr, const int32_t* filt_indices,
                  int64_t* out_ids, float* out_scores, int32_t* out_counts,
                  void* workspace, size_t workspace_bytes, int32_t users_per_pass, hipStream_t stream) {
  if (n_users < 0 || n_candidates < 0 || d <= 0 || (d & 3) != 0 || k <= 0) return RT_ERR_INVALID_ARG;
  if (distance < DIST_DOT || distance > DIST_EUCLID) return RT_ERR_INVALID_ARG;
  if ((user_stride & 3) != 0 || (item_stride & 3) != 0) return RT_ERR_INVALID_ARG;
  if (((uintptr_t)users & 15) != 0 || ((uintptr_t)items & 15) != 0) return RT_ERR_INVALID_ARG;
  if (n_candidates >= (1LL << 31)) return RT_ERR_UNSUPPORTED;
  if (k > n_candidates) return RT_ERR_INVALID_ARG;  // caller clamps: k = min(k, n_candidates)
  if (n_users == 0) return RT_OK;
  if (n_candidates == 0) {
    return hipMemsetAsync(out_counts, 0, sizeof(int32_t) * (size_t)n_users, stream) == hipSuccess ? RT_OK : RT_ERR_LAUNCH;
  }
  const int tu = pick_tu(users_per_pass, n_users);
  const int ub = 32 * tu;
  int ga, gb; long long split; WsLayout L;
  plan(ub, k, n_candidates, &ga, &gb, &split, &L);
  if (workspace == nullptr || workspace_bytes < L.total) return RT_ERR_WORKSPACE;
  char* ws = reinterpret_cast<char*>(workspace);

  for (int u0 = 0; u0 < n_users; u0 += ub) {
    const int nb = (n_users - u0) < ub ? (n_users - u0) : ub;
    TopkArgs a{};
    a.users = user_rows ? users : users + (long long)u0 * user_stride;
    a.user_stride = user_stride;
    a.user_rows = user_rows ? reinterpret_cast<const long long*>(user_rows) + u0 : nullptr;
    a.n_users = nb;
    a.items = items; a.item_stride = item_stride;
    a.whitelist = reinterpret_cast<const long long*>(whitelist);
    a.d = d; a.distance = distance; a.k = k;
    a.filt_indptr = filt_indptr ? reinterpret_cast<const long long*>(filt_indptr) + u0 : nullptr;
    a.filt_indices = filt_indices;
    MergeArgs m{};
    m.k = k; m.n_users = nb; m.whitelist = a.whitelist; m.distance = distance;
    m.out_ids = reinterpret_cast<long long*>(out_ids) + (long long)u0 * k;
    m.out_scores = out_scores + (long long)u0 * k;
    m.out_counts = out_counts + u0;
    int rc;
    if (tu == 1) rc = launch_batch<1>(a, m, ga, gb, split, n_candidates, L, ws, stream);
    else if (tu == 2) rc = launch_batch<2>(a, m, ga, gb, split, n_candidates, L, ws, stream);
    else rc = launch_batch<4>(a, m, ga, gb, split, n_candidates, L, ws, stream);
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

}  // extern "C"
