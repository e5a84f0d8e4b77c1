// K12 — exact full-catalog top-k scorer (recommend() hot loop).
//
// Replaces the per-128-user loop of the reference `TorchRanker.rank`
// (rectools/models/rank/rank_torch.py:122-155): `U @ I.T` -> masked_fill(-inf) from a dense
// `csr.toarray()` -> `torch.topk` -> D2H.  Here the catalog is streamed from HBM (or L2 for small
// catalogs), scores live only in MFMA accumulators, and per-lane top-k lists absorb them.
//
// Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain), so scores match an
// fp32 reference up to the rounding of a different summation order.
//
// Structure (gfx950):
//   * grid = (S catalog segments) x (user tiles of UB = 32*TU users).  Workgroup (sx, ty) = 4 waves walks
//     item blocks sx, sx+S, ... of 128 catalog rows (32 per wave) for the users of tile ty.  S is chosen
//     so that ~2 workgroups per CU are resident; many-user launches (recommend for a whole user base
//     against an L2-resident catalog) get S = 1 and walk the full catalog per tile.
//   * MFMA operand roles: A = items (rows i), B = users (cols j); lane l feeds A[item l&31][k] and
//     B[k][user l&31] for k = 8s + 4(l>>5) + t, t = 0..3 — a fixed permutation of the reduction index
//     that lets every lane fetch its 4 MFMA steps with ONE ds_read_b128 per operand.
//   * D layout (col = lane&31 = user, row = (r&3)+8(r>>2)+4(lane>>5) = item) puts 16 items of ONE user
//     in each lane: selection is lane-local, no cross-lane traffic.
//   * two staging engines for the 32-float k-chunks:
//       - `stream` (d % 32 == 0): global_load_lds_dwordx4 (LDS-DMA) into an NS-deep ring of unpadded,
//         XOR-swizzled tiles, counted vmcnt + raw s_barrier, NS-1 chunks in flight across block seams;
//       - `staged` (any d % 4 == 0): HBM -> VGPR -> padded LDS, double-buffered.
//     Chunk order is rotated per workgroup so that concurrently running workgroups do not all hit the
//     same 128-byte column of their 2 KB-strided rows (HBM channel camping).
//   * selection: `score >= thr` fast path in registers; rare slow path does the viewed-items check
//     (binary search in the user's CSR row) and a replace-worst insert into the lane's list (global
//     workspace, L2 resident).  A per-user threshold (atomicMax of every full list's worst score) is
//     shared by all workgroups.
//   * merge kernel: one wave per user compacts that user's lists and extracts the k best in order
//     (score desc, position asc — the tie rule of oracle/ranker_oracle.py).
#include <stdlib.h>

#include "rt_common.h"

namespace {

constexpr int IB = 128;       // catalog rows per item block
constexpr int KC = 32;        // floats per k-chunk
constexpr int K_LDS_LISTS = 16;  // selection lists live in LDS up to this k
constexpr int LDK = KC + 4;   // padded LDS row stride (floats) of the `staged` engine
constexpr int NTHREADS = 256;
constexpr int LISTS_PER_WG = 8;  // 4 waves x 2 half-waves
// threads of the per-user merge / seed workgroups: 1024 when a user has many lists, 256 otherwise

enum { DIST_DOT = 0, DIST_COSINE = 1, DIST_EUCLID = 2 };

struct TopkArgs {
  const float* users; long long user_stride; const long long* user_rows; int n_users;
  const float* items; long long item_stride; const long long* whitelist;
  long long n_cand; long long id_offset;  // whitelist == NULL: candidate p has item id p + id_offset
  int d; int distance; int k;
  const long long* filt_indptr; const int* filt_indices;
  const int* filt_hash; int filt_u0;   // optional per-user open-addressing tables (rt_filter_hash_build), first user of the launch
  float* list_scores; int* list_pos; int* list_counts;  // [n_lists][n_users_pad][k], [n_lists][n_users_pad]
  int n_users_pad;
  unsigned* gthr;  // [n_users_pad] ordered keys
  long long blk_begin, blk_end;  // item blocks handled by this launch (phase)
  int n_seg;                     // workgroups sx < n_seg share those blocks round-robin (others idle)
  int resume;                    // 1: lists already hold entries from an earlier phase
  int rotate;
  int debug;  // ablation switch (RT_TOPK_DEBUG): 1 = skip selection
  int list_base;                 // 16-user tile: merged list of workgroup sx is list `list_base + sx` of `n_lists_total`
  int n_lists_total;             //   merged lists are stored [user][list][k] (a user's lists are contiguous for the selection)
  int use_bound;                 // 16-user tile: 0 = seeding prefix (the shared bound is neither read nor published)
  int bloom;                     // engine 2: 1 = a 1024-bit Bloom filter per user of the tile sits in LDS behind the lists
  int h_only;                    // HM kernels: 1 = the images hold ONE bf16 (round-to-nearest) per value (rows of d / 2 words): one MFMA per slot
  int xmap_gx, xmap_gy;          // topk_coarse_frag_kernel on a 1-D grid: the (segments, user tiles) grid it stands for (0, 0: plain 2-D grid)
};

// acc + |v|^2 as one fixed fma chain: engine 2 and the two-stage exact pass (topk_replay_kernel) must round a row norm alike
__device__ __forceinline__ float sumsq4(float acc, const f32x4& v) {
  acc = __builtin_fmaf(v[0], v[0], acc); acc = __builtin_fmaf(v[1], v[1], acc);
  acc = __builtin_fmaf(v[2], v[2], acc); return __builtin_fmaf(v[3], v[3], acc);
}

__device__ __forceinline__ bool better(float s, long long p, float s2, long long p2) {
  return (s > s2) || (s == s2 && p < p2);
}

// Per-user hash set of the filter indices: user u owns the slots [4*indptr[u] + 4*u, +4*cnt_u + 4) of one int array and
// uses the largest power of two inside them (load factor <= 1/2), linear probing, -1 = empty.  A membership test is
// 1-2 loads instead of the log2(cnt) DEPENDENT loads of a binary search over the CSR row — the dominant cost of the
// selection slow path when thousands of users are ranked against a small catalog (recommend()).
__device__ __forceinline__ unsigned filt_hash_of(unsigned cid, int lg) { return (cid * 0x9E3779B1u) >> (32 - lg); }
__device__ __forceinline__ int filt_hash_lg(long long cnt) { return 31 - __clz((int)(4 * cnt + 4)); }

// Is candidate id `cid` among the filter indices of this user?
__device__ __forceinline__ bool is_filtered(const TopkArgs& a, int u, long long cid) {
  if (a.filt_indptr == nullptr) return false;
  long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
  if (a.filt_hash != nullptr) {
    if (hi == lo) return false;
    const int lg = filt_hash_lg(hi - lo);
    const int* tab = a.filt_hash + 4 * lo + 4 * (long long)(u + a.filt_u0);
    unsigned h = filt_hash_of((unsigned)cid, lg);
    const unsigned mask = (1u << lg) - 1u;
    for (;;) {
      const int v = tab[h];
      if (v == (int)cid) return true;
      if (v < 0) return false;
      h = (h + 1) & mask;
    }
  }
  const long long end = hi;
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    long long v = (long long)a.filt_indices[mid];
    if (v < cid) lo = mid + 1; else hi = mid;
  }
  return lo < end && (long long)a.filt_indices[lo] == cid;
}

// one wave per user: insert the row's indices into the user's table (slots pre-filled with -1)
__global__ __launch_bounds__(256) void filter_hash_build_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                int n_users, int* __restrict__ hash) {
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= n_users) return;
  const long long lo = indptr[u], hi = indptr[u + 1];
  if (hi == lo) return;
  const int lg = filt_hash_lg(hi - lo);
  int* tab = hash + 4 * lo + 4 * (long long)u;
  const unsigned mask = (1u << lg) - 1u;
  for (long long e = lo + lane; e < hi; e += 64) {
    const int cid = indices[e];
    unsigned h = filt_hash_of((unsigned)cid, lg);
    for (;;) {
      const int prev = atomicCAS(tab + h, -1, cid);
      if (prev == -1 || prev == cid) break;
      h = (h + 1) & mask;
    }
  }
}

// List storage pointer types.  LDS lists MUST be addressed through address_space(3) pointers: with generic
// pointers hipcc emits flat stores, cannot prove they miss the DMA ring, and puts `s_waitcnt vmcnt(0)` in
// front of the fragment reads of every chunk — which drains the LDS-DMA pipeline.
template <bool LL> struct ListTypes { typedef float* fptr; typedef int* iptr; };
template <> struct ListTypes<true> {
  typedef __attribute__((address_space(3))) float* fptr;
  typedef __attribute__((address_space(3))) int* iptr;
};

// Per-lane selection state: one list per (segment, wave, half-wave, user).
template <int TU, bool LL>
struct SelState {
  typedef typename ListTypes<LL>::fptr fptr;
  typedef typename ListTypes<LL>::iptr iptr;
  float worst_s[TU]; long long worst_p[TU]; int worst_slot[TU]; int cnt[TU]; float thr[TU];
  float g_seen[TU];          // last value of the shared bound this lane has observed / published
  fptr ls[TU]; iptr lp[TU];  // this lane's list storage per user tile (LDS for small k, else global)
  long long fbase[TU]; int flg[TU];   // this lane's filter hash table (bind_filter)
  typedef __attribute__((address_space(3))) const unsigned* bptr;
  bptr bl[TU];               // this lane's user's Bloom filter words (32 x 32 bits) when TopkArgs::bloom
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      worst_s[tu] = -INFINITY; worst_p[tu] = -1; worst_slot[tu] = 0; cnt[tu] = 0; thr[tu] = -INFINITY;
      g_seen[tu] = -INFINITY; ls[tu] = nullptr; lp[tu] = nullptr; bl[tu] = nullptr;
    }
  }
  // continue lists left by an earlier phase (global image; copied into the bound storage if that is LDS)
  __device__ __forceinline__ void resume(const TopkArgs& a, int list_id, int user0, int lane, bool copy) {
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      const int u = user0 + tu * 32 + (lane & 31);
      if (u >= a.n_users) continue;
      const long long li = (long long)list_id * a.n_users_pad + u;
      const int c = a.list_counts[li];
      cnt[tu] = c;
      if (copy) {
        for (int e = 0; e < c; ++e) { ls[tu][e] = a.list_scores[li * a.k + e]; lp[tu][e] = a.list_pos[li * a.k + e]; }
      }
      if (c == a.k) {
        float ws = ls[tu][0]; long long wp = lp[tu][0]; int wslot = 0;
        for (int e = 1; e < c; ++e) {
          float es = ls[tu][e]; long long ep = lp[tu][e];
          if ((ws > es) || (ws == es && wp < ep)) { ws = es; wp = ep; wslot = e; }
        }
        worst_s[tu] = ws; worst_p[tu] = wp; worst_slot[tu] = wslot; thr[tu] = ws;
      }
    }
  }
  // global-memory lists: [n_lists][n_users_pad][k]
  __device__ __forceinline__ void bind_global(const TopkArgs& a, int list_id, int user0, int lane) {
    if constexpr (!LL) {
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        const long long lbase = ((long long)list_id * a.n_users_pad + (user0 + tu * 32 + (lane & 31))) * a.k;
        ls[tu] = a.list_scores + lbase; lp[tu] = a.list_pos + lbase;
      }
    }
  }
  // LDS lists: scores [8][UB][kp] then positions [8][UB][kp] behind the staging ring, kp = k rounded up to 4 so that a
  // list is a whole number of 16-byte groups (the worst-entry scan reads it with ds_read_b128); the pad entries hold
  // +inf scores: they are never the worst entry and never counted
  __device__ __forceinline__ void bind_lds(float* base, int k, int wave, int lane) {
    if constexpr (LL) {
      constexpr int UB = 32 * TU;
      const int kp = (k + 3) & ~3;
      fptr b = (fptr)base;
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) {
        const int L = ((wave * 2 + (lane >> 5)) * UB + tu * 32 + (lane & 31)) * kp;
        ls[tu] = b + L; lp[tu] = (iptr)(b + LISTS_PER_WG * UB * kp) + L;
        for (int e = k; e < kp; ++e) { ls[tu][e] = INFINITY; lp[tu][e] = -1; }
      }
    }
  }
  // per-lane view of this user's filter hash table (base slot, log2 size; lg < 0: no filter rows)
  __device__ __forceinline__ void bind_filter(const TopkArgs& a, int user0, int lane) {
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      fbase[tu] = 0; flg[tu] = -1;
      const int u = user0 + tu * 32 + (lane & 31);
      if (a.filt_hash != nullptr && u < a.n_users) {
        const long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
        if (hi > lo) { fbase[tu] = 4 * lo + 4 * (long long)(u + a.filt_u0); flg[tu] = filt_hash_lg(hi - lo); }
      }
    }
  }
};

// Absorb one finished item block (accumulators `acc`) into the lane's lists.
template <int TU, bool LL, bool GL = false>
__device__ __forceinline__ void select_block(const TopkArgs& a, SelState<TU, LL>& st, const f32x16 (&acc)[TU],
                                             float nrm_i, const float (&nrm_u)[TU], long long pos0,
                                             int list_id, int user0, int lane, int wave,
                                             const unsigned* g_lds = nullptr) {
  const int col = lane & 31, half = lane >> 5;
  const bool need_norm = a.distance != DIST_DOT;
  float ni_full = 0.f;
  if (need_norm) ni_full = nrm_i + __shfl_xor(nrm_i, 32, 64);  // lanes r and r+32 hold item row r
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int u = user0 + tu * 32 + col;
    const bool uvalid = u < a.n_users;
    if (uvalid) {  // refresh from the shared per-user bound (LDS copy brought in by the DMA ring, if any)
      float g;
      if constexpr (GL) g = key_to_f32(g_lds[tu * 32 + col]);
      else g = key_to_f32(__hip_atomic_load(a.gthr + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      st.g_seen[tu] = fmaxf(st.g_seen[tu], g);
      st.thr[tu] = fmaxf(st.thr[tu], g);
    }
    float nu_full = 0.f, inv_u = 1.f;
    if (need_norm) {
      nu_full = nrm_u[tu] + __shfl_xor(nrm_u[tu], 32, 64);
      inv_u = 1.0f / fmaxf(sqrtf(nu_full), 1e-8f);
    }
    float sc[16];
    unsigned cmask = 0;
    if (!need_norm) {
      // dot products: ONE maximum per tile decides whether any of the lane's 16 scores can enter its list (a measured 10 ms of the
      // 4,096-user coarse pass went into 16 threshold + bounds tests per tile and block that almost never fire: with one wave per
      // SIMD the vector work of the selection adds to the matrix time instead of hiding under it)
      float mx = fmaxf(fmaxf(fmaxf(acc[tu][0], acc[tu][1]), fmaxf(acc[tu][2], acc[tu][3])), fmaxf(fmaxf(acc[tu][4], acc[tu][5]), fmaxf(acc[tu][6], acc[tu][7])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(acc[tu][8], acc[tu][9]), fmaxf(acc[tu][10], acc[tu][11])), fmaxf(fmaxf(acc[tu][12], acc[tu][13]), fmaxf(acc[tu][14], acc[tu][15]))));
      if (!__any(uvalid && mx >= st.thr[tu])) continue;
    }
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
      if (uvalid && p < a.n_cand && s >= st.thr[tu]) cmask |= (1u << r);
    }
#ifdef RT_ABLATION_BUILD
    if (a.debug & 2) cmask = 0;      // (thresholds tested, nothing ever inserted)
#endif
    if (__any(cmask != 0)) {
      // rare slow path: viewed-items check + replace-worst insert into this lane's list
      const typename ListTypes<LL>::fptr lsc = st.ls[tu]; const typename ListTypes<LL>::iptr lps = st.lp[tu];
      while (cmask != 0) {
        const int r = __ffs(cmask) - 1;
        cmask &= cmask - 1;
        float s = sc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) s = (r == i) ? sc[i] : s;  // static-index select: no scratch
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const long long p = pos0 + wave * 32 + row;
        if (!(s >= st.thr[tu])) continue;  // threshold may have risen inside this loop
        bool take = (st.cnt[tu] < a.k) || better(s, p, st.worst_s[tu], st.worst_p[tu]);
        if (!take) continue;
        const long long cid = a.whitelist ? a.whitelist[p] : p + a.id_offset;
        if (a.filt_hash != nullptr) {   // O(1) probe of the user's hash set (table view cached per lane)
          bool hit = false;
          bool maybe = st.flg[tu] >= 0;
          if (a.bloom && maybe) {   // two bits of the user's LDS Bloom filter: a clear bit proves "not viewed" without the
            // global-memory probe below (a dependent L2 / Infinity-Cache round trip inside the serialised slow path)
            const unsigned h1 = ((unsigned)cid * 0x9E3779B1u) >> 22, h2 = ((unsigned)cid * 0x85EBCA77u) >> 22;
            maybe = ((st.bl[tu][h1 >> 5] >> (h1 & 31)) & (st.bl[tu][h2 >> 5] >> (h2 & 31)) & 1u) != 0;
          }
          if (maybe) {
            const int* tab = a.filt_hash + st.fbase[tu];
            const unsigned mask = (1u << st.flg[tu]) - 1u;
            unsigned h = filt_hash_of((unsigned)cid, st.flg[tu]);
            for (;;) {
              const int v = tab[h];
              if (v == (int)cid) { hit = true; break; }
              if (v < 0) break;
              h = (h + 1) & mask;
            }
          }
          if (hit) continue;
        } else if (is_filtered(a, u, cid)) continue;
        if (st.cnt[tu] < a.k) {
          lsc[st.cnt[tu]] = s;
          lps[st.cnt[tu]] = (int)p;
          st.cnt[tu] += 1;
        } else {
          lsc[st.worst_slot[tu]] = s;
          lps[st.worst_slot[tu]] = (int)p;
        }
        if (st.cnt[tu] == a.k) {  // list full: (re)locate its worst entry and publish the bound
          float ws; long long wp; int wslot;
          if constexpr (LL) {
            // whole list in flight at once (<= 4 + 4 ds_read_b128), then a register scan: a scalar loop over a runtime
            // k serialises k dependent LDS round trips on every insert
            typedef __attribute__((address_space(3))) f32x4* fptr4;
            typedef __attribute__((address_space(3))) i32x4* iptr4;
            const int kp = (a.k + 3) & ~3;
            f32x4 sv[K_LDS_LISTS / 4]; i32x4 pv[K_LDS_LISTS / 4];
#pragma unroll
            for (int q = 0; q < K_LDS_LISTS / 4; ++q)
              if (4 * q < kp) { sv[q] = *((fptr4)lsc + q); pv[q] = *((iptr4)lps + q); }
            ws = INFINITY; wp = -1; wslot = 0;
#pragma unroll
            for (int q = 0; q < K_LDS_LISTS / 4; ++q)
              if (4 * q < kp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float es = sv[q][i]; const long long ep = pv[q][i];
                  if (4 * q + i < a.k && ((4 * q + i == 0) || better(ws, wp, es, ep))) { ws = es; wp = ep; wslot = 4 * q + i; }
                }
              }
          } else {
            ws = lsc[0]; wp = lps[0]; wslot = 0;
            for (int e = 1; e < a.k; ++e) {
              float es = lsc[e]; long long ep = lps[e];
              if (better(ws, wp, es, ep)) { ws = es; wp = ep; wslot = e; }
            }
          }
          st.worst_s[tu] = ws; st.worst_p[tu] = wp; st.worst_slot[tu] = wslot;
          st.thr[tu] = fmaxf(st.thr[tu], ws);
        }
      }
      // publish at most once per block and list, and only a bound that beats the shared one
      if (st.cnt[tu] == a.k && st.worst_s[tu] > st.g_seen[tu]) {
        atomicMax(a.gthr + u, f32_to_key(st.worst_s[tu]));
        st.g_seen[tu] = st.worst_s[tu];
      }
      // Leave the (rare) slow path with an empty VMEM scoreboard: otherwise hipcc guards register reuse at
      // the loop head with an unconditional `s_waitcnt vmcnt(0)`, which drains the DMA ring on EVERY block.
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
    }
  }
}

template <int TU, bool LL>
__device__ __forceinline__ void publish_counts(const TopkArgs& a, const SelState<TU, LL>& st, int list_id, int user0,
                                               int lane, bool flush_lds_lists) {
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int u = user0 + tu * 32 + (lane & 31);
    if (u < a.n_users) {
      a.list_counts[(long long)list_id * a.n_users_pad + u] = st.cnt[tu];
      if (flush_lds_lists) {
        const long long lbase = ((long long)list_id * a.n_users_pad + u) * a.k;
        for (int e = 0; e < st.cnt[tu]; ++e) {
          a.list_scores[lbase + e] = st.ls[tu][e];
          a.list_pos[lbase + e] = st.lp[tu][e];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Engine 1: register-staged, padded LDS, double-buffered.  Handles any d % 4 == 0.
// ------------------------------------------------------------------------------------------------
template <int TU>
__global__ __launch_bounds__(NTHREADS) void topk_staged_kernel(TopkArgs a) {
  constexpr int UB = 32 * TU;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][IB][LDK]
  float* Us = smem + 2 * IB * LDK;       // [2][UB][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int col = lane & 31;
  const int half = lane >> 5;
  const int S = a.n_seg;
  const int user0 = blockIdx.y * UB;

  const int n_chunks = (a.d + KC - 1) / KC;

  const int list_id = blockIdx.x * LISTS_PER_WG + wave * 2 + half;
  SelState<TU, false> st;
  st.init();
  st.bind_global(a, list_id, user0, lane);
  st.bind_filter(a, user0, lane);
  if (a.resume) st.resume(a, list_id, user0, lane, false);

  const float* urow[TU]; int ur_r[TU];
  const int c4 = tid & 7;
#pragma unroll
  for (int j = 0; j < TU; ++j) {
    int r = (tid >> 3) + 32 * j;
    ur_r[j] = r;
    int u = user0 + r;
    if (u < a.n_users) {
      long long src = a.user_rows ? a.user_rows[u] : (long long)u;
      urow[j] = a.users + src * a.user_stride;
    } else {
      urow[j] = nullptr;
    }
  }

  for (long long blk = a.blk_begin + blockIdx.x; blk < a.blk_end && (int)blockIdx.x < S; blk += S) {
    const long long pos0 = blk * IB;
    // chunk order rotated by the ITEM BLOCK (not by the workgroup): concurrent workgroups still start on different
    // 128-byte columns, and the k order in which a score is summed no longer depends on the launch geometry
    const int rot = a.rotate ? (int)((unsigned long long)(blk * 5) % (unsigned)n_chunks) : 0;
    const float* irow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long long p = pos0 + (tid >> 3) + 32 * j;
      if (p < a.n_cand) {
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
      int cc = c + rot; if (cc >= n_chunks) cc -= n_chunks;
      const int kofs = cc * KC + c4 * 4;
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
    select_block<TU, false>(a, st, acc, nrm_i, nrm_u, pos0, list_id, user0, lane, wave);
  }
  publish_counts<TU, false>(a, st, list_id, user0, lane, false);
}

// ------------------------------------------------------------------------------------------------
// Engine 2: LDS-DMA ring (global_load_lds_dwordx4), NS stages, counted vmcnt, raw s_barrier.
// Requires d % 32 == 0.  LDS image per stage: items [128][32] + users [UB][32] floats, unpadded;
// 16-byte slot c4 of row r is stored at physical slot c4 ^ ((r>>1)&7) (conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------------
// LDS byte address of a __shared__ pointer (wave-uniform values only: it is moved into M0).
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB).
// Issued through inline asm so that hipcc does not see an asynchronous LDS write: with the builtin it
// drains vmcnt(0) in front of the ds_reads of OTHER ring slots (it cannot prove they do not alias).
// The DMA is therefore invisible to the compiler's waitcnt bookkeeping; completion is waited for by
// hand (wait_vmcnt<N>) — in-order return makes every compiler-inserted vmcnt wait at least as strong.
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ void dma4(const unsigned* gsrc, unsigned lds_dst) {  // 64 lanes x 4 B
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NLD = 0: every wave both issues its share of the DMA ring and computes.  NLD = 2: waves 4,5 are LOADERS — they issue
// all LDS-DMA pieces and wait for them — and waves 0..3 only compute.  An LDS-DMA instruction costs 60-185 issue
// cycles in the wave that issues it (address VALU + the piece itself); with 6 pieces per 16 MFMAs that is a third of
// a compute wave's time at 32 users per launch, exactly the regime where the kernel should be HBM-bound.  A loader
// shares its SIMD's issue port with one compute wave, but its pieces overlap that wave's MFMA execution.
// HM = true: the coarse pass of the two-stage top-k (rt_topk_score_two_stage).  Users and catalog are "hm images" (rt_to_hm_rows): every
// fp32 value x replaced by the 32-bit word (h << 16) | m, h = bf16 truncation of x, m = bf16 truncation of x - h (x - h - m < 2^-15 |x|).
// Same geometry as the fp32 rows — same ring, same strides, same swizzle — and one 16-byte LDS read is one k = 16 operand of
// v_mfma_f32_32x32x16_bf16 holding (m, h) pairs of four k positions: A . B sums h h' + m m', A with its halves swapped . B sums
// h m' + m h': two bf16 instructions (a quarter of the matrix-pipe time of the four f32-input ones) give (h + m)(h' + m').  Dot products only.
template <int TU, int NS, bool WL, bool LL, int NLD, bool HM = false>
__global__ __launch_bounds__(NTHREADS + NLD * 64) void topk_stream_kernel(TopkArgs a) {
  constexpr int NISS = NLD ? NLD : 4;      // issuing waves
  constexpr int IPI = 16 / NISS;           // item pieces per issuer and stage (128 rows x 32 floats = 16 KiB = 16 pieces)
  constexpr int UPI = 4 * TU / NISS;       // user pieces per issuer and stage
  constexpr int UB = 32 * TU;
  constexpr int SA = IB * KC;            // floats per stage, items
  constexpr int SU = UB * KC;            // floats per stage, users
  constexpr int SG = 128;                // shared-bound copy (uints), refreshed with every stage
  constexpr int STAGE = SA + SU + SG;
  constexpr int NL = IPI + UPI + 1;      // LDS-DMA instructions per issuing wave per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NS][STAGE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;
  const int half = lane >> 5;
  const int S = a.n_seg;
  const int user0 = blockIdx.y * UB;

  const long long n_blocks = a.blk_end - a.blk_begin;
  const int n_chunks = a.d / KC;
  int rot = 0;   // chunk rotation of the block being ISSUED: a function of the item block only (see topk_staged_kernel)
  const long long my_blocks = ((int)blockIdx.x < S && n_blocks > blockIdx.x) ? (n_blocks - blockIdx.x + S - 1) / S : 0;
  const long long T = my_blocks * n_chunks;  // flattened (block, chunk) steps of this workgroup

  const bool issuer = NLD ? wave >= 4 : true;
  const bool computes = NLD ? wave < 4 : true;
  const int iw = NLD ? (wave >= 4 ? wave - 4 : 0) : wave;   // index among the issuing waves (0 in a compute wave of the loader form: its
  //                                                           source addresses below are never used, but they are COMPUTED — a negative index
  //                                                           read user_rows[-32 .. -1], a fault when the array starts an allocator segment)
  const int cwave = computes ? wave : 0;
  const int list_id = blockIdx.x * LISTS_PER_WG + cwave * 2 + half;
  SelState<TU, LL> st;
  st.init();
  if (computes) {
    st.bind_lds(smem + NS * STAGE, a.k, cwave, lane);
    st.bind_global(a, list_id, user0, lane);
    st.bind_filter(a, user0, lane);
    if (a.resume) st.resume(a, list_id, user0, lane, LL);
  }
  if (T == 0) { if (computes && !a.resume) publish_counts<TU, LL>(a, st, list_id, user0, lane, false); return; }

  if (a.bloom) {
    // Per-user Bloom filters of the viewed items (1024 bits, two hash functions: ~6 % false positives at 144 items) behind
    // the lists: built once per workgroup from the CSR rows of its UB users; the slow path consults them before the exact
    // hash-set probe in global memory.  Every wave (loaders included) takes part; both barriers precede the ring.
    typedef __attribute__((address_space(3))) unsigned* wptr;
    const int kp_l = (a.k + 3) & ~3;
    wptr bl = (wptr)(smem + NS * STAGE + (LL ? 2 * LISTS_PER_WG * UB * kp_l : 0));
    constexpr int NWV = (NTHREADS + NLD * 64) / 64;
    for (int i = tid; i < UB * 32; i += NWV * 64) bl[i] = 0u;
    __syncthreads();
    for (int ul = wave; ul < UB; ul += NWV) {
      const int u = user0 + ul;
      if (u >= a.n_users) continue;
      const long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
      for (long long e = lo + lane; e < hi; e += 64) {
        const unsigned cid = (unsigned)a.filt_indices[e];
        const unsigned h1 = (cid * 0x9E3779B1u) >> 22, h2 = (cid * 0x85EBCA77u) >> 22;
        atomicOr((unsigned*)(bl + ul * 32 + (h1 >> 5)), 1u << (h1 & 31));   // generic pointer: before the ring starts
        atomicOr((unsigned*)(bl + ul * 32 + (h2 >> 5)), 1u << (h2 & 31));
      }
    }
    __syncthreads();
    if (computes) {
#pragma unroll
      for (int tu = 0; tu < TU; ++tu) st.bl[tu] = (typename SelState<TU, LL>::bptr)(bl + (tu * 32 + (lane & 31)) * 32);
    }
  }

  // ---- DMA source assignment ----
  // items: piece j of issuer w fills rows (w*IPI+j)*8 .. +7 ; lane -> row + (lane>>3), slot lane&7
  const int l_row8 = lane >> 3, l_slot = lane & 7;
  int i_row[IPI]; int i_colofs[IPI];
#pragma unroll
  for (int j = 0; j < IPI; ++j) {
    i_row[j] = (iw * IPI + j) * 8 + l_row8;
    i_colofs[j] = (l_slot ^ ((i_row[j] >> 1) & 7)) * 4;  // logical float offset inside the chunk
  }
  // users: piece j of issuer w fills user rows (j*NISS + w)*8 .. +7   (UPI pieces per issuer)
  const float* u_src[UPI]; int u_row[UPI];
#pragma unroll
  for (int j = 0; j < UPI; ++j) {
    u_row[j] = (j * NISS + iw) * 8 + l_row8;
    int u = user0 + u_row[j];
    if (u >= a.n_users) u = a.n_users - 1;  // clamp: padded user columns are never selected
    long long src = a.user_rows ? a.user_rows[u] : (long long)u;
    u_src[j] = a.users + src * a.user_stride + (l_slot ^ ((u_row[j] >> 1) & 7)) * 4;
  }

  const float* i_src[IPI];
  auto set_item_rows = [&](long long blk_local) {
    const long long pos0 = (a.blk_begin + blockIdx.x + blk_local * S) * IB;
    rot = a.rotate ? (int)((unsigned long long)((a.blk_begin + blockIdx.x + blk_local * S) * 5) % (unsigned)n_chunks) : 0;
    long long p[IPI];
#pragma unroll
    for (int j = 0; j < IPI; ++j) {
      p[j] = pos0 + i_row[j];
      if (p[j] >= a.n_cand) p[j] = a.n_cand - 1;  // clamp: rows past the end are masked at selection
    }
    if (WL) {
      // whitelist indirection, once per item block.  Loaded through inline asm with its own full wait:
      // a compiler-visible load inside the streaming loop makes hipcc place `s_waitcnt vmcnt(0)` in
      // front of the fragment reads of EVERY chunk (register-reuse hazard), draining the DMA ring.
      // (loads and their wait live in ONE asm statement: the results must not be touched before the s_waitcnt)
#pragma unroll
      for (int j0 = 0; j0 < IPI; j0 += 4) {
        long long s0, s1, s2, s3;
        asm volatile(
            "global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %5, off\n\t"
            "global_load_dwordx2 %2, %6, off\n\tglobal_load_dwordx2 %3, %7, off\n\ts_waitcnt vmcnt(0)"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
            : "v"(a.whitelist + p[j0]), "v"(a.whitelist + p[j0 + 1]), "v"(a.whitelist + p[j0 + 2]), "v"(a.whitelist + p[j0 + 3])
            : "memory");
        p[j0] = s0; p[j0 + 1] = s1; p[j0 + 2] = s2; p[j0 + 3] = s3;
      }
    }
#pragma unroll
    for (int j = 0; j < IPI; ++j) i_src[j] = a.items + p[j] * a.item_stride + i_colofs[j];
  };
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // shared-bound copy: issuers 0/2 bring users [0,64), issuers 1/3 users [64,128) of this tile (4 B per lane)
  int g_idx = user0 + (iw & 1) * 64 + lane;
  if (g_idx >= a.n_users_pad) g_idx = a.n_users_pad - 1;
  const unsigned* g_src = a.gthr + g_idx;
  auto issue = [&](int stage, int c) {  // DMA chunk c (already rotated) of the current issue block
    const unsigned sbase = smem_base + (unsigned)(stage * STAGE * 4);
    const int kofs = c * KC;
#pragma unroll
    for (int j = 0; j < IPI; ++j) dma16(i_src[j] + kofs, sbase + (unsigned)((iw * IPI + j) * 8 * KC * 4));
#pragma unroll
    for (int j = 0; j < UPI; ++j) dma16(u_src[j] + kofs, sbase + (unsigned)((SA + (j * NISS + iw) * 8 * KC) * 4));
    dma4(g_src, sbase + (unsigned)((SA + SU + (iw & 1) * 64) * 4));
  };

  // ---- prologue: NS-1 stages in flight ----
  long long iss_blk = 0; int iss_c = 0; long long issued = 0; int iss_stage = 0;
  if (issuer) set_item_rows(0);
  auto issue_next = [&]() {
    int cc = iss_c + rot; if (cc >= n_chunks) cc -= n_chunks;
    issue(iss_stage, cc);
    iss_stage = (iss_stage + 1 == NS) ? 0 : iss_stage + 1;
    ++issued; ++iss_c;
    if (iss_c == n_chunks) { iss_c = 0; ++iss_blk; if (iss_blk < my_blocks) set_item_rows(iss_blk); }
  };
  if (issuer) {
#pragma unroll 1
    for (int s = 0; s < NS - 1; ++s) if (issued < T) issue_next();
  }
  if (NLD && !computes) {   // loader: keep the ring full, one barrier per chunk in step with the compute waves
#pragma unroll 1
    for (long long g = 0; g < T; ++g) {
      if (issued - g == NS - 1) wait_vmcnt<NL*(NS - 2)>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // chunk g is in LDS; the compute waves are done with chunk g-1
      asm volatile("" ::: "memory");
      if (issued < T) issue_next();
    }
    return;
  }

  f32x16 acc[TU];
  float nrm_i = 0.f; float nrm_u[TU];
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    nrm_u[tu] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tu][r] = 0.f;
  }

  // fragment read offsets (swizzled)
  const int a_row = cwave * 32 + col;
  const int a_swz = (a_row >> 1) & 7;
  int u_swz[TU];
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) u_swz[tu] = ((tu * 32 + col) >> 1) & 7;

  int cons_stage = 0; long long g = 0;
#pragma unroll 1
  for (long long blk_local = 0; blk_local < my_blocks; ++blk_local) {
    const unsigned* g_lds = nullptr;
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c, ++g) {
      // chunk g landed?  outstanding stages allowed: NS-2 in steady state, 0 in the drain
      if (NLD == 0) {
        if (issued - g == NS - 1) wait_vmcnt<NL*(NS - 2)>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (NLD == 0 && issued < T) issue_next();  // refill the buffer consumed at step g-1

      const float* sbase = smem + cons_stage * STAGE;
      cons_stage = (cons_stage + 1 == NS) ? 0 : cons_stage + 1;
      g_lds = reinterpret_cast<const unsigned*>(sbase + SA + SU);
      const float* Ab = sbase + a_row * KC;
      const float* Ub = sbase + SA + col * KC;
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        f32x4 av = *reinterpret_cast<const f32x4*>(Ab + (((2 * s + half) ^ a_swz) << 2));
        bf16x8 a_hm, a_mh;
        if constexpr (HM) {
          const u32x4 w = __builtin_bit_cast(u32x4, av);
          u32x4 x;
#pragma unroll
          for (int t = 0; t < 4; ++t) x[t] = __builtin_amdgcn_alignbit(w[t], w[t], 16);   // (h, m) -> (m, h)
          a_hm = __builtin_bit_cast(bf16x8, w); a_mh = __builtin_bit_cast(bf16x8, x);
        } else {
          nrm_i = sumsq4(nrm_i, av);
        }
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) {
          f32x4 bv = *reinterpret_cast<const f32x4*>(Ub + tu * 32 * KC + (((2 * s + half) ^ u_swz[tu]) << 2));
          if constexpr (HM) {
            // both operands take their four (m, h) pairs from the same 16-byte slot, so the k positions pair up whatever the
            // instruction's own numbering of them is (the sum over k is order-free)
            const bf16x8 b8 = __builtin_bit_cast(bf16x8, bv);
            if (!a.h_only) acc[tu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mh, b8, acc[tu], 0, 0, 0);   // h m' + m h'
            acc[tu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hm, b8, acc[tu], 0, 0, 0);   // h h' + m m'  (h-only image: eight k per slot)
          } else {
            nrm_u[tu] = sumsq4(nrm_u[tu], bv);
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[tu] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[tu], 0, 0, 0);
          }
        }
      }
    }
    const long long pos0 = (a.blk_begin + blockIdx.x + blk_local * S) * IB;
    if (!(a.debug & 1)) select_block<TU, LL, true>(a, st, acc, nrm_i, nrm_u, pos0, list_id, user0, lane, cwave, g_lds);
    else if (acc[0][0] + nrm_i == 1.2345e30f) st.cnt[0] = 1;
    nrm_i = 0.f;
#pragma unroll
    for (int tu = 0; tu < TU; ++tu) {
      nrm_u[tu] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tu][r] = 0.f;
    }
  }
  publish_counts<TU, LL>(a, st, list_id, user0, lane, LL);
}

// ------------------------------------------------------------------------------------------------
// Merge: one workgroup per user.
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
  const float* list_scores; const int* list_pos; const int* list_counts;
  int n_lists; int n_users_pad; int k; int n_users;
  float* compact_scores; int* compact_pos; long long compact_cap;  // per-user scratch
  const long long* whitelist; long long id_offset; int distance;
  long long* out_ids; float* out_scores; int* out_counts;
  int out_k;          // entries extracted per user (0 = k).  The two-stage coarse pass takes k_cand >= k of them ...
  int* out_pos;       // ... as candidate POSITIONS [n_users][out_k] (the exact pass needs them) instead of ids
};

// Block-wide radix select: the ordered key (f32_to_key) of the `need`-th largest of the scores `each(f)` enumerates (f(score) once per entry
// and thread), need >= 1 and <= their count.  Three passes over the entries (11 + 11 + 10 bits, most significant first): a 2048-bin
// histogram in LDS, the bin that holds the need-th largest found by one thread walking it from the top.  The selection kernels walked the
// candidates once per RESULT before (k or 64 block-wide rounds: 240 us for 16 users x 24 k candidates); this is three walks.
template <int NT, typename Each>
__device__ __forceinline__ unsigned block_kth_largest_key(int need, Each each, unsigned* hist /* [2048 + 64] LDS */, unsigned* s_bcast /* [2] LDS */) {
  unsigned prefix = 0u;      // the bits fixed so far (high part of the key)
  int fixed = 0;             // how many
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int bits = pass < 2 ? 11 : 10, shift = 32 - fixed - bits;
    for (int i = threadIdx.x; i < 2048; i += NT) hist[i] = 0u;
    __syncthreads();
    each([&](float sc) {
      const unsigned key = f32_to_key(sc);
      if (fixed == 0 || (key >> (32 - fixed)) == prefix) atomicAdd(&hist[(key >> shift) & ((1u << bits) - 1u)], 1u);
    });
    __syncthreads();
    // the bin of the need-th largest, walking from the top: 64 threads sum 32 bins each, one thread walks the 64 sums and then the 32
    // bins of the group it stops in (a single thread walking 2048 bins paid 2048 dependent LDS round trips per pass)
    unsigned* sup = hist + 2048;                            // [64] group sums (the caller's array has room: 2048 + 64)
    if (threadIdx.x < 64) {
      unsigned t = 0u;
      for (int i = 0; i < 32; ++i) t += hist[threadIdx.x * 32 + i];
      sup[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int cum = 0, g = ((1 << bits) >> 5) - 1;
      for (; g > 0; --g) { if (cum + (int)sup[g] >= need) break; cum += (int)sup[g]; }
      int b = g * 32 + 31;
      for (; b > g * 32; --b) { if (cum + (int)hist[b] >= need) break; cum += (int)hist[b]; }
      s_bcast[0] = (unsigned)b; s_bcast[1] = (unsigned)(need - cum);
    }
    __syncthreads();
    prefix = (prefix << bits) | s_bcast[0]; need = (int)s_bcast[1]; fixed += bits;
    __syncthreads();
  }
  return prefix;
}

template <int NT_MERGE>
__global__ __launch_bounds__(NT_MERGE) void topk_merge_kernel(MergeArgs m) {
  // one workgroup per user
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  __shared__ int s_scan[NT_MERGE];
  __shared__ float s_ws[NT_MERGE / 64]; __shared__ long long s_wp[NT_MERGE / 64];
  __shared__ float s_bs; __shared__ long long s_bp;

  // pass 1: per-thread candidate counts -> exclusive scan -> compaction
  int my = 0;
  for (int l = tid; l < m.n_lists; l += NT_MERGE) my += m.list_counts[(long long)l * m.n_users_pad + u];
  s_scan[tid] = my;
  __syncthreads();
  for (int o = 1; o < NT_MERGE; o <<= 1) {
    int v = (tid >= o) ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const int total = s_scan[NT_MERGE - 1];
  int ofs = s_scan[tid] - my;
  float* cs = m.compact_scores + (long long)u * m.compact_cap;
  int* cp = m.compact_pos + (long long)u * m.compact_cap;
  for (int l = tid; l < m.n_lists; l += NT_MERGE) {
    const int c = m.list_counts[(long long)l * m.n_users_pad + u];
    const long long lb = ((long long)l * m.n_users_pad + u) * m.k;
    for (int e = 0; e < c; ++e) {
      cs[ofs] = m.list_scores[lb + e];
      cp[ofs] = m.list_pos[lb + e];
      ++ofs;
    }
  }
  __syncthreads();

  const int out_k = m.out_k > 0 ? m.out_k : m.k;
  const int n_out = total < out_k ? total : out_k;
  // few entries (recommend(): thousands of users, a handful of lists each): rank sort — every thread counts the entries ahead of its
  // own, one pass, no block-wide rounds
  constexpr int RANK_SORT_MAX = 1024;
  if (total <= RANK_SORT_MAX) {
    __shared__ float r_s[RANK_SORT_MAX]; __shared__ int r_p[RANK_SORT_MAX];
    for (int e = tid; e < total; e += NT_MERGE) { r_s[e] = cs[e]; r_p[e] = cp[e]; }
    __syncthreads();
    for (int e = tid; e < total; e += NT_MERGE) {
      const float se = r_s[e]; const long long pe = r_p[e];
      int rank = 0;
      for (int o = 0; o < total; ++o) rank += better(r_s[o], (long long)r_p[o], se, pe) ? 1 : 0;
      if (rank < n_out) {
        if (m.out_pos != nullptr) m.out_pos[(long long)u * out_k + rank] = (int)pe;
        else m.out_ids[(long long)u * out_k + rank] = m.whitelist ? m.whitelist[pe] : pe + m.id_offset;
        m.out_scores[(long long)u * out_k + rank] = (m.distance == DIST_EUCLID) ? -se : se;
      }
    }
    if (tid == 0) m.out_counts[u] = n_out;
    return;
  }
  {
    // many entries (a few users against a long catalog: thousands of lists): radix-select the n_out-th largest score, gather the entries
    // at or above it (n_out plus the ties of the last place) and rank-sort those
    __shared__ unsigned r_hist[2048 + 64]; __shared__ unsigned r_bc[2];
    __shared__ float g_s[RANK_SORT_MAX]; __shared__ int g_p[RANK_SORT_MAX]; __shared__ int g_n;
    const unsigned kth = block_kth_largest_key<NT_MERGE>(n_out, [&](auto f) { for (int e = tid; e < total; e += NT_MERGE) f(cs[e]); }, r_hist, r_bc);
    if (tid == 0) g_n = 0;
    __syncthreads();
    for (int e = tid; e < total; e += NT_MERGE) {
      const float sc = cs[e];
      if (f32_to_key(sc) >= kth) {
        const int at = atomicAdd(&g_n, 1);
        if (at < RANK_SORT_MAX) { g_s[at] = sc; g_p[at] = cp[e]; }
      }
    }
    __syncthreads();
    const int gn = g_n;
    if (gn <= RANK_SORT_MAX) {      // (more ties than that at the last place: the rounds below)
      for (int e = tid; e < gn; e += NT_MERGE) {
        const float se = g_s[e]; const long long pe = g_p[e];
        int rank = 0;
        for (int o = 0; o < gn; ++o) rank += better(g_s[o], (long long)g_p[o], se, pe) ? 1 : 0;
        if (rank < n_out) {
          if (m.out_pos != nullptr) m.out_pos[(long long)u * out_k + rank] = (int)pe;
          else m.out_ids[(long long)u * out_k + rank] = m.whitelist ? m.whitelist[pe] : pe + m.id_offset;
          m.out_scores[(long long)u * out_k + rank] = (m.distance == DIST_EUCLID) ? -se : se;
        }
      }
      if (tid == 0) m.out_counts[u] = n_out;
      return;
    }
  }
  // pass 2: k rounds of block-wide arg-best strictly below the previous winner
  float prev_s = INFINITY; long long prev_p = -1;
  for (int r = 0; r < n_out; ++r) {
    float bs = -INFINITY; long long bp = 0x7fffffffffffffffLL;
    for (int e = tid; e < total; e += NT_MERGE) {
      float sc = cs[e]; long long p = cp[e];
      bool below = (r == 0) || better(prev_s, prev_p, sc, p);
      if (below && better(sc, p, bs, bp)) { bs = sc; bp = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float os = __shfl_xor(bs, o, 64); long long op = __shfl_xor(bp, o, 64);
      if (better(os, op, bs, bp)) { bs = os; bp = op; }
    }
    if (lane == 0) { s_ws[wave] = bs; s_wp[wave] = bp; }
    __syncthreads();
    if (tid == 0) {
      float fs = s_ws[0]; long long fp = s_wp[0];
      for (int w = 1; w < NT_MERGE / 64; ++w)
        if (better(s_ws[w], s_wp[w], fs, fp)) { fs = s_ws[w]; fp = s_wp[w]; }
      s_bs = fs; s_bp = fp;
      if (m.out_pos != nullptr) m.out_pos[(long long)u * out_k + r] = (int)fp;
      else m.out_ids[(long long)u * out_k + r] = m.whitelist ? m.whitelist[fp] : fp + m.id_offset;
      m.out_scores[(long long)u * out_k + r] = (m.distance == DIST_EUCLID) ? -fs : fs;
    }
    __syncthreads();
    prev_s = s_bs; prev_p = s_bp;
  }
  if (tid == 0) m.out_counts[u] = n_out;
}

// Seed of the shared bound: k-th best score among the entries of the first `n_lists` lists of each user
// (one workgroup per user).  Any k real candidates give a valid lower bound of the final k-th best.
template <int NT_MERGE>
__global__ __launch_bounds__(NT_MERGE) void topk_seed_kernel(MergeArgs m, unsigned* gthr) {
  const int u = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ int s_total;
  int my = 0;
  for (int l = tid; l < m.n_lists; l += NT_MERGE) my += m.list_counts[(long long)l * m.n_users_pad + u];
  if (tid == 0) s_total = 0;
  __syncthreads();
  atomicAdd(&s_total, my);
  __syncthreads();
  if (s_total < m.k) return;
  // the k-th largest score of the prefix lists by radix select (three walks over the entries instead of k)
  __shared__ unsigned r_hist[2048 + 64]; __shared__ unsigned r_bc[2];
  const unsigned kth = block_kth_largest_key<NT_MERGE>(m.k, [&](auto f) {
    for (int l = tid; l < m.n_lists; l += NT_MERGE) {
      const int c = m.list_counts[(long long)l * m.n_users_pad + u];
      const long long lb = ((long long)l * m.n_users_pad + u) * m.k;
      for (int e = 0; e < c; ++e) f(m.list_scores[lb + e]);
    }
  }, r_hist, r_bc);
  if (tid == 0) atomicMax(gthr + u, kth);
}

// ------------------------------------------------------------------------------------------------
// Engine 3: the 16-USER tile — the HBM-bound regime of the exact fp32 formulation.
//
// With 32 users per catalog pass the LDS-DMA ring (1.66 ms for 10.24 GB) and the exact-fp32 MFMA work (1.8 ms) are equally
// long and contend (2.6-2.8 ms together).  Scoring 16 users per pass with v_mfma_f32_16x16x4_f32 halves the matrix work
// per byte streamed (AI = 8 flop/B, half the fp32 ridge), so the pass is bound by HBM alone.  Same ring, same swizzle,
// same dedicated loader waves as engine 2; what changes:
//   * MFMA roles: A = 16 items x 4 k, B = 4 k x 16 users; lane l feeds A[item l&15][k] and B[k][user l&15] for
//     k = 16s + 4(l>>4) + t, t = 0..3 — again ONE ds_read_b128 per operand and lane for 4 MFMA steps (conflict-free under the
//     ring's XOR swizzle for the b128 lane groups of gfx950).  A compute wave owns 32 item rows = 2 accumulators.
//   * D layout: lane l holds user l&15, item rows 4(l>>4) + r of each 16-row sub-tile: 8 scores of ONE user per lane, so
//     selection stays lane-local; a workgroup keeps 16 lists per user (4 waves x 4 lane groups) in LDS.
//   * the user tile is 2 KiB per stage instead of 4, which buys a 7-stage ring next to the lists (one workgroup per CU).
//   * each workgroup MERGES its 16 lists per user into one k-entry list before it leaves (LDS, wave-level argmax rounds):
//     the seed / final selection kernels then see S lists per user instead of 16 S, and run in microseconds
//     (topk_select_kernel) — at a 1.8 ms pass the old 80 + 65 us seed and merge kernels were 8 % of the call.
//   * no `resume`: the seeding prefix (2 item blocks per workgroup, all workgroups) keeps its own merged lists, which
//     simply take part in the final selection next to the main pass's lists.
// Scores are the same k-ordered fmaf chains up to association: (k, k+4, k+8, k+12) per MFMA here, (k, k+4) in engine 2.
// ------------------------------------------------------------------------------------------------
constexpr int UB16 = 16;          // users per tile
constexpr int LISTS16 = 16;       // lists per workgroup and user: 4 compute waves x 4 lane groups
constexpr int SG16 = 128;         // shared-bound copy (uints) per stage, two 64-lane dword pieces
constexpr int STAGE16 = IB * KC + UB16 * KC + SG16;   // floats per ring stage

inline size_t stream16_lds_bytes(int ns, int k) {
  const int kp = (k + 3) & ~3;
  return (size_t)ns * STAGE16 * sizeof(float) + (size_t)LISTS16 * UB16 * kp * 8 + (size_t)LISTS16 * UB16 * 4;
}

struct Sel16 {
  typedef ListTypes<true>::fptr fptr;
  typedef ListTypes<true>::iptr iptr;
  float worst_s; long long worst_p; int worst_slot; int cnt; float thr; float g_seen;
  fptr ls; iptr lp;
  long long fbase; int flg;
};

__device__ __forceinline__ void select_block16(const TopkArgs& a, Sel16& st, const f32x4 (&acc)[2], const float (&nrm_i)[2],
                                               float nrm_u, long long pos0, int user0, int lane, int wave,
                                               const unsigned* g_lds) {
  const int col = lane & 15, grp = lane >> 4;
  const bool need_norm = a.distance != DIST_DOT;
  float ni_full[2] = {0.f, 0.f}, nu_full = 0.f, inv_u = 1.f;
  if (need_norm) {   // lanes c, c+16, c+32, c+48 hold the four k-quarters of row / user c
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float v = nrm_i[it];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      ni_full[it] = v;
    }
    float v = nrm_u;
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    nu_full = v;
    inv_u = 1.0f / fmaxf(sqrtf(nu_full), 1e-8f);
  }
  const int u = user0 + col;
  const bool uvalid = u < a.n_users;
  if (uvalid && a.use_bound) {  // refresh from the shared per-user bound (LDS copy brought in by the DMA ring)
    const float g = key_to_f32(g_lds[col]);
    st.g_seen = fmaxf(st.g_seen, g);
    st.thr = fmaxf(st.thr, g);
  }
  float sc[8];
  unsigned cmask = 0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = it * 16 + 4 * grp + r;
      float s = acc[it][r];
      if (need_norm) {
        const float ni = __shfl(ni_full[it], 4 * grp + r, 64);   // lane (row & 15) holds that row's norm
        if (a.distance == DIST_COSINE) s = s * inv_u * (1.0f / fmaxf(sqrtf(ni), 1e-8f));
        else s = -sqrtf(fmaxf(nu_full + ni - 2.0f * s, 0.f));
      }
      sc[it * 4 + r] = s;
      const long long p = pos0 + wave * 32 + row;
      if (uvalid && p < a.n_cand && s >= st.thr) cmask |= (1u << (it * 4 + r));
    }
  }
  if (__any(cmask != 0)) {
    const Sel16::fptr lsc = st.ls; const Sel16::iptr lps = st.lp;
    while (cmask != 0) {
      const int q = __ffs(cmask) - 1;
      cmask &= cmask - 1;
      float s = sc[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) s = (q == i) ? sc[i] : s;  // static-index select: no scratch
      const int row = (q >> 2) * 16 + 4 * grp + (q & 3);
      const long long p = pos0 + wave * 32 + row;
      if (!(s >= st.thr)) continue;  // threshold may have risen inside this loop
      const bool take = (st.cnt < a.k) || better(s, p, st.worst_s, st.worst_p);
      if (!take) continue;
      const long long cid = a.whitelist ? a.whitelist[p] : p + a.id_offset;
      if (a.filt_hash != nullptr) {
        bool hit = false;
        if (st.flg >= 0) {
          const int* tab = a.filt_hash + st.fbase;
          const unsigned mask = (1u << st.flg) - 1u;
          unsigned h = filt_hash_of((unsigned)cid, st.flg);
          for (;;) {
            const int v = tab[h];
            if (v == (int)cid) { hit = true; break; }
            if (v < 0) break;
            h = (h + 1) & mask;
          }
        }
        if (hit) continue;
      } else if (is_filtered(a, u, cid)) continue;
      if (st.cnt < a.k) {
        lsc[st.cnt] = s; lps[st.cnt] = (int)p; st.cnt += 1;
      } else {
        lsc[st.worst_slot] = s; lps[st.worst_slot] = (int)p;
      }
      if (st.cnt == a.k) {  // list full: (re)locate its worst entry (whole list in flight, then a register scan)
        typedef __attribute__((address_space(3))) f32x4* fptr4;
        typedef __attribute__((address_space(3))) i32x4* iptr4;
        const int kp = (a.k + 3) & ~3;
        f32x4 sv[K_LDS_LISTS / 4]; i32x4 pv[K_LDS_LISTS / 4];
#pragma unroll
        for (int j = 0; j < K_LDS_LISTS / 4; ++j)
          if (4 * j < kp) { sv[j] = *((fptr4)lsc + j); pv[j] = *((iptr4)lps + j); }
        float ws = INFINITY; long long wp = -1; int wslot = 0;
#pragma unroll
        for (int j = 0; j < K_LDS_LISTS / 4; ++j)
          if (4 * j < kp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float es = sv[j][i]; const long long ep = pv[j][i];
              if (4 * j + i < a.k && ((4 * j + i == 0) || better(ws, wp, es, ep))) { ws = es; wp = ep; wslot = 4 * j + i; }
            }
          }
        st.worst_s = ws; st.worst_p = wp; st.worst_slot = wslot;
        st.thr = fmaxf(st.thr, ws);
      }
    }
    // publish at most once per block and list.  NOT in the seeding prefix: there every list fills up in its second block,
    // and 4096 lists x 16 users of atomics on ONE 64-byte line serialise in L2 (200 us for a pass that streams in 30)
    if (a.use_bound && st.cnt == a.k && st.worst_s > st.g_seen) {
      atomicMax(a.gthr + u, f32_to_key(st.worst_s));
      st.g_seen = st.worst_s;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // leave the slow path with an empty VMEM scoreboard (see select_block)
  }
}

template <int NS, bool WL>
__global__ __launch_bounds__(NTHREADS + 128) void topk_stream16_kernel(TopkArgs a) {
  constexpr int IPI = 8;                 // item pieces per loader and stage (16 pieces of 8 rows)
  constexpr int SA = IB * KC, SU = UB16 * KC;
  constexpr int NL = IPI + 1 + 1;        // LDS-DMA instructions per loader and stage: items, one user piece, the bound copy
  static_assert(NL * (NS - 2) < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NS][STAGE16] | list scores | list positions | counts

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, grp = lane >> 4;
  const int S = a.n_seg;
  const int user0 = blockIdx.y * UB16;
  const int kp = (a.k + 3) & ~3;

  const long long n_blocks = a.blk_end - a.blk_begin;
  const int n_chunks = a.d / KC;
  const long long my_blocks = ((int)blockIdx.x < S && n_blocks > blockIdx.x) ? (n_blocks - blockIdx.x + S - 1) / S : 0;
  const long long T = my_blocks * n_chunks;

  const bool issuer = wave >= 4;
  const int iw = wave - 4;
  const int cwave = issuer ? 0 : wave;
  float* const lists_s = smem + NS * STAGE16;                            // [16 lists][16 users][kp]
  int* const lists_p = reinterpret_cast<int*>(lists_s + LISTS16 * UB16 * kp);
  int* const lists_c = lists_p + LISTS16 * UB16 * kp;                    // [16 lists][16 users]
  const int out_list = a.list_base + blockIdx.x;   // merged list (user u, out_list) lives at [u * n_lists_total + out_list]

  if (T == 0) {   // no item block for this workgroup in this launch: its merged lists are empty
    if (tid < UB16) a.list_counts[(long long)(user0 + tid) * a.n_lists_total + out_list] = 0;
    return;
  }

  if (issuer) {
    // ---- loader waves: keep the ring full, one barrier per chunk in step with the compute waves ----
    const int l_row8 = lane >> 3, l_slot = lane & 7;
    int i_row[IPI]; int i_colofs[IPI];
#pragma unroll
    for (int j = 0; j < IPI; ++j) {
      i_row[j] = (iw * IPI + j) * 8 + l_row8;
      i_colofs[j] = (l_slot ^ ((i_row[j] >> 1) & 7)) * 4;
    }
    const int u_row = iw * 8 + l_row8;
    int uu = user0 + u_row;
    if (uu >= a.n_users) uu = a.n_users - 1;    // clamp: padded user columns are never selected
    const long long usrc = a.user_rows ? a.user_rows[uu] : (long long)uu;
    const float* u_src = a.users + usrc * a.user_stride + (l_slot ^ ((u_row >> 1) & 7)) * 4;
    int g_idx = user0 + lane;
    if (g_idx >= a.n_users_pad) g_idx = a.n_users_pad - 1;
    const unsigned* g_src = a.gthr + g_idx;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));

    const float* i_src[IPI];
    int rot = 0;
    auto set_item_rows = [&](long long blk_local) {
      const long long blk = a.blk_begin + blockIdx.x + blk_local * S;
      rot = a.rotate ? (int)((unsigned long long)(blk * 5) % (unsigned)n_chunks) : 0;
      long long p[IPI];
#pragma unroll
      for (int j = 0; j < IPI; ++j) {
        p[j] = blk * IB + i_row[j];
        if (p[j] >= a.n_cand) p[j] = a.n_cand - 1;  // clamp: rows past the end are masked at selection
      }
      if (WL) {   // whitelist indirection through inline asm with its own wait (see topk_stream_kernel)
#pragma unroll
        for (int j0 = 0; j0 < IPI; j0 += 4) {
          long long s0, s1, s2, s3;
          asm volatile(
              "global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %5, off\n\t"
              "global_load_dwordx2 %2, %6, off\n\tglobal_load_dwordx2 %3, %7, off\n\ts_waitcnt vmcnt(0)"
              : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
              : "v"(a.whitelist + p[j0]), "v"(a.whitelist + p[j0 + 1]), "v"(a.whitelist + p[j0 + 2]), "v"(a.whitelist + p[j0 + 3])
              : "memory");
          p[j0] = s0; p[j0 + 1] = s1; p[j0 + 2] = s2; p[j0 + 3] = s3;
        }
      }
#pragma unroll
      for (int j = 0; j < IPI; ++j) i_src[j] = a.items + p[j] * a.item_stride + i_colofs[j];
    };
    long long iss_blk = 0, issued = 0; int iss_c = 0, iss_stage = 0;
    set_item_rows(0);
    auto issue_next = [&]() {
      int cc = iss_c + rot; if (cc >= n_chunks) cc -= n_chunks;
      const unsigned sbase = smem_base + (unsigned)(iss_stage * STAGE16 * 4);
      const int kofs = cc * KC;
#pragma unroll
      for (int j = 0; j < IPI; ++j) dma16(i_src[j] + kofs, sbase + (unsigned)((iw * IPI + j) * 8 * KC * 4));
      dma16(u_src + kofs, sbase + (unsigned)((SA + iw * 8 * KC) * 4));
      dma4(g_src, sbase + (unsigned)((SA + SU + iw * 64) * 4));
      iss_stage = (iss_stage + 1 == NS) ? 0 : iss_stage + 1;
      ++issued; ++iss_c;
      if (iss_c == n_chunks) { iss_c = 0; ++iss_blk; if (iss_blk < my_blocks) set_item_rows(iss_blk); }
    };
#pragma unroll 1
    for (int s = 0; s < NS - 1; ++s) if (issued < T) issue_next();
#pragma unroll 1
    for (long long g = 0; g < T; ++g) {
      if (issued - g == NS - 1) wait_vmcnt<NL*(NS - 2)>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // chunk g is in LDS; the compute waves are done with chunk g-1
      asm volatile("" ::: "memory");
      if (issued < T) issue_next();
    }
  } else {
    // ---- compute waves ----
    Sel16 st;
    st.worst_s = -INFINITY; st.worst_p = -1; st.worst_slot = 0; st.cnt = 0; st.thr = -INFINITY; st.g_seen = -INFINITY;
    {
      const int L = ((cwave * 4 + grp) * UB16 + col) * kp;
      st.ls = (Sel16::fptr)lists_s + L;
      st.lp = (Sel16::iptr)lists_p + L;
      for (int e = a.k; e < kp; ++e) { st.ls[e] = INFINITY; st.lp[e] = -1; }
      st.fbase = 0; st.flg = -1;
      const int u = user0 + col;
      if (a.filt_hash != nullptr && u < a.n_users) {
        const long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
        if (hi > lo) { st.fbase = 4 * lo + 4 * (long long)(u + a.filt_u0); st.flg = filt_hash_lg(hi - lo); }
      }
    }
    f32x4 acc[2];
    float nrm_i[2] = {0.f, 0.f}; float nrm_u = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[it][r] = 0.f;
    // fragment read offsets (swizzled): item rows cwave*32 + it*16 + col, user row col; 16-byte slot 4s + grp
    int a_off[2], a_swz[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = cwave * 32 + it * 16 + col;
      a_off[it] = row * KC; a_swz[it] = (row >> 1) & 7;
    }
    const int u_off = SA + col * KC, u_swz = (col >> 1) & 7;
    const bool need_norm = a.distance != DIST_DOT;

    int cons_stage = 0;
#pragma unroll 1
    for (long long blk_local = 0; blk_local < my_blocks; ++blk_local) {
      const unsigned* g_lds = nullptr;
#pragma unroll 1
      for (int c = 0; c < n_chunks; ++c) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* sbase = smem + cons_stage * STAGE16;
        cons_stage = (cons_stage + 1 == NS) ? 0 : cons_stage + 1;
        g_lds = reinterpret_cast<const unsigned*>(sbase + SA + SU);
#pragma unroll
        for (int s = 0; s < KC / 16; ++s) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sbase + u_off + (((4 * s + grp) ^ u_swz) << 2));
          const f32x4 av0 = *reinterpret_cast<const f32x4*>(sbase + a_off[0] + (((4 * s + grp) ^ a_swz[0]) << 2));
          const f32x4 av1 = *reinterpret_cast<const f32x4*>(sbase + a_off[1] + (((4 * s + grp) ^ a_swz[1]) << 2));
          if (need_norm) {   // wave-uniform: dot products skip the row-norm arithmetic
            nrm_u += bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2] + bv[3] * bv[3];
            nrm_i[0] += av0[0] * av0[0] + av0[1] * av0[1] + av0[2] * av0[2] + av0[3] * av0[3];
            nrm_i[1] += av1[0] * av1[0] + av1[1] * av1[1] + av1[2] * av1[2] + av1[3] * av1[3];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {   // two independent accumulators alternate: the 40-cycle dependent latency is hidden
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[t], bv[t], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[t], bv[t], acc[1], 0, 0, 0);
          }
        }
      }
      const long long pos0 = (a.blk_begin + blockIdx.x + blk_local * S) * IB;
      select_block16(a, st, acc, nrm_i, nrm_u, pos0, user0, lane, cwave, g_lds);
      nrm_i[0] = nrm_i[1] = 0.f; nrm_u = 0.f;
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[it][r] = 0.f;
    }
    lists_c[(cwave * 4 + grp) * UB16 + col] = st.cnt;
  }

  // ---- workgroup-level merge: 16 lists per user -> ONE list of <= k entries, best first, into the global list array ----
  __syncthreads();
  if (issuer) return;
  for (int uu = 0; uu < 4; ++uu) {
    const int ul = cwave * 4 + uu;               // user handled by this wave in this round
    const int u = user0 + ul;
    if (u >= a.n_users_pad) break;
    const int list = lane >> 2, q4 = lane & 3;  // lane -> (list, quarter of its kp entries)
    const int cnt = lists_c[list * UB16 + ul];
    const int epl = kp >> 2;                    // entries per lane (kp <= 16 -> <= 4)
    float es[4]; long long ep[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = q4 * epl + j;
      const bool ok = j < epl && e < cnt;
      es[j] = ok ? lists_s[(list * UB16 + ul) * kp + e] : -INFINITY;
      ep[j] = ok ? (long long)lists_p[(list * UB16 + ul) * kp + e] : 0x7fffffffffffffffLL;
    }
    int n_out = 0;
    for (int r = 0; r < a.k; ++r) {
      float bs = -INFINITY; long long bp = 0x7fffffffffffffffLL;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (better(es[j], ep[j], bs, bp)) { bs = es[j]; bp = ep[j]; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(bs, o, 64); const long long op = __shfl_xor(bp, o, 64);
        if (better(os, op, bs, bp)) { bs = os; bp = op; }
      }
      if (bp == 0x7fffffffffffffffLL) break;    // wave-uniform: nothing left
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ep[j] == bp) { es[j] = -INFINITY; ep[j] = 0x7fffffffffffffffLL; }   // positions are unique: exactly one entry leaves
      if (lane == 0 && u < a.n_users) {
        a.list_scores[((long long)u * a.n_lists_total + out_list) * a.k + r] = bs;
        a.list_pos[((long long)u * a.n_lists_total + out_list) * a.k + r] = (int)bp;
      }
      ++n_out;
    }
    if (lane == 0) a.list_counts[(long long)u * a.n_lists_total + out_list] = (u < a.n_users) ? n_out : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Selection over the merged per-workgroup lists (16-user path): one workgroup of 256 threads per user.  The user's lists
// ([list][k], each sorted best first by the workgroup merge) are copied into LDS with coalesced loads; a thread owns the
// HEADS of up to 4 lists, and each of the k rounds is one block-wide arg-best over the heads (wave shuffles + 4 partials),
// after which the winning list advances.  SEED: store the k-th best of the seeding prefix as the shared bound (or -inf);
// otherwise write the final (id, score) rows, best first, ties to the lower position.
// ------------------------------------------------------------------------------------------------
template <bool SEED>
__global__ __launch_bounds__(256) void topk_select_kernel(MergeArgs m, int first_list, int n_lists, int n_lists_total, unsigned* gthr) {
  extern __shared__ __attribute__((aligned(16))) float sel_smem[];   // scores [n_lists * k] | positions [n_lists * k]
  const int u = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float s_ws[2][4]; __shared__ long long s_wp[2][4];
  float* ss = sel_smem; int* sp = reinterpret_cast<int*>(sel_smem + (long long)n_lists * m.k);
  const long long base = ((long long)u * n_lists_total + first_list) * m.k;
  for (int i = tid; i < n_lists * m.k; i += 256) { ss[i] = m.list_scores[base + i]; sp[i] = m.list_pos[base + i]; }
  int cnt[4], head[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int l = tid + 256 * j;
    cnt[j] = l < n_lists ? m.list_counts[(long long)u * n_lists_total + first_list + l] : 0;
    head[j] = 0;
  }
  __syncthreads();
  int n_out = 0;
  float last_s = -INFINITY;
  for (int r = 0; r < m.k; ++r) {
    float bs = -INFINITY; long long bp = 0x7fffffffffffffffLL; int bj = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (head[j] < cnt[j]) {
        const int e = (tid + 256 * j) * m.k + head[j];
        const float es = ss[e]; const long long ep = sp[e];
        if (better(es, ep, bs, bp)) { bs = es; bp = ep; bj = j; }
      }
    const long long my_p = bp;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64); const long long op = __shfl_xor(bp, o, 64);
      if (better(os, op, bs, bp)) { bs = os; bp = op; }
    }
    if (lane == 0) { s_ws[r & 1][wave] = bs; s_wp[r & 1][wave] = bp; }
    __syncthreads();
    bs = s_ws[r & 1][0]; bp = s_wp[r & 1][0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (better(s_ws[r & 1][w], s_wp[r & 1][w], bs, bp)) { bs = s_ws[r & 1][w]; bp = s_wp[r & 1][w]; }
    if (bp == 0x7fffffffffffffffLL) break;   // block-uniform: fewer than k candidates
    if (my_p == bp) {                        // positions are unique: exactly one thread owns the winner
#pragma unroll
      for (int j = 0; j < 4; ++j) head[j] += (j == bj) ? 1 : 0;
    }
    if (!SEED && tid == 0) {
      m.out_ids[(long long)u * m.k + r] = m.whitelist ? m.whitelist[bp] : bp + m.id_offset;
      m.out_scores[(long long)u * m.k + r] = (m.distance == DIST_EUCLID) ? -bs : bs;
    }
    last_s = bs;
    ++n_out;
  }
  if (tid == 0) {
    if (SEED) gthr[u] = f32_to_key(n_out == m.k ? last_s : -INFINITY);   // plain store: nothing else touches the bound yet
    else m.out_counts[u] = n_out;
  }
}

__global__ void fill_u32_kernel(unsigned* p, unsigned v, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// Two-stage top-k, the exact pass: the candidates the coarse (hm) pass kept are scored again IN THE ARITHMETIC OF ENGINE 2 — the same
// v_mfma_f32_32x32x2_f32 chain over the same k order (chunk rotation of the candidate's item block, then (k, k + 4) pairs inside a chunk) —
// so a candidate's exact score is bit for bit what topk_stream_kernel computes for it, and the k best of them in (score desc, position asc)
// order are what rt_topk_score returns whenever the candidate set provably holds the exact top-k.
// One workgroup per user, one wave per 32 candidates.  Lane column j feeds row j of A with candidate j and column j of B with the user,
// BOTH walked in candidate j's chunk order: the diagonal of the 32 x 32 tile holds the 32 scores (every output element of the instruction
// depends on its own row and column only).
// Proof of completeness per user: every (user, item) pair the coarse pass dropped had a coarse score <= tau = max(shared bound at the end
// of the pass, worst coarse score among the candidates when the candidate set is full); |coarse - exact| <= eps = err_coef |u| max|v|
// (rt_topk_score_two_stage); so with e_k = the k-th best exact score among the candidates, e_k - eps > tau means nothing outside can
// reach e_k.  Users for whom that fails are flagged in out_unproven (their outputs are still the best k candidates, not proven complete).
// ------------------------------------------------------------------------------------------------
struct ReplayArgs {
  const float* users; long long user_stride; const long long* user_rows;
  const float* items; long long item_stride; const long long* whitelist; long long id_offset;
  int d, k, kc, rotate, cosine;
  const int* cand_pos; const float* cand_coarse; const int* cand_counts;   // [n_users][kc], best coarse first
  const unsigned* gthr; const float* user_norms; float max_item_norm, err_coef;
  long long* out_ids; float* out_scores; int* out_counts; int* out_unproven;
};

template <int NW>   // waves per user = kc / 32
__global__ __launch_bounds__(64 * NW) void topk_replay_kernel(ReplayArgs r) {
  const int u = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  __shared__ float s_sc[32 * NW]; __shared__ int s_pos[32 * NW];
  __shared__ float s_ek;
  const int cnt = r.cand_counts[u];
  const int ci = wave * 32 + col;
  const bool cv = ci < cnt;
  const int pos = cnt > 0 ? r.cand_pos[(long long)u * r.kc + (cv ? ci : 0)] : 0;        // idle columns repeat candidate 0
  const long long irow = r.whitelist ? r.whitelist[pos] : (long long)pos;
  const int n_chunks = r.d / KC;
  const int rot = r.rotate ? (int)((unsigned long long)((long long)(pos / IB) * 5) % (unsigned)n_chunks) : 0;
  const float* ip = r.items + irow * r.item_stride;
  const float* up = r.users + (r.user_rows ? r.user_rows[u] : (long long)u) * r.user_stride;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float nrm_i = 0.f, nrm_u = 0.f;      // cosine: the row norms as engine 2 accumulates them (this lane's half of the k positions, chunk order)
  if (wave * 32 < cnt) {
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
      int cc = c + rot; if (cc >= n_chunks) cc -= n_chunks;
      f32x4 av[KC / 8], bv[KC / 8];
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        av[s] = *reinterpret_cast<const f32x4*>(ip + cc * KC + (2 * s + half) * 4);
        bv[s] = *reinterpret_cast<const f32x4*>(up + cc * KC + (2 * s + half) * 4);
      }
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        nrm_i = sumsq4(nrm_i, av[s]); nrm_u = sumsq4(nrm_u, bv[s]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][t], bv[s][t], acc, 0, 0, 0);
      }
    }
  }
  const float ni_full = nrm_i + __shfl_xor(nrm_i, 32, 64), nu_full = nrm_u + __shfl_xor(nrm_u, 32, 64);   // as select_block
  // the diagonal: element (i, i) sits in the lane with column i and half (i >> 2) & 1, register (i & 3) + 4 (i >> 3)
  if (half == ((col >> 2) & 1)) {
    const int rr = (col & 3) + 4 * (col >> 3);
    float sc = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) sc = (rr == i) ? acc[i] : sc;
    if (r.cosine) {                       // select_block's expression, term for term
      const float inv_u = 1.0f / fmaxf(sqrtf(nu_full), 1e-8f);
      sc = sc * inv_u * (1.0f / fmaxf(sqrtf(ni_full), 1e-8f));
    }
    s_sc[ci] = cv ? sc : -INFINITY;
    s_pos[ci] = cv ? pos : 0x7fffffff;
  }
  __syncthreads();
  if (wave != 0) return;
  // rank sort of the <= 32 NW entries by (score desc, position asc): lane e counts the entries ahead of its own
  const int n_out = cnt < r.k ? cnt : r.k;
  float ek = -INFINITY;
  for (int e = lane; e < 32 * NW; e += 64) {
    const float se = s_sc[e]; const long long pe = s_pos[e];
    int rank = 0;
    for (int o = 0; o < 32 * NW; ++o) rank += better(s_sc[o], (long long)s_pos[o], se, pe) ? 1 : 0;
    if (e < cnt && rank < n_out) {
      r.out_ids[(long long)u * r.k + rank] = r.whitelist ? r.whitelist[pe] : pe + r.id_offset;
      r.out_scores[(long long)u * r.k + rank] = se;
      if (rank == r.k - 1) ek = se;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ek = fmaxf(ek, __shfl_xor(ek, o, 64));
  if (lane == 0) {
    r.out_counts[u] = n_out;
    float tau = key_to_f32(r.gthr[u]);
    if (cnt == r.kc) tau = fmaxf(tau, r.cand_coarse[(long long)u * r.kc + r.kc - 1]);
    // cosine: the coarse pass scored unit rows (|u| = |v| = 1 up to the 2 ulp of the normalisation, 2^-22 on the dot product)
    const float eps = r.cosine ? r.err_coef + 3e-7f : r.err_coef * r.user_norms[u] * r.max_item_norm;
    const bool proven = (tau == -INFINITY) || (cnt >= r.k && ek - eps > tau);
    r.out_unproven[u] = proven ? 0 : 1;
  }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

constexpr size_t LDS_PER_CU = 160 * 1024;

inline size_t stream_lds_bytes(int tu, int ns, int k_lds) {   // LDS lists use a stride of k rounded up to 4 entries
  return (size_t)ns * (IB * KC + 32 * tu * KC + 128) * sizeof(float) + (size_t)LISTS_PER_WG * 32 * tu * ((k_lds + 3) & ~3) * 8;
}

struct Plan {
  int tu, ub, S, n_tiles, n_users_pad, n_lists, users_per_launch;
  int ns, wg_per_cu; bool lds_lists;
  int S_seed; long long blocks_seed;  // phase A (threshold seeding) geometry; 0 = single phase
  size_t o_gthr, o_scores, o_pos, o_counts, o_cscores, o_cpos, total;
};

constexpr int MAX_USERS_PER_LAUNCH = 16384;

inline int pick_tu(int users_per_pass, int n_users) {
  int upp = users_per_pass;
  if (upp <= 0) upp = 64;  // fp32 MFMA vs HBM balance point on gfx950
  if (upp > 128) upp = 128;
  int tu = upp <= 32 ? 1 : (upp <= 64 ? 2 : 4);
  if (n_users <= 32) tu = 1; else if (n_users <= 64 && tu > 2) tu = 2;  // no empty user tiles
  return tu;
}

inline Plan make_plan(int n_users, long long n_cand, int k, int users_per_pass) {
  Plan P;
  P.tu = pick_tu(users_per_pass, n_users);
  P.ub = 32 * P.tu;
  P.users_per_launch = n_users < MAX_USERS_PER_LAUNCH ? n_users : MAX_USERS_PER_LAUNCH;
  P.n_tiles = (P.users_per_launch + P.ub - 1) / P.ub;
  P.n_users_pad = P.n_tiles * P.ub;
  const long long n_blocks = (n_cand + IB - 1) / IB;
  // LDS budget decides ring depth, list placement and residency (160 KiB per CU).  The ring must keep
  // ~HBM latency x per-CU bandwidth (~80-100 KB) in flight, so: ONE workgroup per CU with the deepest ring
  // that fits (<= 6 stages); lists go to LDS when they still fit next to >= 4 stages.
  P.lds_lists = (k <= K_LDS_LISTS);
  if (P.lds_lists && stream_lds_bytes(P.tu, 4, k) > LDS_PER_CU) P.lds_lists = false;
  const int kl = P.lds_lists ? k : 0;
  P.wg_per_cu = 1;
  P.ns = 3;
  for (int ns = 6; ns >= 3; --ns)
    if (stream_lds_bytes(P.tu, ns, kl) <= LDS_PER_CU) { P.ns = ns; break; }
  long long target = (long long)P.wg_per_cu * rt_num_cus();
  long long S = (target + P.n_tiles - 1) / P.n_tiles;
  if (S > n_blocks) S = n_blocks;
  if (S < 1) S = 1;
  P.S = (int)S;
  P.n_lists = P.S * LISTS_PER_WG;
  // Threshold seeding (phase A): the first workgroups score 2 blocks each of a catalog prefix; the k-th best of
  // that prefix (topk_seed_kernel) becomes the shared bound, so that in phase B a list sees ~1 candidate
  // instead of ~k*ln(blocks): the slow path (and its pipeline drain) becomes rare at wave level.
  P.S_seed = 0; P.blocks_seed = 0;
  {
    // 128 workgroups x 2 blocks (round 6, 5 M x 512, 16 users: 1.112 ms per call against 1.140 with 64 x 8, 1.184 with 256 x 2 — the
    // seed kernel's lists grow with the workgroups —, 1.336 without seeding; gpurun_out/r6_seed)
    const int ss = P.S < 128 ? P.S : 128;
    const long long bs = (long long)ss * 2;
    if (n_blocks >= 32 * bs && P.n_tiles <= 8) { P.S_seed = ss; P.blocks_seed = bs; }
  }
  size_t o = 0;
  P.o_gthr = o; o = align_up(o + (size_t)P.n_users_pad * 4, 256);
  const size_t ent = (size_t)P.n_lists * P.n_users_pad * (size_t)k;
  P.o_scores = o; o = align_up(o + ent * 4, 256);
  P.o_pos = o; o = align_up(o + ent * 4, 256);
  P.o_counts = o; o = align_up(o + (size_t)P.n_lists * P.n_users_pad * 4, 256);
  const size_t cap = (size_t)P.n_lists * k;
  P.o_cscores = o; o = align_up(o + (size_t)P.users_per_launch * cap * 4, 256);
  P.o_cpos = o; o = align_up(o + (size_t)P.users_per_launch * cap * 4, 256);
  P.total = o;
  return P;
}

template <int TU, int NS, bool WL, bool LL, int NLD, bool HM = false>
int launch_stream_nld(const TopkArgs& a_in, dim3 grid, hipStream_t stream) {
  TopkArgs a = a_in;
  size_t lds = stream_lds_bytes(TU, NS, LL ? a.k : 0);
  // per-user Bloom filters of the viewed items, if the filter comes with hash sets and 128 B per user still fit in LDS
  // (not for a phase of a few blocks per workgroup — the seeding prefix: building the filters costs every workgroup more than the handful of
  // exact probes they would save there; measured 168 vs 112 us for the 16-user prefix over 5 M x 512)
  const bool tiny_phase = (a.blk_end - a.blk_begin) <= 8LL * (a.n_seg > 0 ? a.n_seg : 1);
  a.bloom = (a.filt_hash != nullptr && !tiny_phase && lds + (size_t)32 * TU * 128 <= LDS_PER_CU) ? 1 : 0;
  if (a.bloom) lds += (size_t)32 * TU * 128;
  static size_t attr_lds = 0;
  if (lds > 64 * 1024 && lds > attr_lds) {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_stream_kernel<TU, NS, WL, LL, NLD, HM>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  topk_stream_kernel<TU, NS, WL, LL, NLD, HM><<<grid, NTHREADS + NLD * 64, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
// Loader waves: two dedicated loader waves for the small user tiles (0 = every wave issues and computes)
// that are HBM-bound; the 128-user tile is MFMA-bound and keeps its issue slots for compute waves only)
template <int TU, int NS, bool WL, bool LL, bool HM>
int launch_stream_impl(const TopkArgs& a, dim3 grid, hipStream_t stream) {
  // the coarse pass spends a quarter of the matrix time per chunk: the DMA issue slots weigh as much as the MFMAs even at 128 users
  static const int loaders = HM ? (2) : (TU <= 2 ? 2 : 0);
  if constexpr (TU <= 2 || (HM && (8 + 2 * TU + 1) * (NS - 2) < 64)) {   // vmcnt is a 6-bit counter: (8 + 2 TU + 1) pieces x (NS - 2) stages must stay below 64
    if (loaders == 2) return launch_stream_nld<TU, NS, WL, LL, 2, HM>(a, grid, stream);
  }
  return launch_stream_nld<TU, NS, WL, LL, 0, HM>(a, grid, stream);
}
template <int TU, int NS, bool HM>
int launch_stream(const TopkArgs& a, dim3 grid, bool lds_lists, hipStream_t stream) {
  if (a.whitelist) {
    return lds_lists ? launch_stream_impl<TU, NS, true, true, HM>(a, grid, stream)
                     : launch_stream_impl<TU, NS, true, false, HM>(a, grid, stream);
  }
  return lds_lists ? launch_stream_impl<TU, NS, false, true, HM>(a, grid, stream)
                   : launch_stream_impl<TU, NS, false, false, HM>(a, grid, stream);
}

template <int TU, bool HM>
int launch_stream_ns(int ns, const TopkArgs& a, dim3 grid, bool ll, hipStream_t stream) {
  switch (ns) {
    case 3: return launch_stream<TU, 3, HM>(a, grid, ll, stream);
    case 4: return launch_stream<TU, 4, HM>(a, grid, ll, stream);
    case 5: return launch_stream<TU, 5, HM>(a, grid, ll, stream);
    default: return launch_stream<TU, 6, HM>(a, grid, ll, stream);
  }
}
template <bool HM>
int launch_stream_any(int tu, int ns, const TopkArgs& a, dim3 grid, bool ll, hipStream_t stream) {
  if (tu == 1) return launch_stream_ns<1, HM>(ns, a, grid, ll, stream);
  if (tu == 2) return launch_stream_ns<2, HM>(ns, a, grid, ll, stream);
  return launch_stream_ns<4, HM>(ns, a, grid, ll, stream);
}

// ------------------------------------------------------------------------------------------------
// Coarse pass over FRAGMENT-MAJOR one-plane images (rt_one_plane_to_fragments; TwoStage::h_only == 2).
//
// What bounds topk_stream_kernel<..., HM> with many user tiles is the LDS: every item chunk is written into it by the DMA ring,
// every user chunk is written into it again for every item block, and both are read back as fragments — ring 30 ms + products 19 ms
// + selection 7 ms per 4,096-user call over 5 M x 512, additive (profiles/r3_topk5m_u4096_one_plane_analysis.md).  Here
//   * the item fragment of a wave — 32 rows x 16 k of bf16 = the A operand of one v_mfma_f32_32x32x16_bf16 — is ONE coalesced 1 KB
//     global load straight into registers: an item row is used by exactly one wave, it has no business in the LDS.  The image is
//     stored fragment-major for that: 16-byte unit (row r, slot 2 s + half) at unit ((r / 32) n_s + s) 64 + 32 half + r % 32;
//   * the user tile (32 TU users x d bf16, fragment-major too) is copied into the LDS once per workgroup and stays there;
//   * a wave takes the 32-row slices of TWO item blocks at a time, so one B fragment read from the LDS feeds two products.
// LDS traffic per product: 1 KB per two v_mfma (was ~3 KB written + read per one).  P steps of item fragments are in flight per wave
// (plain loads in program order: the compiler's own vmcnt counting holds as long as no other vector-memory operation sits in the
// loop — the shared bound is therefore read from an LDS copy that wave 0 refreshes every few block pairs).
// Geometry, lists, bound, phases (seed / resume) and the merge are engine 2's: item blocks of IB rows, workgroup sx < n_seg takes
// the blocks blk_begin + sx + j n_seg, wave w their rows [32 w, 32 w + 32), lists in global memory.
// ---- selection of the fragment-major coarse pass: ONE list per user and WORKGROUP, in the LDS ------------------------------------------
// Engine 2's selection (a list per lane, in global memory for this kernel: the LDS holds the user tile) cost the 4,096-user pass more
// than its products — measured with the selection switched off (scripts/gpu/ablate.sh): products 19.6 ms, + threshold tests 10.2 ms,
// + inserts 15.0 ms (+ 9 ms with a viewed filter).  An insert read the lane's list and the filter's hash table from global memory, and
// loads return in order: every insert waited for the whole fragment pipeline of its wave (s_waitcnt vmcnt(0): ~4 us, twice per block
// pair and wave).  (Touching the blocks ahead of the demand loads — every wave of an XCD pulling its share of the next 12 blocks into the
// L2 — measured 19.3 vs 18.8 ms for the products alone: the loads are not what the waves wait for.)  Here the lists are the workgroup's: [UB][FRAG_KP] scores / positions behind the user tile, one spin lock, entry count
// and threshold per user, the viewed items as a 512-bit Bloom filter per user.  An insert is LDS traffic only (lgkmcnt: the fragment
// loads stay in flight); the exact probe of the filter's hash table — a global round trip — is left for candidates the Bloom filter
// cannot clear.  One list per user instead of eight also means one threshold per user: the bound tightens eight times faster.
// The thresholds every rejection used are published into `gthr` (no-return atomics) as before: the proof of the exact pass is unchanged.
constexpr int FRAG_KP = 16;            // list stride (entries); the lists hold a.k <= FRAG_KP entries
constexpr int FRAG_BLOOM_WORDS = 16;   // 512 bits per user

typedef __attribute__((address_space(3))) unsigned* lds_u32p;
typedef __attribute__((address_space(3))) int* lds_i32p;
typedef __attribute__((address_space(3))) float* lds_f32p;

struct FragSel {
  lds_u32p thr;      // [UB] ordered keys: max(shared bound as last seen, the list's worst entry once it is full)
  lds_i32p lock;     // [UB] 0 free / 1 held
  lds_i32p cnt;      // [UB]
  lds_f32p sc;       // [UB][FRAG_KP]
  lds_i32p pos;      // [UB][FRAG_KP]
  lds_u32p bloom;    // [UB][FRAG_BLOOM_WORDS] (only with a filter)
};
inline size_t frag_sel_lds_bytes(int ub) { return (size_t)ub * (4 + 4 + 4 + FRAG_KP * 8 + FRAG_BLOOM_WORDS * 4); }

// one candidate (score s at catalog position p) of user `ul` (tile-local): the caller holds no lock; returns after the list is updated
__device__ __forceinline__ void frag_insert(const TopkArgs& a, const FragSel& L, int ul, int u, float s, long long p, bool have) {
  // The loop leaves on a WAVE-UNIFORM condition and the critical section sits inside it.  With a per-lane exit (`while (!done)`) the
  // critical section is the loop's exit block: the compiler is free to run it after the loop has reconverged — a lane that holds a lock
  // then waits for a lane of its own wave that spins on a lock held, the same way, in another wave (seen as a hang on hardware).
  bool pending = have;              // (called by the whole wave: lanes without a candidate just keep the others company)
  while (__any(pending)) {
    if (pending && __hip_atomic_exchange(L.lock + ul, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
      asm volatile("" ::: "memory");
      int c = L.cnt[ul];
      const lds_f32p ls = L.sc + ul * FRAG_KP; const lds_i32p lp = L.pos + ul * FRAG_KP;
      // ONE pass over the list finds its worst entry and the runner-up: after the worst is replaced by the candidate the new worst is
      // the worse of the two — no second pass (the list is 16 entries of LDS; an insert is paid by the whole wave)
      float ws = INFINITY, w2s = INFINITY; long long wp = -1, w2p = -1; int wslot = 0;
      const bool full = c >= a.k;
      if (full || c + 1 == a.k) {
#pragma unroll
        for (int e = 0; e < FRAG_KP; ++e)
          if (e < c) {
            const float es = ls[e]; const long long ep = lp[e];
            if (wp < 0 || better(ws, wp, es, ep)) { w2s = ws; w2p = wp; ws = es; wp = ep; wslot = e; }
            else if (w2p < 0 || better(w2s, w2p, es, ep)) { w2s = es; w2p = ep; }
          }
      }
      float nws = ws; bool changed = false;        // nws: the worst entry of the list as it will be
      if (!full) {
        ls[c] = s; lp[c] = (int)p; c += 1; L.cnt[ul] = c; changed = true;
        if (c == a.k && !(wp < 0 || better(ws, wp, s, p))) nws = ws; else nws = s;      // (c == a.k: the scan above covered the other entries)
      } else if (better(s, p, ws, wp)) {
        ls[wslot] = s; lp[wslot] = (int)p; changed = true;
        nws = (w2p < 0 || better(w2s, w2p, s, p)) ? s : w2s;
      }
      if (changed && c == a.k) {      // the list is full: its worst entry bounds what it still accepts
        const unsigned key = f32_to_key(nws);
        const unsigned old = __hip_atomic_fetch_max(L.thr + ul, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (key > old) (void)__hip_atomic_fetch_max(a.gthr + u, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no return value: no wait
      }
      asm volatile("" ::: "memory");
      __hip_atomic_store(L.lock + ul, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    // (LDS operations of a wave execute in order)
      pending = false;
    }
  }
}

// absorb one finished item block of one wave: acc[tu][r] = score of row (r & 3) + 8 (r >> 2) + 4 half of the wave's 32 rows x user col
template <int TU>
__device__ __forceinline__ void frag_select_block(const TopkArgs& a, const FragSel& L, const f32x16 (&acc)[TU], long long pos0, int user0,
                                                  int lane, int wave) {
  const int col = lane & 31, half = lane >> 5;
  const long long row0 = pos0 + wave * 32;
  const int rows_left = (int)(a.n_cand - row0 < 32 ? (a.n_cand - row0 < 0 ? 0 : a.n_cand - row0) : 32);
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) {
    const int ul = tu * 32 + col, u = user0 + ul;
    const bool uvalid = u < a.n_users;
    float thr = key_to_f32(__hip_atomic_load(L.thr + ul, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));   // (other waves raise it)
    float mx = fmaxf(fmaxf(fmaxf(acc[tu][0], acc[tu][1]), fmaxf(acc[tu][2], acc[tu][3])), fmaxf(fmaxf(acc[tu][4], acc[tu][5]), fmaxf(acc[tu][6], acc[tu][7])));
    mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(acc[tu][8], acc[tu][9]), fmaxf(acc[tu][10], acc[tu][11])), fmaxf(fmaxf(acc[tu][12], acc[tu][13]), fmaxf(acc[tu][14], acc[tu][15]))));
    if (!__any(uvalid && mx >= thr)) continue;
    unsigned cmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (uvalid && row < rows_left && acc[tu][r] >= thr) cmask |= (1u << r);
    }
    // lanes l and l + 32 hold the same user (rows of the other half): one half at a time, so that no two lanes of a wave ever spin on
    // the same lock
    for (int hh = 0; hh < 2; ++hh) {
      unsigned mine = half == hh ? cmask : 0u;
      while (__any(mine != 0)) {       // wave-uniform: every lane takes every round (the lock loop inside needs the whole wave)
        bool have = false;
        float s = 0.f; long long p = 0;
        if (mine != 0) {
          const int r = __ffs(mine) - 1;
          mine &= mine - 1;
          s = acc[tu][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) s = (r == i) ? acc[tu][i] : s;      // static-index select: no scratch
          thr = key_to_f32(__hip_atomic_load(L.thr + ul, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          have = s >= thr;                                                   // (the bound may have risen since the tile's test)
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          p = row0 + row;
        }
        if (a.bloom && have) {      // viewed items: two bits of the user's Bloom filter clear most candidates without leaving the LDS
          const long long cid = p + a.id_offset;
          const unsigned h1 = ((unsigned)cid * 0x9E3779B1u) >> 23, h2 = ((unsigned)cid * 0x85EBCA77u) >> 23;
          const lds_u32p bl = L.bloom + ul * FRAG_BLOOM_WORDS;
          if (((bl[h1 >> 5] >> (h1 & 31)) & (bl[h2 >> 5] >> (h2 & 31)) & 1u) != 0) {
            have = !is_filtered(a, u, cid);                                  // exact: hash-table probe in global memory (a round trip)
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // leave with an empty VMEM scoreboard (see select_block)
          }
        }
        frag_insert(a, L, ul, u, s, p, have);
      }
    }
  }
}

template <int TU>
__global__ __launch_bounds__(NTHREADS) void topk_coarse_frag_kernel(TopkArgs a) {
  constexpr int UB = 32 * TU;
  constexpr int IW = 2;           // item blocks per step of a wave
  constexpr int P = 8;            // fragment loads in flight per wave and item block
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4* const ufrag = reinterpret_cast<u32x4*>(smem);                    // [TU][n_s][64] units of 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int S = a.n_seg;
  // Workgroup -> (segment sx, user tile ty).  A segment's item blocks are read by EVERY user tile; on the plain 2-D grid the tiles of a
  // segment are n_seg workgroups apart — resident at different times, on whichever XCD their linear index lands — and each streams its
  // share of the image from HBM again (PMC: 92 GB per 4,096-user launch for a 5.12 GB image).  1-D grid: workgroup w runs on XCD w % 8;
  // that XCD owns the segments sx = 8 i + w % 8 and walks them tile after tile, so the 32 CUs of an XCD hold the user tiles of ONE
  // segment at a time, reading the same item blocks in step: one HBM fetch, the others hit that XCD's L2.
  int sx = blockIdx.x, ty = blockIdx.y;
  if (a.xmap_gy > 0) {
    const int w = blockIdx.x, q = w >> 3;
    sx = (q / a.xmap_gy) * 8 + (w & 7);
    ty = q % a.xmap_gy;
    if (sx >= a.xmap_gx) return;
  }
  const int user0 = ty * UB;
  const int n_s = a.d >> 3;                                                // k = 16 slots per row (a.d = words per image row = d / 2)
  // behind the user tile: the workgroup's selection state (FragSel)
  FragSel L;
  {
    typedef __attribute__((address_space(3))) unsigned char* lds_bytep;
    lds_bytep base = (lds_bytep)(smem) + (size_t)TU * n_s * 64 * 16;
    L.thr = (lds_u32p)base; base += UB * 4;
    L.lock = (lds_i32p)base; base += UB * 4;
    L.cnt = (lds_i32p)base; base += UB * 4;
    L.sc = (lds_f32p)base; base += UB * FRAG_KP * 4;
    L.pos = (lds_i32p)base; base += UB * FRAG_KP * 4;
    L.bloom = (lds_u32p)base;
  }

  const long long n_blocks = a.blk_end - a.blk_begin;
  const long long my_blocks = (sx < S && n_blocks > sx) ? (n_blocks - sx + S - 1) / S : 0;
  // the workgroup's list of a user is list `slot` of its LISTS_PER_WG slots in the global arrays the merge kernels read (the other slots
  // stay empty); a resumed phase (after the seeding prefix) fills slot 1 and leaves the prefix's slot 0 as it is
  const int slot = a.resume ? 1 : 0;
  auto publish = [&](bool with_lists) {
    for (int i = tid; i < UB * LISTS_PER_WG; i += NTHREADS) {
      const int ul = i % UB, sl = i / UB, u = user0 + ul;
      if (u >= a.n_users || (a.resume && sl != slot)) continue;
      a.list_counts[(long long)(sx * LISTS_PER_WG + sl) * a.n_users_pad + u] = (with_lists && sl == slot) ? L.cnt[ul] : 0;
    }
    if (with_lists)
      for (int i = tid; i < UB * FRAG_KP; i += NTHREADS) {
        const int ul = i / FRAG_KP, e = i % FRAG_KP, u = user0 + ul;
        if (u >= a.n_users || e >= L.cnt[ul]) continue;
        const long long dst = ((long long)(sx * LISTS_PER_WG + slot) * a.n_users_pad + u) * a.k + e;
        a.list_scores[dst] = L.sc[i]; a.list_pos[dst] = L.pos[i];
      }
  };
  if (my_blocks == 0) { publish(false); return; }

  {   // the user tile (the image is padded to whole tiles), the bound, empty lists, the viewed items' Bloom filters
    const u32x4* ug = reinterpret_cast<const u32x4*>(a.users) + (long long)(user0 >> 5) * n_s * 64;
    for (int i = tid; i < TU * n_s * 64; i += NTHREADS) ufrag[i] = ug[i];
    for (int i = tid; i < UB; i += NTHREADS) {
      int u = user0 + i; if (u >= a.n_users_pad) u = a.n_users_pad - 1;
      L.thr[i] = __hip_atomic_load(a.gthr + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      L.lock[i] = 0; L.cnt[i] = 0;
    }
    if (a.bloom) {
      for (int i = tid; i < UB * FRAG_BLOOM_WORDS; i += NTHREADS) L.bloom[i] = 0u;
      __syncthreads();
      for (int ul = wave; ul < UB; ul += NTHREADS / 64) {
        const int u = user0 + ul;
        if (u >= a.n_users) continue;
        const long long lo = a.filt_indptr[u], hi = a.filt_indptr[u + 1];
        for (long long e = lo + lane; e < hi; e += 64) {
          const unsigned cid = (unsigned)a.filt_indices[e];
          const unsigned h1 = (cid * 0x9E3779B1u) >> 23, h2 = (cid * 0x85EBCA77u) >> 23;
          (void)__hip_atomic_fetch_or(L.bloom + ul * FRAG_BLOOM_WORDS + (h1 >> 5), 1u << (h1 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          (void)__hip_atomic_fetch_or(L.bloom + ul * FRAG_BLOOM_WORDS + (h2 >> 5), 1u << (h2 & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    __syncthreads();
  }

  const u32x4* const items = reinterpret_cast<const u32x4*>(a.items);
  const long long last_blk = a.blk_begin + sx + (my_blocks - 1) * S;
  // fragment (s = 0) of this wave's rows of item block `blk`, this lane's unit
  auto frag0 = [&](long long blk) -> const u32x4* { return items + ((blk * (IB / 32) + wave) * n_s) * 64 + lane; };
  const long long n_pairs = (my_blocks + IW - 1) / IW;
  f32x16 acc[IW][TU];
#pragma unroll
  for (int iw = 0; iw < IW; ++iw)
#pragma unroll
    for (int tu = 0; tu < TU; ++tu)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[iw][tu][r] = 0.f;

  // issue cursor: P slots ahead of the consumer, over the flattened (pair, slot) stream; past the end it re-reads the last block
  const u32x4* ip[IW]; int is = 0; long long ij = 0;
  auto set_pair = [&](long long j) {
#pragma unroll
    for (int iw = 0; iw < IW; ++iw) {
      long long blk = a.blk_begin + sx + (j * IW + iw) * S;
      if (blk > last_blk) blk = last_blk;
      ip[iw] = frag0(blk);
    }
  };
  set_pair(0);
  u32x4 abuf[P][IW];
  auto issue = [&](u32x4 (&dst)[IW]) {
#pragma unroll
    for (int iw = 0; iw < IW; ++iw) { dst[iw] = *ip[iw]; ip[iw] += 64; }   // (plain loads: the tiles of an XCD share the lines in L2)
    if (++is == n_s) { is = 0; ++ij; set_pair(ij < n_pairs ? ij : n_pairs - 1); }
  };
#pragma unroll
  for (int q = 0; q < P; ++q) issue(abuf[q]);

  // the user fragments of slot s + 1 are read while the products of slot s run (one wave per SIMD: nobody else hides the LDS latency)
  u32x4 bf[2][TU];
#pragma unroll
  for (int tu = 0; tu < TU; ++tu) bf[0][tu] = ufrag[(tu * n_s) * 64 + lane];
#pragma unroll 1
  for (long long j = 0; j < n_pairs; ++j) {
#pragma unroll 1
    for (int s0 = 0; s0 < n_s; s0 += P) {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const int nxt = (s0 + q + 1 == n_s) ? 0 : s0 + q + 1;     // the tile is the same for every block pair: the slot index just wraps
#pragma unroll
        for (int tu = 0; tu < TU; ++tu) bf[(q + 1) & 1][tu] = ufrag[(tu * n_s + nxt) * 64 + lane];
        u32x4 af[IW];
#pragma unroll
        for (int iw = 0; iw < IW; ++iw) af[iw] = abuf[q][iw];
        issue(abuf[q]);
#pragma unroll
        for (int iw = 0; iw < IW; ++iw)
#pragma unroll
          for (int tu = 0; tu < TU; ++tu)
            acc[iw][tu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[iw]), __builtin_bit_cast(bf16x8, bf[q & 1][tu]),
                                                                  acc[iw][tu], 0, 0, 0);
      }
    }
#pragma unroll
    for (int iw = 0; iw < IW; ++iw) {
      const long long blk = a.blk_begin + sx + (j * IW + iw) * S;
#ifdef RT_ABLATION_BUILD
      if (a.debug & 1) { if (acc[iw][0][0] == 1.2345e-30f) a.gthr[0] = 0u; } else       // (no selection: the products only)
#endif
      if (blk <= last_blk) frag_select_block<TU>(a, L, acc[iw], blk * IB, user0, lane, wave);
#pragma unroll
      for (int tu = 0; tu < TU; ++tu)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[iw][tu][r] = 0.f;
    }
    // fold the shared bound (other segments' lists) into the thresholds: the only vector-memory read of the loop.  Its wave waits for the
    // load behind its whole fragment pipeline (loads return in order): every 16 pairs, and the four waves take turns — the workgroup ends
    // with its slowest wave
    if ((j & 15) == 15 && wave == (int)((j >> 4) & 3)) {
      for (int i = lane; i < UB; i += 64) {
        int u = user0 + i; if (u >= a.n_users_pad) u = a.n_users_pad - 1;
        (void)__hip_atomic_fetch_max(L.thr + i, __hip_atomic_load(a.gthr + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  publish(true);
}

inline size_t coarse_frag_lds_bytes(int tu, int d_words) { return (size_t)tu * (d_words / 8) * 64 * 16 + frag_sel_lds_bytes(32 * tu); }

template <int TU>
int launch_coarse_frag_t(const TopkArgs& a0, dim3 grid, hipStream_t stream) {
  TopkArgs a = a0;
  if (grid.y > 1) {     // 1-D grid, XCD-owned segments (see the kernel)
    a.xmap_gx = (int)grid.x; a.xmap_gy = (int)grid.y;
    grid = dim3(8u * ((grid.x + 7u) / 8u) * grid.y, 1u, 1u);
  }
  a.bloom = (a.filt_indptr != nullptr && a.filt_indices != nullptr) ? 1 : 0;      // (512 bits per user in the LDS, in front of the exact test)
  const size_t lds = coarse_frag_lds_bytes(TU, a.d);
  static size_t attr_lds = 0;
  if (lds > 64 * 1024 && lds > attr_lds) {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_coarse_frag_kernel<TU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  topk_coarse_frag_kernel<TU><<<grid, NTHREADS, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
int launch_coarse_frag(int tu, const TopkArgs& a, dim3 grid, hipStream_t stream) {
  if (tu == 1) return launch_coarse_frag_t<1>(a, grid, stream);
  if (tu == 2) return launch_coarse_frag_t<2>(a, grid, stream);
  return launch_coarse_frag_t<4>(a, grid, stream);
}

// ---- 16-user tile: plan and launch -------------------------------------------------------------------------------
struct Plan16 {
  int S, n_tiles, n_users_pad, users_per_launch, ns;
  long long seed_blocks;     // item blocks of the seeding prefix (2 per workgroup), 0 = single phase
  size_t o_gthr, o_scores, o_pos, o_counts, total;
};

inline bool wants_tile16(int n_users, int k, int users_per_pass) {
  if (k > K_LDS_LISTS) return false;                       // its lists live in LDS
  if (users_per_pass > 0) return users_per_pass <= 16;
  return n_users <= 16;                                    // auto: a launch that fits one 16-user tile is HBM-bound
}

inline Plan16 make_plan16(int n_users, long long n_cand, int k) {
  Plan16 P;
  P.users_per_launch = n_users < MAX_USERS_PER_LAUNCH ? n_users : MAX_USERS_PER_LAUNCH;
  P.n_tiles = (P.users_per_launch + UB16 - 1) / UB16;
  P.n_users_pad = P.n_tiles * UB16;
  const long long n_blocks = (n_cand + IB - 1) / IB;
  P.ns = 3;
  for (int ns = 7; ns >= 3; --ns)
    if (stream16_lds_bytes(ns, k) <= LDS_PER_CU) { P.ns = ns; break; }
  long long S = ((long long)rt_num_cus() + P.n_tiles - 1) / P.n_tiles;   // one workgroup per CU owns the whole LDS
  if (S > n_blocks) S = n_blocks;
  if (S < 1) S = 1;
  if (S > 512) S = 512;                                    // topk_select_kernel: a thread owns the heads of <= 4 lists
  P.S = (int)S;
  P.seed_blocks = 0;
  if (n_blocks >= 32 * S && P.n_tiles <= 16) P.seed_blocks = 2 * S;
  size_t o = 0;
  P.o_gthr = o; o = align_up(o + (size_t)P.n_users_pad * 4, 256);
  const size_t ent = (size_t)2 * P.S * P.n_users_pad * (size_t)k;     // list region of the main pass + of the seeding prefix
  P.o_scores = o; o = align_up(o + ent * 4, 256);
  P.o_pos = o; o = align_up(o + ent * 4, 256);
  P.o_counts = o; o = align_up(o + (size_t)2 * P.S * P.n_users_pad * 4, 256);
  P.total = o;
  return P;
}

template <int NS, bool WL>
int launch_stream16_t(const TopkArgs& a, dim3 grid, hipStream_t stream) {
  const size_t lds = stream16_lds_bytes(NS, a.k);
  static size_t attr_lds = 0;
  if (lds > 64 * 1024 && lds > attr_lds) {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_stream16_kernel<NS, WL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  topk_stream16_kernel<NS, WL><<<grid, NTHREADS + 128, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}
template <bool WL>
int launch_stream16_ns(int ns, const TopkArgs& a, dim3 grid, hipStream_t stream) {
  switch (ns) {
    case 3: return launch_stream16_t<3, WL>(a, grid, stream);
    case 4: return launch_stream16_t<4, WL>(a, grid, stream);
    case 5: return launch_stream16_t<5, WL>(a, grid, stream);
    case 6: return launch_stream16_t<6, WL>(a, grid, stream);
    default: return launch_stream16_t<7, WL>(a, grid, stream);
  }
}
template <bool SEED>
int launch_select(const MergeArgs& m, int first_list, int n_lists, int n_lists_total, unsigned* gthr, int n_users, hipStream_t stream) {
  const size_t lds = (size_t)n_lists * m.k * 8;
  static size_t attr_lds = 0;
  if (lds > 64 * 1024 && lds > attr_lds) {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_select_kernel<SEED>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds = lds;
  }
  topk_select_kernel<SEED><<<n_users, 256, lds, stream>>>(m, first_list, n_lists, n_lists_total, gthr);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

template <int TU>
int launch_staged(const TopkArgs& a, dim3 grid, hipStream_t stream) {
  constexpr int UB = 32 * TU;
  const size_t lds = (size_t)(2 * IB * LDK + 2 * UB * LDK) * sizeof(float);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_staged_kernel<TU>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  topk_staged_kernel<TU><<<grid, NTHREADS, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace

extern "C" {

size_t rt_topk_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t users_per_pass) {
  if (n_users <= 0 || n_candidates <= 0 || k <= 0) return 256;
  long long kk = k < n_candidates ? k : n_candidates;
  size_t need = make_plan(n_users, n_candidates, (int)kk, users_per_pass).total;   // also the fallback when d % 32 != 0
  if (wants_tile16(n_users, (int)kk, users_per_pass)) {
    const size_t need16 = make_plan16(n_users, n_candidates, (int)kk).total;
    if (need16 > need) need = need16;
  }
  return need;
}

// bytes of the per-user hash tables over a filter CSR with `nnz` indices
size_t rt_filter_hash_bytes(int32_t n_users, int64_t nnz) { return 4 * (4 * (size_t)nnz + 4 * (size_t)n_users + 4); }

// Build the tables rt_topk_score probes instead of binary-searching the CSR rows (optional; pass NULL to skip them).
int rt_filter_hash_build(const int64_t* filt_indptr, const int32_t* filt_indices, int32_t n_users, int64_t nnz, int32_t* hash,
                         hipStream_t stream) {
  (void)hipGetLastError();
  if (n_users < 0 || nnz < 0 || nnz >= (1LL << 29)) return RT_ERR_INVALID_ARG;
  if (n_users == 0) return RT_OK;
  RT_CHECK_HIP(hipMemsetAsync(hash, 0xFF, rt_filter_hash_bytes(n_users, nnz), stream));
  filter_hash_build_kernel<<<(n_users + 3) / 4, 256, 0, stream>>>(reinterpret_cast<const long long*>(filt_indptr), filt_indices,
                                                                   n_users, hash);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Two-stage call (rt_topk_score_two_stage): the streaming pass runs over the hm images with k-entry lists as usual, the merge hands
// out the k_cand best coarse candidates per user as positions, the exact pass (topk_replay_kernel) scores and orders them.
struct TwoStage {
  const float* users_hm; const float* items_hm;     // images: users dense [n_users, d] (row u = user u of the call), items strided like `items`
  const float* user_norms; float max_item_norm;
  int k_cand; int* out_unproven;
  int cosine;      // the images hold L2-normalised rows: stage 1 ranks their dot products, stage 2 the exact cosine of the fp32 rows
  int h_only;      // the images hold ONE bf16 (RNE) per value — half the bytes of the catalog: the HBM-bound regime (a few users per pass)
  long long items_img_stride;   // row stride of items_hm in 32-bit words (hm: = item_stride; h-only: bf16 stride / 2)
};
// list capacity of the coarse pass: a pair is dropped when it falls below its list's worst kept entry, and the proof needs that bound to
// stay below the k-th exact score — k entries per list would put the bound AT the best score for k = 1 and leave no room when one list
// happens to hold all of a user's top k; the spare entries up to the next multiple of four (the LDS lists are padded to it anyway, at
// least one) keep the bound below the (k + 1)-th best score
inline int two_stage_list_k(int k) { const int kl = (k + 4) & ~3; return kl < K_LDS_LISTS ? kl : K_LDS_LISTS; }
inline size_t two_stage_extra_bytes(int users_per_launch, int k_cand) {
  return align_up((size_t)users_per_launch * k_cand * 4, 256) * 2 + align_up((size_t)users_per_launch * 4, 256);
}

static int topk_score_impl(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users,
                           const float* items, int64_t item_stride, const int64_t* whitelist, int64_t n_candidates,
                           int64_t candidate_id_offset, int32_t d, int32_t distance, int32_t k,
                           const int64_t* filt_indptr, const int32_t* filt_indices, const int32_t* filt_hash,
                           int64_t* out_ids, float* out_scores, int32_t* out_counts,
                           void* workspace, size_t workspace_bytes, int32_t users_per_pass, const TwoStage* ts, hipStream_t stream) {
  (void)hipGetLastError();  // do not inherit a stale error from the caller's earlier HIP calls
  if (n_users < 0 || n_candidates < 0 || d <= 0 || (d & 3) != 0 || k <= 0) return RT_ERR_INVALID_ARG;
  if (distance < DIST_DOT || distance > DIST_EUCLID) return RT_ERR_INVALID_ARG;
  if (ts != nullptr && ts->h_only && d % (2 * KC) != 0) return RT_ERR_UNSUPPORTED;
  if (ts != nullptr && ts->h_only == 2 && (d % 128 != 0 || whitelist != nullptr)) return RT_ERR_UNSUPPORTED;   // fragment-major images: 8 slots per unrolled step, rows in place
  if (ts != nullptr && ((distance != DIST_DOT && distance != DIST_COSINE) || d % KC != 0 || k > K_LDS_LISTS || (ts->k_cand != 32 && ts->k_cand != 64) || ts->k_cand < k))
    return RT_ERR_UNSUPPORTED;
  if ((user_stride & 3) != 0 || (item_stride & 3) != 0) return RT_ERR_INVALID_ARG;
  if (((uintptr_t)users & 15) != 0 || ((uintptr_t)items & 15) != 0) return RT_ERR_INVALID_ARG;
  if (n_candidates >= (1LL << 31)) return RT_ERR_UNSUPPORTED;
  if (k > n_candidates) return RT_ERR_INVALID_ARG;  // caller clamps: k = min(k, n_candidates)
  if (n_users == 0) return RT_OK;
  if (n_candidates == 0) {
    return hipMemsetAsync(out_counts, 0, sizeof(int32_t) * (size_t)n_users, stream) == hipSuccess ? RT_OK : RT_ERR_LAUNCH;
  }
  const bool stream_ok = ts != nullptr || (d % KC == 0);      // else: the register-staged engine (any d)
  char* ws = reinterpret_cast<char*>(workspace);
  if (ts == nullptr && stream_ok && wants_tile16(n_users, k, users_per_pass)) {
    // ---- 16-user tile (engine 3): [seeding prefix -> seed] -> main pass -> selection over the merged lists ----
    const Plan16 Q = make_plan16(n_users, n_candidates, k);
    if (workspace == nullptr || workspace_bytes < Q.total) return RT_ERR_WORKSPACE;
    const long long n_blocks = (n_candidates + IB - 1) / IB;
    for (int u0 = 0; u0 < n_users; u0 += Q.users_per_launch) {
      const int nb = (n_users - u0) < Q.users_per_launch ? (n_users - u0) : Q.users_per_launch;
      const int n_tiles = (nb + UB16 - 1) / UB16;
      TopkArgs a{};
      a.users = user_rows ? users : users + (long long)u0 * user_stride;
      a.user_stride = user_stride;
      a.user_rows = user_rows ? reinterpret_cast<const long long*>(user_rows) + u0 : nullptr;
      a.n_users = nb;
      a.items = items; a.item_stride = item_stride;
      a.whitelist = reinterpret_cast<const long long*>(whitelist);
      a.n_cand = n_candidates; a.id_offset = whitelist ? 0 : candidate_id_offset;
      a.d = d; a.distance = distance; a.k = k;
      a.filt_indptr = filt_indptr ? reinterpret_cast<const long long*>(filt_indptr) + u0 : nullptr;
      a.filt_indices = filt_indices;
      a.filt_hash = filt_indptr ? filt_hash : nullptr; a.filt_u0 = u0;
      a.list_scores = reinterpret_cast<float*>(ws + Q.o_scores);
      a.list_pos = reinterpret_cast<int*>(ws + Q.o_pos);
      a.list_counts = reinterpret_cast<int*>(ws + Q.o_counts);
      a.n_users_pad = Q.n_users_pad;
      a.gthr = reinterpret_cast<unsigned*>(ws + Q.o_gthr);
      a.rotate = 1;
      a.debug = 0;
      a.n_seg = Q.S; a.resume = 0;
      a.n_lists_total = Q.seed_blocks > 0 ? 2 * Q.S : Q.S;
      if (Q.seed_blocks == 0 || Q.n_users_pad > nb) {   // with a seeding prefix the seed kernel stores the bound of every real user
        fill_u32_kernel<<<(Q.n_users_pad + 255) / 256, 256, 0, stream>>>(a.gthr, 0x007FFFFFu /* key(-inf) */, Q.n_users_pad);
        RT_CHECK_LAUNCH();
      }
      dim3 grid(Q.S, n_tiles);
      MergeArgs m{};
      m.list_scores = a.list_scores; m.list_pos = a.list_pos; m.list_counts = a.list_counts;
      m.n_users_pad = Q.n_users_pad; m.k = k; m.n_users = nb;
      m.whitelist = a.whitelist; m.id_offset = a.id_offset; m.distance = distance;
      m.out_ids = reinterpret_cast<long long*>(out_ids) + (long long)u0 * k;
      m.out_scores = out_scores + (long long)u0 * k;
      m.out_counts = out_counts + u0;
      int rc, n_lists = Q.S;
      if (Q.seed_blocks > 0) {   // seeding prefix: lists [S, 2S); its k-th best per user becomes the shared bound
        a.blk_begin = 0; a.blk_end = Q.seed_blocks; a.list_base = Q.S; a.use_bound = 0;
        rc = a.whitelist ? launch_stream16_ns<true>(Q.ns, a, grid, stream) : launch_stream16_ns<false>(Q.ns, a, grid, stream);
        if (rc != RT_OK) return rc;
        rc = launch_select<true>(m, Q.S, Q.S, a.n_lists_total, a.gthr, nb, stream);
        if (rc != RT_OK) return rc;
        n_lists = 2 * Q.S;
      }
      a.blk_begin = Q.seed_blocks; a.blk_end = n_blocks; a.list_base = 0; a.use_bound = 1;
      rc = a.whitelist ? launch_stream16_ns<true>(Q.ns, a, grid, stream) : launch_stream16_ns<false>(Q.ns, a, grid, stream);
      if (rc != RT_OK) return rc;
      rc = launch_select<false>(m, 0, n_lists, a.n_lists_total, nullptr, nb, stream);
      if (rc != RT_OK) return rc;
    }
    return RT_OK;
  }
  const int k_out = k;                        // two-stage: the lists hold more than the k entries the caller asked for
  if (ts != nullptr) k = ts->h_only == 2 ? FRAG_KP : two_stage_list_k(k);     // (the fragment pass keeps ONE list per user and workgroup: a longer one)
  Plan P = make_plan(n_users, n_candidates, k, users_per_pass);
  if (ts != nullptr && ts->h_only == 2) {
    P.lds_lists = false;                                   // topk_coarse_frag_kernel: the LDS holds the user tile
    if (coarse_frag_lds_bytes(P.tu, d / 2) > LDS_PER_CU) return RT_ERR_UNSUPPORTED;
  }
  const size_t ws_need = P.total + (ts ? two_stage_extra_bytes(P.users_per_launch, ts->k_cand) : 0);
  if (workspace == nullptr || workspace_bytes < ws_need) return RT_ERR_WORKSPACE;

  for (int u0 = 0; u0 < n_users; u0 += P.users_per_launch) {
    const int nb = (n_users - u0) < P.users_per_launch ? (n_users - u0) : P.users_per_launch;
    const int n_tiles = (nb + P.ub - 1) / P.ub;
    TopkArgs a{};
    a.users = user_rows ? users : users + (long long)u0 * user_stride;
    a.user_stride = user_stride;
    a.user_rows = user_rows ? reinterpret_cast<const long long*>(user_rows) + u0 : nullptr;
    a.n_users = nb;
    a.items = items; a.item_stride = item_stride;
    const int d_img = (ts != nullptr && ts->h_only) ? d / 2 : d;      // words per image row
    if (ts != nullptr) {   // the coarse pass streams the images (users: dense rows in call order)
      a.users = ts->users_hm + (long long)u0 * d_img; a.user_stride = d_img; a.user_rows = nullptr;
      a.items = ts->items_hm; a.item_stride = ts->items_img_stride; a.h_only = ts->h_only;
    }
    a.whitelist = reinterpret_cast<const long long*>(whitelist);
    a.n_cand = n_candidates; a.id_offset = whitelist ? 0 : candidate_id_offset;
    a.d = d_img; a.distance = ts != nullptr ? DIST_DOT : distance; a.k = k;     // (two-stage: the coarse pass ranks dot products of the images)
    a.filt_indptr = filt_indptr ? reinterpret_cast<const long long*>(filt_indptr) + u0 : nullptr;
    a.filt_indices = filt_indices;
    a.filt_hash = filt_indptr ? filt_hash : nullptr; a.filt_u0 = u0;
    a.list_scores = reinterpret_cast<float*>(ws + P.o_scores);
    a.list_pos = reinterpret_cast<int*>(ws + P.o_pos);
    a.list_counts = reinterpret_cast<int*>(ws + P.o_counts);
    a.n_users_pad = P.n_users_pad;
    a.gthr = reinterpret_cast<unsigned*>(ws + P.o_gthr);
    a.rotate = 1;
#ifdef RT_ABLATION_BUILD
    a.debug = env_int("RT_TOPK_DEBUG", 0);   // 1 = skip selection: only exists in ablation builds (-DRT_ABLATION_BUILD)
#else
    a.debug = 0;
#endif

    fill_u32_kernel<<<(P.n_users_pad + 255) / 256, 256, 0, stream>>>(a.gthr, 0x007FFFFFu /* key(-inf) */,
                                                                     P.n_users_pad);
    RT_CHECK_LAUNCH();
    dim3 grid(P.S, n_tiles);
    const long long n_blocks = (n_candidates + IB - 1) / IB;
    MergeArgs m{};
    m.list_scores = a.list_scores; m.list_pos = a.list_pos; m.list_counts = a.list_counts;
    m.n_lists = P.n_lists; m.n_users_pad = P.n_users_pad; m.k = k; m.n_users = nb;
    auto run_phase = [&](long long b0, long long b1, int n_seg, int resume) -> int {
      a.blk_begin = b0; a.blk_end = b1; a.n_seg = n_seg; a.resume = resume;
      if (ts != nullptr && ts->h_only == 2) return launch_coarse_frag(P.tu, a, grid, stream);
      if (ts != nullptr) return launch_stream_any<true>(P.tu, P.ns, a, grid, P.lds_lists, stream);
      if (stream_ok) return launch_stream_any<false>(P.tu, P.ns, a, grid, P.lds_lists, stream);
      if (P.tu == 1) return launch_staged<1>(a, grid, stream);
      if (P.tu == 2) return launch_staged<2>(a, grid, stream);
      return launch_staged<4>(a, grid, stream);
    };
    int rc;
    if (P.blocks_seed > 0) {
      rc = run_phase(0, P.blocks_seed, P.S_seed, 0);  // workgroups sx >= S_seed get no blocks, publish empty lists
      if (rc != RT_OK) return rc;
      MergeArgs ms = m; ms.n_lists = P.S_seed * LISTS_PER_WG;
      if (ms.n_lists >= 256) topk_seed_kernel<1024><<<nb, 1024, 0, stream>>>(ms, a.gthr);
      else topk_seed_kernel<256><<<nb, 256, 0, stream>>>(ms, a.gthr);
      RT_CHECK_LAUNCH();
      rc = run_phase(P.blocks_seed, n_blocks, P.S, 1);
    } else {
      rc = run_phase(0, n_blocks, P.S, 0);
    }
    if (rc != RT_OK) return rc;

    m.compact_scores = reinterpret_cast<float*>(ws + P.o_cscores);
    m.compact_pos = reinterpret_cast<int*>(ws + P.o_cpos);
    m.compact_cap = (long long)P.n_lists * k;
    m.whitelist = a.whitelist; m.id_offset = a.id_offset; m.distance = a.distance;
    m.out_ids = reinterpret_cast<long long*>(out_ids) + (long long)u0 * k;
    m.out_scores = out_scores + (long long)u0 * k;
    m.out_counts = out_counts + u0;
    ReplayArgs r{};
    if (ts != nullptr) {   // candidates (positions, coarse scores, counts) behind the plan's buffers
      char* extra = ws + P.total;
      const size_t cand_bytes = align_up((size_t)P.users_per_launch * ts->k_cand * 4, 256);
      r.cand_pos = reinterpret_cast<int*>(extra); r.cand_coarse = reinterpret_cast<float*>(extra + cand_bytes);
      r.cand_counts = reinterpret_cast<int*>(extra + 2 * cand_bytes);
      m.out_k = ts->k_cand; m.out_pos = const_cast<int*>(r.cand_pos); m.out_scores = const_cast<float*>(r.cand_coarse);
      m.out_counts = const_cast<int*>(r.cand_counts); m.out_ids = nullptr;
    }
    if (m.n_lists >= 256) topk_merge_kernel<1024><<<nb, 1024, 0, stream>>>(m);
    else topk_merge_kernel<256><<<nb, 256, 0, stream>>>(m);
    RT_CHECK_LAUNCH();
    if (ts != nullptr) {
      r.users = user_rows ? users : users + (long long)u0 * user_stride; r.user_stride = user_stride;
      r.user_rows = user_rows ? reinterpret_cast<const long long*>(user_rows) + u0 : nullptr;
      r.items = items; r.item_stride = item_stride; r.whitelist = a.whitelist; r.id_offset = a.id_offset;
      r.d = d; r.k = k_out; r.kc = ts->k_cand; r.rotate = a.rotate; r.cosine = ts->cosine;
      r.gthr = a.gthr; r.user_norms = ts->user_norms + u0; r.max_item_norm = ts->max_item_norm;
      // |coarse - exact| <= (2^-14 [the dropped l parts] + 4 d 2^-24 [fp32 accumulation of the 4 d bf16 products]
      //                      + d 2^-24 [the exact chain's own rounding]) sum_k |u_k v_k|, with 3 % slack (norms are fp32 too)
      r.err_coef = 1.03f * (6.103515625e-5f + 5.0f * (float)d * 5.9604644775390625e-8f);
      // h-only images: each operand rounded to nearest bf16 — 8 significant bits, unit roundoff 2^-8 (x = 1 + 2^-8 rounds to 1) —
      // so |u v - u^ v^| <= (2^-7 + 2^-16) |u v| per product (tests/test_two_stage_bounds.py: rows of such halfway values reach
      // 0.99 of it); d bf16 products accumulated in fp32 + the exact chain's own rounding
      if (ts->h_only) r.err_coef = 1.03f * (7.8125e-3f + 1.52587890625e-5f + 2.0f * (float)d * 5.9604644775390625e-8f);
      r.out_ids = reinterpret_cast<long long*>(out_ids) + (long long)u0 * k_out; r.out_scores = out_scores + (long long)u0 * k_out;
      r.out_counts = out_counts + u0; r.out_unproven = ts->out_unproven + u0;
      if (ts->k_cand == 64) topk_replay_kernel<2><<<nb, 128, 0, stream>>>(r);
      else topk_replay_kernel<1><<<nb, 64, 0, stream>>>(r);
      RT_CHECK_LAUNCH();
    }
  }
  return RT_OK;
}

int rt_topk_score(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users,
                  const float* items, int64_t item_stride, const int64_t* whitelist, int64_t n_candidates,
                  int64_t candidate_id_offset, int32_t d, int32_t distance, int32_t k,
                  const int64_t* filt_indptr, const int32_t* filt_indices, const int32_t* filt_hash,
                  int64_t* out_ids, float* out_scores, int32_t* out_counts,
                  void* workspace, size_t workspace_bytes, int32_t users_per_pass, hipStream_t stream) {
  return topk_score_impl(users, user_stride, user_rows, n_users, items, item_stride, whitelist, n_candidates, candidate_id_offset, d,
                         distance, k, filt_indptr, filt_indices, filt_hash, out_ids, out_scores, out_counts, workspace,
                         workspace_bytes, users_per_pass, nullptr, stream);
}

size_t rt_topk_two_stage_workspace_bytes(int32_t n_users, int64_t n_candidates, int32_t k, int32_t k_cand, int32_t users_per_pass) {
  if (n_users <= 0 || n_candidates <= 0 || k <= 0 || k_cand <= 0) return 0;
  const Plan P = make_plan(n_users, n_candidates, FRAG_KP, users_per_pass);       // (the longest lists any coarse pass keeps)
  return P.total + two_stage_extra_bytes(P.users_per_launch, k_cand);
}

// Two-stage exact top-k for dot products and cosine similarity (the MFMA-bound regime: many users per catalog pass).  Cosine: the images
// hold L2-NORMALISED rows (rt_to_hm_rows normalize = 1), stage 1 ranks their dot products, stage 2 evaluates rt_topk_score's cosine
// expression on the fp32 rows with the row norms accumulated as engine 2 accumulates them.  Stage 1 = the streaming selection of
// rt_topk_score over hm images (rt_to_hm_rows; two bf16 matrix instructions per four k instead of four f32-input ones), keeping the
// k_cand (32 or 64) best COARSE candidates per user; stage 2 = their exact scores in the arithmetic of rt_topk_score's 32-wide engine,
// ordered (score desc, position asc).  out_unproven[u] = 0: the k results of user u are exactly rt_topk_score's (ids, order, score
// bits); 1: the candidate set could not be proven complete for u (ties or near-ties at the k-th place closer than the coarse error
// bound) — rank those users with rt_topk_score.  users_hm [n_users, d] dense in call order; items_hm strided and offset like `items`;
// user_norms [n_users] and max_item_norm = L2 norms (rt_to_hm_rows).  d % 32 == 0, k <= 16 <= k_cand.
// h_only = 1: the images hold ONE round-to-nearest bf16 per value (rt_to_hm_rows mode 2 / 3; rows of item_stride bf16 values, users dense
// d) — half the catalog bytes per pass, for the regime where the pass is bound by HBM (a few users); coarse error 2^-7 |u| |v|, d % 64 == 0.
int rt_topk_score_two_stage(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users, const float* items,
                            int64_t item_stride, const uint32_t* users_hm, const uint32_t* items_hm, int32_t h_only, const float* user_norms,
                            float max_item_norm, const int64_t* whitelist, int64_t n_candidates, int64_t candidate_id_offset, int32_t d,
                            int32_t distance, int32_t k, int32_t k_cand, const int64_t* filt_indptr, const int32_t* filt_indices,
                            const int32_t* filt_hash, int64_t* out_ids, float* out_scores, int32_t* out_counts, int32_t* out_unproven,
                            void* workspace, size_t workspace_bytes, int32_t users_per_pass, hipStream_t stream) {
  if (users_hm == nullptr || items_hm == nullptr || user_norms == nullptr || out_unproven == nullptr || !(max_item_norm >= 0.f) ||
      ((uintptr_t)users_hm & 15) != 0 || ((uintptr_t)items_hm & 15) != 0)
    return RT_ERR_INVALID_ARG;
  if (distance != DIST_DOT && distance != DIST_COSINE) return RT_ERR_UNSUPPORTED;
  if (n_users > 0 && n_candidates == 0) {
    if (hipMemsetAsync(out_unproven, 0, sizeof(int32_t) * (size_t)n_users, stream) != hipSuccess) return RT_ERR_LAUNCH;
  }
  TwoStage ts{reinterpret_cast<const float*>(users_hm), reinterpret_cast<const float*>(items_hm), user_norms, max_item_norm, k_cand,
              out_unproven, distance == DIST_COSINE ? 1 : 0, h_only == 2 ? 2 : (h_only ? 1 : 0), h_only ? item_stride / 2 : item_stride};
  if (h_only == 1 && (item_stride & 7) != 0) return RT_ERR_INVALID_ARG;  // (the bf16 image mirrors the fp32 rows: stride item_stride bf16 values)
  return topk_score_impl(users, user_stride, user_rows, n_users, items, item_stride, whitelist, n_candidates, candidate_id_offset, d,
                         distance, k, filt_indptr, filt_indices, filt_hash, out_ids, out_scores, out_counts, workspace, workspace_bytes,
                         users_per_pass, &ts, stream);
}

}  // extern "C"
