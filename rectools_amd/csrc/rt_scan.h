// Exclusive scan of an int histogram (counting sort of ids): 3 launches — per-block scan of 4096 elements, scan of the
// block totals, add-back.  Shared by the sampled-loss backward (rt_loss.hip) and the embedding backward (rt_rowops.hip).
#pragma once
#include "rt_common.h"

namespace {

constexpr int SCAN_T = 1024, SCAN_E = 4;   // 4096 elements per block
__global__ __launch_bounds__(SCAN_T) void scan_local_kernel(const int* __restrict__ count, int n, int* __restrict__ offsets,
                                                            int* __restrict__ blocksum) {
  __shared__ int s_w[SCAN_T / 64];
  const int base = (blockIdx.x * SCAN_T + threadIdx.x) * SCAN_E;
  int v[SCAN_E], t = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) { v[e] = (base + e < n) ? count[base + e] : 0; t += v[e]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int x = (lane < SCAN_T / 64) ? s_w[lane] : 0, inc = x;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane < SCAN_T / 64) s_w[lane] = inc - x;   // exclusive prefix of the wave totals
    if (lane == SCAN_T / 64 - 1) blocksum[blockIdx.x] = inc;
  }
  __syncthreads();
  int run = s_w[wave] + incl - t;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) { if (base + e < n) offsets[base + e] = run; run += v[e]; }
}
__global__ __launch_bounds__(1024) void scan_blocksums_kernel(int* __restrict__ blocksum, int nb) {
  __shared__ int s_w[16]; __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int x = (i < nb) ? blocksum[i] : 0;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
      int y = (lane < 16) ? s_w[lane] : 0, yi = y;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { int u = __shfl_up(yi, o, 64); if (lane >= o) yi += u; }
      if (lane < 16) s_w[lane] = yi - y;
    }
    __syncthreads();
    const int excl = s_carry + s_w[wave] + inc - x;
    if (i < nb) blocksum[i] = excl;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = excl + x;
    __syncthreads();
  }
}
// Finalises the scan and cuts POPULAR keys into chunks: a key with more than `heavy_t` entries gets ceil(count / chunk)
// consecutive slots of the chunk list (heavy_key / heavy_chunk) and cursor[key] = its first slot; every other key gets
// cursor[key] = -1.  A chunk is reduced by its own workgroup into a partial row (slab), so the cost of a popular key is
// spread over the chip instead of serialising on one wave or one workgroup.
__global__ void scan_add_kernel(int* __restrict__ offsets, int* __restrict__ cursor, const int* __restrict__ blocksum, int n,
                                const int* __restrict__ count, int heavy_t, int chunk, int* __restrict__ heavy_count,
                                int* __restrict__ heavy_key, int* __restrict__ heavy_chunk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  offsets[i] = offsets[i] + blocksum[i / (SCAN_T * SCAN_E)];
  int slot = -1;
  if (heavy_count != nullptr && count[i] > heavy_t) {
    const int n_ch = (count[i] + chunk - 1) / chunk;
    slot = atomicAdd(heavy_count, n_ch);
    for (int c = 0; c < n_ch; ++c) { heavy_key[slot + c] = i; heavy_chunk[slot + c] = c; }
  }
  cursor[i] = slot;
}

// The whole scan in ONE launch for histograms of up to 32 K bins (the C2 catalog: 26,745): one 1024-thread workgroup, every thread owns
// E = ceil(n / 1024) CONSECUTIVE bins — ONE round of loads, a block-wide scan of the 1024 thread totals, one round of stores with each
// bin finalised on the spot (what scan_add_kernel does).  The three-launch form costs three dependent ~5 us launches for ~100 KB of data,
// twice per training step; a first single-launch form that walked the bins in 4096-element rounds with a carry paid a dependent global
// round trip per round (26 us for 7 rounds, profiles/r4_timeline_train.txt).
constexpr int SCAN1_E = 32;
__global__ __launch_bounds__(SCAN_T) void scan_single_kernel(const int* __restrict__ count, int n, int* __restrict__ offsets,
                                                             int* __restrict__ cursor, int heavy_t, int chunk, int* __restrict__ heavy_count,
                                                             int* __restrict__ heavy_key, int* __restrict__ heavy_chunk) {
  __shared__ int s_w[SCAN_T / 64];
  const int E = (n + SCAN_T - 1) / SCAN_T;               // <= SCAN1_E (host)
  const int base = threadIdx.x * E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int v[SCAN1_E], t = 0;
#pragma unroll
  for (int e = 0; e < SCAN1_E; ++e) { v[e] = (e < E && base + e < n) ? count[base + e] : 0; t += v[e]; }
  int incl = t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    int x = (lane < SCAN_T / 64) ? s_w[lane] : 0, inc = x;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane < SCAN_T / 64) s_w[lane] = inc - x;         // exclusive prefix of the wave totals
  }
  __syncthreads();
  int run = s_w[wave] + incl - t;
#pragma unroll
  for (int e = 0; e < SCAN1_E; ++e) {
    const int i = base + e;
    if (e < E && i < n) {
      offsets[i] = run;
      int slot = -1;
      if (heavy_count != nullptr && v[e] > heavy_t) {
        const int n_ch = (v[e] + chunk - 1) / chunk;
        slot = atomicAdd(heavy_count, n_ch);
        for (int c = 0; c < n_ch; ++c) { heavy_key[slot + c] = i; heavy_chunk[slot + c] = c; }
      }
      cursor[i] = slot;
    }
    run += v[e];
  }
}

// ---- workgroup-aggregated histogram / scatter of skewed keys -------------------------------------------------------
// Device-scope atomics on ONE address serialise at ~50-100 ns each on gfx950 (they resolve memory-side), so a popular
// item (Zipf catalogs: the top id holds ~9% of all targets) turns a plain per-element atomicAdd histogram into a
// multi-100-us serial chain.  Here a 1024-thread workgroup first counts its keys in an LDS hash table (LDS atomics
// serialise at a few cycles), then issues ONE global atomic per distinct key.  Keys equal to 0 (PAD) are skipped.
constexpr int AGG_T = 1024, AGG_SLOTS = 4096;
struct AggTable { int keys[AGG_SLOTS]; int cnt[AGG_SLOTS]; int base[AGG_SLOTS]; };

__device__ __forceinline__ void agg_clear(AggTable& t) {
  for (int i = threadIdx.x; i < AGG_SLOTS; i += AGG_T) { t.keys[i] = -1; t.cnt[i] = 0; }
  __syncthreads();
}
// returns the slot of `key`; rank = number of earlier inserts of the same key by this workgroup
__device__ __forceinline__ int agg_insert(AggTable& t, int key, int& rank) {
  unsigned h = ((unsigned)key * 0x9E3779B1u) >> 20;   // 12 bits
  for (;;) {
    const int prev = atomicCAS(&t.keys[h], -1, key);
    if (prev == -1 || prev == key) { rank = atomicAdd(&t.cnt[h], 1); return (int)h; }
    h = (h + 1) & (AGG_SLOTS - 1);
  }
}
// count[key] += (occurrences in this workgroup); rank[i * stride] = position of element i among ALL elements of its key
// (the old value of the global counter + the element's rank inside the workgroup) — the scatter that follows the scan
// is then a plain store to offsets[key] + rank, without a second round of atomics.
__global__ __launch_bounds__(AGG_T) void agg_rank_kernel(const long long* __restrict__ keys, int n, int* __restrict__ count,
                                                         int* __restrict__ rank, int stride) {
  __shared__ AggTable t;
  agg_clear(t);
  const int i = blockIdx.x * AGG_T + threadIdx.x;
  const int key = (i < n) ? (int)keys[i] : 0;
  int r = 0, slot = -1;
  if (key != 0) slot = agg_insert(t, key, r);
  __syncthreads();
  for (int s = threadIdx.x; s < AGG_SLOTS; s += AGG_T)
    if (t.keys[s] != -1) t.base[s] = atomicAdd(count + t.keys[s], t.cnt[s]);
  __syncthreads();
  if (slot >= 0) rank[(long long)i * stride] = t.base[slot] + r;
}

// offsets[i] = sum(count[0..i)); cursor[i] = first chunk slot of a popular key or -1; blocksum: ceil(n/4096) + 1 ints of scratch
inline size_t scan_blocks(size_t n) { return (n + SCAN_T * SCAN_E - 1) / (SCAN_T * SCAN_E); }
inline int exclusive_scan_counts(const int* count, int n, int* offsets, int* cursor, int* blocksum, hipStream_t stream,
                                 int heavy_t = 0, int chunk = 1, int* heavy_count = nullptr, int* heavy_key = nullptr,
                                 int* heavy_chunk = nullptr) {
  if (n <= SCAN1_E * SCAN_T) {
    scan_single_kernel<<<1, SCAN_T, 0, stream>>>(count, n, offsets, cursor, heavy_t, chunk, heavy_count, heavy_key, heavy_chunk);
    RT_CHECK_LAUNCH();
    return RT_OK;
  }
  const int nb = (int)scan_blocks((size_t)n);
  scan_local_kernel<<<nb, SCAN_T, 0, stream>>>(count, n, offsets, blocksum);
  RT_CHECK_LAUNCH();
  scan_blocksums_kernel<<<1, 1024, 0, stream>>>(blocksum, nb);
  RT_CHECK_LAUNCH();
  scan_add_kernel<<<(n + 255) / 256, 256, 0, stream>>>(offsets, cursor, blocksum, n, count, heavy_t, chunk, heavy_count, heavy_key,
                                                       heavy_chunk);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace
