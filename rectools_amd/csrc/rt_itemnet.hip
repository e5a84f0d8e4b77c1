// Item-row producer for feature-aware item nets (all fp32, HBM-bound, one wave per row).
// Reference call sites:
//   K1  SumOfEmbeddingsConstructor.forward / get_all_embeddings      item_net.py:361-368,463-482
//       IdEmbeddingsItemNet.forward                                  item_net.py:266-281
//       CatFeaturesItemNet.forward (nn.EmbeddingBag mode="sum" over the item's category (feature, value) ids, then
//       nn.Dropout on the bag sums)                                  item_net.py:101-132
// The reference gathers `emb_bag_inputs[offsets[i] : offsets[i] + input_lengths[i]]` for the whole catalog on every
// forward, runs EmbeddingBag, stacks the blocks and sums them.  Here the catalog matrix is produced in ONE pass:
//   table[i,:] = ids_emb[i,:] + drop( sum_j cat_emb[emb_bag_inputs[offsets[i] + j], :] )
// and the gradient of cat_emb is a transposed-CSR row reduction (the item -> feature structure is static, so its
// transpose is built once on the host): no float atomics, fixed summation order.  Popular feature values
// ("genre = drama" tags thousands of items) are cut into chunks by the host; each chunk is reduced by its own wave
// into a slab row, then one wave per feature value adds that value's slab rows in order.
#include "rt_common.h"

namespace {

__device__ __forceinline__ f32x4 bag_drop4(f32x4 v, unsigned long long seed, unsigned long long stream,
                                            unsigned long long idx4, float p, float inv_keep) {
  return rt_drop4(v, seed, stream, idx4, p, inv_keep);
}

// One wave per catalog row; a lane owns float4 columns lane*4 + 256*t.  The row's feature ids are wave-uniform
// (scalar loads), the feature table is small and stays in L2, so the pass streams ids_emb in and the table out.
__global__ __launch_bounds__(256) void bag_sum_fwd_kernel(const float* __restrict__ ids_emb, const float* __restrict__ cat_emb,
                                                          const long long* __restrict__ inputs,
                                                          const long long* __restrict__ offsets,
                                                          const long long* __restrict__ lengths, int V, int d, float p,
                                                          unsigned long long seed, unsigned long long stream,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (i >= V) return;
  const long long s = offsets[i];
  const int n = (int)lengths[i];
  const long long* row = inputs + s;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 4 <= n; j += 4) {   // four independent row loads in flight
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(cat_emb + row[j] * (long long)d + c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(cat_emb + row[j + 1] * (long long)d + c);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(cat_emb + row[j + 2] * (long long)d + c);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(cat_emb + row[j + 3] * (long long)d + c);
      acc += a0; acc += a1; acc += a2; acc += a3;    // left-to-right, the order EmbeddingBag's sum uses
    }
    for (; j < n; ++j) acc += *reinterpret_cast<const f32x4*>(cat_emb + row[j] * (long long)d + c);
    if (p > 0.f) acc = bag_drop4(acc, seed, stream, ((unsigned long long)i * d + c) >> 2, p, inv_keep);
    if (ids_emb != nullptr) acc += *reinterpret_cast<const f32x4*>(ids_emb + (long long)i * d + c);
    *reinterpret_cast<f32x4*>(out + (long long)i * d + c) = acc;
  }
}

// One wave per chunk of the transposed structure: slab[chunk,:] = sum_{e in chunk} drop(dE[t_items[e], :]).
__global__ __launch_bounds__(256) void bag_sum_bwd_chunks_kernel(const float* __restrict__ d_out,
                                                                 const long long* __restrict__ t_items,
                                                                 const long long* __restrict__ chunk_ptr, int n_chunks, int d,
                                                                 float p, unsigned long long seed, unsigned long long stream,
                                                                 float* __restrict__ slab) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (ch >= n_chunks) return;
  const long long s = chunk_ptr[ch], e = chunk_ptr[ch + 1];
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long j = s;
    for (; j + 8 <= e; j += 8) {   // eight gradient rows in flight per lane
      long long it[8];
      f32x4 g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) it[u] = t_items[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) g[u] = *reinterpret_cast<const f32x4*>(d_out + it[u] * (long long)d + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (p > 0.f) g[u] = bag_drop4(g[u], seed, stream, ((unsigned long long)it[u] * d + c) >> 2, p, inv_keep);
        acc += g[u];
      }
    }
    for (; j < e; ++j) {
      const long long it = t_items[j];
      f32x4 g = *reinterpret_cast<const f32x4*>(d_out + it * (long long)d + c);
      if (p > 0.f) g = bag_drop4(g, seed, stream, ((unsigned long long)it * d + c) >> 2, p, inv_keep);
      acc += g;
    }
    *reinterpret_cast<f32x4*>(slab + (long long)ch * d + c) = acc;
  }
}

// One wave per feature value: d_cat[f,:] = slab rows feat_chunk_ptr[f] .. feat_chunk_ptr[f+1], added in order.
__global__ __launch_bounds__(256) void bag_sum_bwd_combine_kernel(const float* __restrict__ slab,
                                                                  const long long* __restrict__ feat_chunk_ptr, int F, int d,
                                                                  float* __restrict__ d_cat) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (f >= F) return;
  const long long s = feat_chunk_ptr[f], e = feat_chunk_ptr[f + 1];
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long j = s;
    for (; j + 4 <= e; j += 4) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + j * d + c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + (j + 1) * d + c);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(slab + (j + 2) * d + c);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(slab + (j + 3) * d + c);
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; j < e; ++j) acc += *reinterpret_cast<const f32x4*>(slab + j * d + c);
    *reinterpret_cast<f32x4*>(d_cat + (long long)f * d + c) = acc;
  }
}

}  // namespace

extern "C" {

int rt_bag_sum_fwd(const float* ids_emb, const float* cat_emb, const int64_t* emb_bag_inputs, const int64_t* offsets,
                   const int64_t* input_lengths, int32_t V, int32_t d, float p, uint64_t seed, uint64_t stream_id,
                   float* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (V <= 0) return RT_OK;
  if ((d & 3) != 0 || d <= 0 || cat_emb == nullptr || offsets == nullptr || input_lengths == nullptr || out == nullptr ||
      p < 0.f || p >= 1.f)
    return RT_ERR_INVALID_ARG;
  bag_sum_fwd_kernel<<<(V + 3) / 4, 256, 0, stream>>>(ids_emb, cat_emb, reinterpret_cast<const long long*>(emb_bag_inputs),
                                                       reinterpret_cast<const long long*>(offsets),
                                                       reinterpret_cast<const long long*>(input_lengths), V, d, p, seed,
                                                       stream_id, out);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

size_t rt_bag_sum_bwd_workspace_bytes(int64_t n_chunks, int32_t d) { return (size_t)(n_chunks > 0 ? n_chunks : 0) * (size_t)d * 4; }

int rt_bag_sum_bwd(const float* d_out, const int64_t* t_items, const int64_t* chunk_ptr, int64_t n_chunks,
                   const int64_t* feat_chunk_ptr, int32_t F, int32_t d, float p, uint64_t seed, uint64_t stream_id,
                   float* d_cat, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  (void)hipGetLastError();
  if (F <= 0) return RT_OK;
  if ((d & 3) != 0 || d <= 0 || d_out == nullptr || d_cat == nullptr || feat_chunk_ptr == nullptr || n_chunks < 0 ||
      n_chunks > 0x7fffffffLL || p < 0.f || p >= 1.f || (n_chunks > 0 && (t_items == nullptr || chunk_ptr == nullptr)))
    return RT_ERR_INVALID_ARG;
  if (n_chunks > 0 && (workspace == nullptr || workspace_bytes < rt_bag_sum_bwd_workspace_bytes(n_chunks, d)))
    return RT_ERR_WORKSPACE;
  float* slab = reinterpret_cast<float*>(workspace);
  if (n_chunks > 0) {
    bag_sum_bwd_chunks_kernel<<<(unsigned)((n_chunks + 3) / 4), 256, 0, stream>>>(
        d_out, reinterpret_cast<const long long*>(t_items), reinterpret_cast<const long long*>(chunk_ptr), (int)n_chunks, d, p,
        seed, stream_id, slab);
    RT_CHECK_LAUNCH();
  }
  bag_sum_bwd_combine_kernel<<<(F + 3) / 4, 256, 0, stream>>>(slab, reinterpret_cast<const long long*>(feat_chunk_ptr), F, d,
                                                               d_cat);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
