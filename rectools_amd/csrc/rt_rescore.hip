// Two-stage exact top-k, the parts around the coarse pass (rt_topk.hip: topk_stream_kernel<..., BF = true>):
//   rt_to_bf16_rows   fp32 rows -> bf16 image (round to nearest even), optionally L2-normalised, plus the fp32 row norms
//   rt_topk_rescore   exact fp32 scores of the few candidates the coarse pass kept, in the arithmetic of the exact kernel
// The reference scores every (user, item) pair in fp32 (rank_torch.py:194-208); here the fp32 arithmetic is spent only on
// candidates that can still be in the top-k (|bf16 score - fp32 score| <= c |u| |v|, c = 2^-8 + ...).
#include "rt_common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per row; a lane owns float4 columns lane*4 + 256*t (d <= 2048)
__global__ __launch_bounds__(256) void to_bf16_rows_kernel(const float* __restrict__ src, long long src_stride,
                                                           const long long* __restrict__ rows, int n_rows, int d, int normalize,
                                                           unsigned short* __restrict__ dst, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= n_rows) return;
  const long long sr = rows ? rows[r] : (long long)r;
  const float* x = src + sr * src_stride;
  f32x4 v[8];
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = lane * 4 + 256 * t;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    v[t] = c < d ? *reinterpret_cast<const f32x4*>(x + c) : z;
    ss += v[t][0] * v[t][0] + v[t][1] * v[t][1] + v[t][2] * v[t][2] + v[t][3] * v[t][3];
  }
  const float nrm = sqrtf(wave_sum_f(ss));
  if (norms != nullptr && lane == 0) norms[r] = nrm;
  const float sc = normalize ? 1.0f / fmaxf(nrm, 1e-8f) : 1.0f;   // the exact kernel's cosine denominator (rt_topk.hip)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = lane * 4 + 256 * t;
    if (c < d) {
      unsigned short o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)(v[t][j] * sc);        // v_cvt_pk_bf16_f32: round to nearest even
        o[j] = __builtin_bit_cast(unsigned short, h);
      }
      uint2 pk;
      pk.x = (unsigned)o[0] | ((unsigned)o[1] << 16);
      pk.y = (unsigned)o[2] | ((unsigned)o[3] << 16);
      *reinterpret_cast<uint2*>(dst + (long long)r * d + c) = pk;
    }
  }
}

enum { RS_DOT = 0, RS_COSINE = 1 };

// one wave per (user, candidate): exact fp32 dot product (and norms for cosine); entries past cand_counts[u] get -inf
__global__ __launch_bounds__(256) void topk_rescore_kernel(const float* __restrict__ users, long long user_stride,
                                                           const long long* __restrict__ user_rows, int n_users,
                                                           const float* __restrict__ items, long long item_stride, int d,
                                                           int distance, const long long* __restrict__ cand_ids,
                                                           const int* __restrict__ cand_counts, int kc,
                                                           float* __restrict__ out_scores) {
  const int lane = threadIdx.x & 63;
  const long long pair = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (pair >= (long long)n_users * kc) return;
  const int u = (int)(pair / kc), j = (int)(pair % kc);
  if (j >= cand_counts[u]) {
    if (lane == 0) out_scores[pair] = -__builtin_inff();
    return;
  }
  const float* ur = users + (user_rows ? user_rows[u] : (long long)u) * user_stride;
  const float* ir = items + cand_ids[pair] * item_stride;
  float dot = 0.f, uu = 0.f, vv = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ur + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ir + c);
    dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    uu += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    vv += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
  }
  dot = wave_sum_f(dot);
  if (distance == RS_COSINE) {
    uu = wave_sum_f(uu); vv = wave_sum_f(vv);
    dot = dot * (1.0f / fmaxf(sqrtf(uu), 1e-8f)) * (1.0f / fmaxf(sqrtf(vv), 1e-8f));
  }
  if (lane == 0) out_scores[pair] = dot;
}

}  // namespace

extern "C" {

int rt_to_bf16_rows(const float* src, int64_t src_stride, const int64_t* rows, int32_t n_rows, int32_t d, int32_t normalize,
                    uint16_t* dst, float* norms, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_rows <= 0) return RT_OK;
  if (src == nullptr || dst == nullptr || d <= 0 || (d & 3) != 0 || d > 2048 || (src_stride & 3) != 0 || ((uintptr_t)src & 15) != 0 ||
      ((uintptr_t)dst & 7) != 0)
    return RT_ERR_INVALID_ARG;
  to_bf16_rows_kernel<<<(n_rows + 3) / 4, 256, 0, stream>>>(src, src_stride, reinterpret_cast<const long long*>(rows), n_rows, d,
                                                            normalize, dst, norms);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

int rt_topk_rescore(const float* users, int64_t user_stride, const int64_t* user_rows, int32_t n_users, const float* items,
                    int64_t item_stride, int32_t d, int32_t distance, const int64_t* cand_ids, const int32_t* cand_counts,
                    int32_t kc, float* out_scores, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_users <= 0 || kc <= 0) return RT_OK;
  if (users == nullptr || items == nullptr || cand_ids == nullptr || cand_counts == nullptr || out_scores == nullptr || d <= 0 ||
      (d & 3) != 0 || (user_stride & 3) != 0 || (item_stride & 3) != 0 || ((uintptr_t)users & 15) != 0 || ((uintptr_t)items & 15) != 0)
    return RT_ERR_INVALID_ARG;
  if (distance != RS_DOT && distance != RS_COSINE) return RT_ERR_UNSUPPORTED;
  const long long pairs = (long long)n_users * kc;
  topk_rescore_kernel<<<(unsigned)((pairs + 3) / 4), 256, 0, stream>>>(users, user_stride, reinterpret_cast<const long long*>(user_rows),
                                                                        n_users, items, item_stride, d, distance,
                                                                        reinterpret_cast<const long long*>(cand_ids), cand_counts, kc,
                                                                        out_scores);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
