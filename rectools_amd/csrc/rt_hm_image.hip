// Two-stage exact top-k, the image builder (the coarse pass is topk_stream_kernel<..., HM = true>, the exact pass topk_replay_kernel,
// both in rt_topk.hip):
//   rt_to_hm_rows   fp32 rows -> "hm image": every value x becomes the 32-bit word (h << 16) | m with h = the bf16 TRUNCATION of x and
//                   m = the bf16 truncation of x - h (exact subtraction), so x = h + m + l with |l| < 2^-15 |x|; plus the fp32 L2 norm of
//                   every row (the coarse pass's error bound is c |u| |v|).  Same row geometry as the source: the streaming kernel reads
//                   an image exactly as it reads fp32 rows.
// The reference scores every (user, item) pair in fp32 (rank_torch.py:194-208); here the fp32-input matrix instruction is spent only on
// candidates that can still be in the top-k.
#include "rt_common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ unsigned hm_word(float x) {
  const unsigned b = __float_as_uint(x);
  const unsigned h = b & 0xFFFF0000u;
  const float r = x - __uint_as_float(h);            // exact: the low 16 significand bits of x
  return h | (__float_as_uint(r) >> 16);
}

// one wave per row; a lane owns float4 columns lane*4 + 256*t (d <= 2048)
__global__ __launch_bounds__(256) void to_hm_rows_kernel(const float* __restrict__ src, long long src_stride,
                                                         const long long* __restrict__ rows, long long n_rows, int d, int normalize,
                                                         unsigned* __restrict__ dst, long long dst_stride, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (r >= n_rows) return;
  const long long sr = rows ? rows[r] : r;
  const float* x = src + sr * src_stride;
  float ss = 0.f;
  f32x4 v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = lane * 4 + 256 * t;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    v[t] = c < d ? *reinterpret_cast<const f32x4*>(x + c) : z;
    ss += v[t][0] * v[t][0] + v[t][1] * v[t][1] + v[t][2] * v[t][2] + v[t][3] * v[t][3];
  }
  const float nrm = sqrtf(wave_sum_f(ss));
  if (norms != nullptr && lane == 0) norms[r] = nrm;
  const float sc = (normalize & 1) ? 1.0f / fmaxf(nrm, 1e-8f) : 1.0f;      // cosine: the image of the unit row (the exact pass divides as rt_topk_score does)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = lane * 4 + 256 * t;
    if (c < d) {
      if (normalize & 2) {      // h-only image: one round-to-nearest bf16 per value, rows of d / 2 words
        unsigned short o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_bit_cast(unsigned short, (__bf16)(v[t][j] * sc));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(dst + r * dst_stride + c / 2) = u32x2{(unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16)};
      } else {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = hm_word(v[t][j] * sc);
        *reinterpret_cast<u32x4*>(dst + r * dst_stride + c) = w;
      }
    }
  }
}


// One-plane image, row-major (rows of d bf16 = d / 8 units of 16 bytes) -> FRAGMENT-MAJOR: unit u of row r goes to unit
// ((r / 32) (d / 16) + u / 2) 64 + 32 (u & 1) + r % 32 — the 64 units of (32-row group, k = 16 slot) are the 64 lanes' A (or B)
// operands of one v_mfma_f32_32x32x16_bf16 in lane order (lane = 32 half + row, half = the slot's upper 8 k).  Rows n_rows .. rows_pad
// are zero.  topk_coarse_frag_kernel reads an item fragment with one coalesced 1 KB load and copies a user tile linearly into the LDS.
__global__ __launch_bounds__(256) void one_plane_to_fragments_kernel(const u32x4* __restrict__ src, long long src_stride_units, long long n_rows,
                                                                     long long rows_pad, int n_units, u32x4* __restrict__ dst) {
  // one workgroup per 32-row group: coalesced reads along the rows, coalesced 1 KB writes per fragment (transposed through the LDS)
  __shared__ u32x4 tile[32][17];
  const long long g = blockIdx.x;
  const int tid = threadIdx.x;
  for (int u0 = 0; u0 < n_units; u0 += 16) {       // 16 units (= 8 slots) of the 32 rows at a time
    for (int i = tid; i < 32 * 16; i += 256) {
      const int r = i >> 4, u = i & 15;
      const long long row = g * 32 + r;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < n_rows && u0 + u < n_units) v = src[row * src_stride_units + u0 + u];
      tile[r][u] = v;
    }
    __syncthreads();
    for (int i = tid; i < 16 * 32; i += 256) {     // i = (unit u, row r): destination lane 32 (u & 1) + r of slot (u0 + u) / 2
      const int u = i >> 5, r = i & 31;
      if (u0 + u < n_units)
        dst[(g * (n_units >> 1) + ((u0 + u) >> 1)) * 64 + 32 * (u & 1) + r] = tile[r][u];
    }
    __syncthreads();
  }
  (void)rows_pad;
}

}  // namespace

extern "C" {

int rt_to_hm_rows(const float* src, int64_t src_stride, const int64_t* rows, int64_t n_rows, int32_t d, int32_t normalize, uint32_t* dst,
                  int64_t dst_stride, float* norms, hipStream_t stream) {
  (void)hipGetLastError();
  if (n_rows <= 0) return RT_OK;
  if (src == nullptr || dst == nullptr || d <= 0 || (d & 3) != 0 || d > 2048 || (src_stride & 3) != 0 || (dst_stride & 3) != 0 ||
      dst_stride < ((normalize & 2) ? d / 2 : d) || ((uintptr_t)src & 15) != 0 || ((uintptr_t)dst & 15) != 0 || n_rows > 0x7FFFFFFFLL * 4)
    return RT_ERR_INVALID_ARG;
  to_hm_rows_kernel<<<(unsigned)((n_rows + 3) / 4), 256, 0, stream>>>(src, src_stride, reinterpret_cast<const long long*>(rows), n_rows, d,
                                                                      normalize, dst, dst_stride, norms);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// src: a one-plane image (rt_to_hm_rows mode 2 / 3; rows of src_stride_words 32-bit words, d bf16 values each), dst: its fragment-major
// form for rt_topk_score_two_stage(h_only = 2): rows_pad (a multiple of 128 >= n_rows) x d bf16 = rows_pad * d * 2 bytes, 16-byte aligned;
// d % 16 == 0.
int rt_one_plane_to_fragments(const uint32_t* src, int64_t src_stride_words, int64_t n_rows, int32_t d, uint32_t* dst, int64_t rows_pad,
                              hipStream_t stream) {
  (void)hipGetLastError();
  if (n_rows < 0 || d <= 0 || (d & 15) != 0 || rows_pad < n_rows || (rows_pad & 127) != 0 || (src_stride_words & 3) != 0 || src_stride_words * 2 < d)
    return RT_ERR_INVALID_ARG;
  if (rows_pad == 0) return RT_OK;
  if (src == nullptr || dst == nullptr || ((uintptr_t)src & 15) != 0 || ((uintptr_t)dst & 15) != 0) return RT_ERR_INVALID_ARG;
  one_plane_to_fragments_kernel<<<(unsigned)(rows_pad / 32), 256, 0, stream>>>(reinterpret_cast<const u32x4*>(src), src_stride_words / 4, n_rows,
                                                                              rows_pad, d / 8, reinterpret_cast<u32x4*>(dst));
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // extern "C"
