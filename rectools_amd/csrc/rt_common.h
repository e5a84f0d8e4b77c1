// Shared device/host helpers for librectools_hip.so (gfx950 / CDNA4 only: 64-wide wavefronts, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_OK 0
#define RT_ERR_INVALID_ARG 1
#define RT_ERR_WORKSPACE 2
#define RT_ERR_LAUNCH 3
#define RT_ERR_UNSUPPORTED 4

#define RT_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));   // one 16-byte operand of v_mfma_f32_32x32x16_bf16

// Last HIP failure seen by this library (file:line + hipGetErrorString), readable through rt_last_error().
void rt_set_last_error(const char* file, int line, hipError_t e);

#define RT_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) {                       \
      rt_set_last_error(__FILE__, __LINE__, e__);  \
      return RT_ERR_LAUNCH;                        \
    }                                              \
  } while (0)

#define RT_CHECK_HIP(call)                         \
  do {                                             \
    hipError_t e__ = (call);                       \
    if (e__ != hipSuccess) {                       \
      rt_set_last_error(__FILE__, __LINE__, e__);  \
      return RT_ERR_LAUNCH;                        \
    }                                              \
  } while (0)

// ---- wave-level reductions (64 lanes) -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- order-preserving float <-> uint32 key (for atomicMax on floats) ---------------------------------
__device__ __forceinline__ unsigned f32_to_key(float f) {
  unsigned u = __float_as_uint(f);
  unsigned mask = (u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
  return u ^ mask;
}
__device__ __forceinline__ float key_to_f32(unsigned k) {
  unsigned mask = (k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(k ^ mask);
}

// ---- counter-based RNG (Philox4x32-10), used for dropout masks and negative sampling -----------------
// One call yields 4 x 32 random bits for (seed, offset, subsequence).  Stateless, so backward kernels
// regenerate exactly the forward mask instead of storing it.
__device__ __forceinline__ uint4 philox4x32(unsigned long long seed, unsigned long long subseq,
                                            unsigned long long offset) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)offset, c1 = (unsigned)(offset >> 32);
  unsigned c2 = (unsigned)subseq, c3 = (unsigned)(subseq >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    unsigned n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u32_to_unit(unsigned x) {  // [0,1)
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---- element-dropout masks (row kernels, GEMM epilogues): hash of a counter, 16 bits per element ----------
// Philox4x32-10 costs 40 quarter-rate 32-bit multiplies per four elements (~840 issue cycles per wave call) — more than the matrix
// work of a fused GEMM epilogue that has to regenerate a mask.  A dropout mask needs decorrelation, not cryptographic strength:
// two avalanche finalisers (murmur3's fmix32, lowbias32) over (counter, seed, stream) give 64 bits per float4 group for five
// multiplies; an element is kept when its 16-bit field >= p * 65536 (the attention kernels' construction, rt_varlen.h).  Forward
// and backward kernels regenerate the same mask from (seed, stream, index of the float4 group).  Philox stays where the draw IS
// the product (negative sampling, rt_collate.hip).
__device__ __forceinline__ unsigned rt_fmix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned rt_lowbias32(unsigned x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint2 rt_drop_bits(unsigned long long seed, unsigned long long stream, unsigned long long idx4) {
  const unsigned k0 = (unsigned)seed ^ ((unsigned)stream * 0x9E3779B1u) ^ (unsigned)(stream >> 32);
  const unsigned k1 = (unsigned)(seed >> 32) + ((unsigned)stream ^ 0xC2B2AE3Du) * 0x85EBCA77u;
  const unsigned lo = (unsigned)idx4, hi = (unsigned)(idx4 >> 32);
  const unsigned a = rt_fmix32((lo * 0x9E3779B1u) ^ k0 ^ (hi * 0x27D4EB2Fu));
  const unsigned b = rt_lowbias32(a + k1 + lo);
  return make_uint2(a, b);
}
__device__ __forceinline__ unsigned rt_drop_thr16(float p) { return (unsigned)(p * 65536.0f); }
// v[j] kept (and scaled by 1 / (1 - p)) or zeroed, j = 0..3: element j of float4 group idx4
__device__ __forceinline__ f32x4 rt_drop4(f32x4 v, unsigned long long seed, unsigned long long stream, unsigned long long idx4, float p,
                                           float inv_keep) {
  const uint2 r = rt_drop_bits(seed, stream, idx4);
  const unsigned thr = rt_drop_thr16(p);
  v[0] = ((r.x & 0xFFFFu) >= thr) ? v[0] * inv_keep : 0.f;
  v[1] = ((r.x >> 16) >= thr) ? v[1] * inv_keep : 0.f;
  v[2] = ((r.y & 0xFFFFu) >= thr) ? v[2] * inv_keep : 0.f;
  v[3] = ((r.y >> 16) >= thr) ? v[3] * inv_keep : 0.f;
  return v;
}

static inline int rt_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
