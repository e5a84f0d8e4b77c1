// Shared by the bf16-plane attention kernels (rt_attention_v2.hip: whole-session images, HSTU's K6v2; rt_attention_v3.hip: streamed
// chunks): the exact three-way bf16 split, the six-term product, the LDS image layout with its swizzle, the two fragment readers
// (rows as MFMA rows / rows as the reduction index through the transpose read) and the staging loops.  See rt_attention_v2.hip's header
// for the design; DESIGN.md K4v2 / K4v3.
#pragma once
#include "rt_varlen.h"

namespace rt_planes {
using namespace rt_varlen;

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define RT_LDS __attribute__((address_space(3)))

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

struct P3 { bf16x8 h, m, l; };   // the three bf16 planes of 8 fp32 values (one MFMA operand each)

// Ablation builds only (-DRT_ABLATION_BUILD, RT_V2_ABLATE bits, scripts/attn_ablate.py): parts of the softmax kernels switched off at run
// time so that their cost can be read off the kernel time.  1 no staging (the LDS holds garbage), 2 staging only (no tile loop), 4 no
// softmax / mask / dropout arithmetic, 8 no three-way split of the probabilities, 16 no products (LDS fragment reads + MFMA), 64 no owner-row
// loads, 128 no result stores; 256 / 512 (host side): skip the dQ / the dK,dV launch.  Results are meaningless under any bit.  A product
// build compiles every test to `false`.
#ifdef RT_ABLATION_BUILD
#define RT_ABL(a, bit) (((a).ablate & (bit)) != 0)
#define RT_ABLV(a) ((a).ablate)
#else
#define RT_ABL(a, bit) false
#define RT_ABLV(a) 0
#endif

// (a, b) -> the packed bf16 pairs {b, a} of the three planes.  Truncation of the top half IS the bf16; both subtractions are exact.
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
  const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
  const float la = ra - __uint_as_float(va & 0xFFFF0000u), lb = rb - __uint_as_float(vb & 0xFFFF0000u);
  h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
  m = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}
__device__ __forceinline__ P3 split8(const float (&x)[8]) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) { unsigned h, m, l; split2(x[2 * q], x[2 * q + 1], h, m, l); ph[q] = h; pm[q] = m; pl[q] = l; }
  P3 r;
  r.h = __builtin_bit_cast(bf16x8, ph); r.m = __builtin_bit_cast(bf16x8, pm); r.l = __builtin_bit_cast(bf16x8, pl);
  return r;
}

#ifdef RT_ABLATION_BUILD
__device__ __forceinline__ P3 fake_split8(const float (&x)[8]) {   // bits 8: the values' bits as planes, no arithmetic
  u32x4 a{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
  u32x4 b{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
  P3 r;
  r.h = __builtin_bit_cast(bf16x8, a); r.m = __builtin_bit_cast(bf16x8, b); r.l = __builtin_bit_cast(bf16x8, a);
  return r;
}
#define RT_SPLIT8(a, x) (RT_ABL(a, 8) ? fake_split8(x) : split8(x))
#else
#define RT_SPLIT8(a, x) split8(x)
#endif

// six-term product: acc += A * B for fp32-accurate A, B given as planes (smallest terms first)
__device__ __forceinline__ f32x4 mfma6(const P3& A, const P3& B, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.l, B.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.h, B.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.m, B.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.m, B.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.h, B.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.h, B.h, acc, 0, 0, 0);
  return acc;
}

template <int HD> struct Lay {
  static constexpr int ROWB = HD * 2;        // bytes of one plane row
  static constexpr int ROW3 = 3 * ROWB;      // bytes of one image row (h | m | l)
  static constexpr int NS = HD / 32;         // MFMA k-steps over the head dimension
  static constexpr int NCB = HD / 16;        // 16-column blocks of the head dimension
  // XOR mask of the 16-byte unit index of row r (scripts/attn/swizzle_search.py)
  // (hd 128, 16 units per plane row: unit bit b + 1 <- row bit b, b = 0 .. 2 — `swizzle_search.py 128` finds it conflict free)
  __device__ static __forceinline__ unsigned swz(int r) {
    return HD == 128 ? (unsigned)((r & 7) << 1) : HD == 64 ? (unsigned)(r & 6) : (unsigned)((r >> 1) & 2);
  }
  static size_t image_bytes(int max_len) { return (size_t)(max_len + 1) * ROW3; }   // + the zero row
};

// Stage rows [0, n) of a [*, ld] fp32 matrix (columns [0, HD) of this head) into an LDS image: value * scale, split into planes.
// Row n of the image is zero-filled.
template <int HD>
__device__ __forceinline__ void stage_image(const float* __restrict__ src, long long ld, int n, float scale, unsigned char* img, int tid,
                                            int nthreads) {
  using L = Lay<HD>;
  constexpr int C4 = HD / 4;
  const int total = n * C4;
#pragma unroll 4
  for (int idx = tid; idx < total; idx += nthreads) {
    const int r = idx / C4, c4 = idx % C4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + c4 * 4) * scale;
    u32x2 h, m, l;
    { unsigned a, b, c; split2(x[0], x[1], a, b, c); h[0] = a; m[0] = b; l[0] = c; }
    { unsigned a, b, c; split2(x[2], x[3], a, b, c); h[1] = a; m[1] = b; l[1] = c; }
    unsigned char* p = img + r * L::ROW3 + (((unsigned)c4 ^ (L::swz(r) << 1)) << 3);
    *reinterpret_cast<u32x2*>(p) = h;
    *reinterpret_cast<u32x2*>(p + L::ROWB) = m;
    *reinterpret_cast<u32x2*>(p + 2 * L::ROWB) = l;
  }
  for (int w = tid; w < L::ROW3 / 8; w += nthreads) *reinterpret_cast<u32x2*>(img + n * L::ROW3 + w * 8) = u32x2{0u, 0u};
}

// Two images in ONE pass: the loads of both (U float4 each per thread and round) are issued before any split arithmetic, so a round costs
// one memory round trip instead of two (stage_image twice: ~4 dependent round trips for a 200-row session, the prologue of a workgroup
// that sits alone on its CU).  U = 8: a 200-row session's 13 float4 per thread (512 threads, hd 64) and a 192-row chunk of the HSTU
// kernels are ONE round — 64 registers that are dead before the tile loop starts.  Same LDS contents as two stage_image calls.
template <int HD>
__device__ __forceinline__ void stage_images2(const float* __restrict__ srcA, long long ldA, float scaleA, unsigned char* imgA,
                                              const float* __restrict__ srcB, long long ldB, float scaleB, unsigned char* imgB,
                                              int n, int tid, int nthreads) {
  using L = Lay<HD>;
  constexpr int C4 = HD / 4, U = 8;
  const int total = n * C4;
  auto put = [&](unsigned char* img, int idx, const f32x4& x) {
    const int r = idx / C4, c4 = idx % C4;
    u32x2 h, m, l;
    { unsigned a, b, c; split2(x[0], x[1], a, b, c); h[0] = a; m[0] = b; l[0] = c; }
    { unsigned a, b, c; split2(x[2], x[3], a, b, c); h[1] = a; m[1] = b; l[1] = c; }
    unsigned char* p = img + r * L::ROW3 + (((unsigned)c4 ^ (L::swz(r) << 1)) << 3);
    *reinterpret_cast<u32x2*>(p) = h;
    *reinterpret_cast<u32x2*>(p + L::ROWB) = m;
    *reinterpret_cast<u32x2*>(p + 2 * L::ROWB) = l;
  };
#pragma unroll 1
  for (int idx0 = tid; idx0 < total; idx0 += U * nthreads) {
    f32x4 xa[U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = idx0 + u * nthreads;
      if (idx < total) {
        const int r = idx / C4, c4 = idx % C4;
        xa[u] = *reinterpret_cast<const f32x4*>(srcA + (long long)r * ldA + c4 * 4);
        xb[u] = *reinterpret_cast<const f32x4*>(srcB + (long long)r * ldB + c4 * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = idx0 + u * nthreads;
      if (idx < total) { put(imgA, idx, xa[u] * scaleA); put(imgB, idx, xb[u] * scaleB); }
    }
  }
  for (int w = tid; w < L::ROW3 / 8; w += nthreads) {
    *reinterpret_cast<u32x2*>(imgA + n * L::ROW3 + w * 8) = u32x2{0u, 0u};
    *reinterpret_cast<u32x2*>(imgB + n * L::ROW3 + w * 8) = u32x2{0u, 0u};
  }
}

// 8 fp32 values of one row for the reduction slots of lane group g: columns 32 s + 8 g + (0..7), times scale, as planes
template <int HD>
__device__ __forceinline__ void load_owner_planes(const float* __restrict__ row, int g, float scale, P3 (&out)[HD / 32]) {
#pragma unroll
  for (int s = 0; s < HD / 32; ++s) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g), x1 = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g + 4);
    const float x[8] = {x0[0] * scale, x0[1] * scale, x0[2] * scale, x0[3] * scale, x1[0] * scale, x1[1] * scale, x1[2] * scale, x1[3] * scale};
    out[s] = split8(x);
  }
}

// The same in two halves: the loads (issued early, under other memory traffic) and the split (when the values are needed)
template <int HD>
__device__ __forceinline__ void load_owner_raw(const float* __restrict__ row, int g, f32x4 (&raw)[HD / 16]) {
#pragma unroll
  for (int s = 0; s < HD / 32; ++s) {
    raw[2 * s] = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g);
    raw[2 * s + 1] = *reinterpret_cast<const f32x4*>(row + 32 * s + 8 * g + 4);
  }
}
template <int HD>
__device__ __forceinline__ void split_owner_raw(const f32x4 (&raw)[HD / 16], float scale, P3 (&out)[HD / 32]) {
#pragma unroll
  for (int s = 0; s < HD / 32; ++s) {
    const f32x4 x0 = raw[2 * s], x1 = raw[2 * s + 1];
    const float x[8] = {x0[0] * scale, x0[1] * scale, x0[2] * scale, x0[3] * scale, x1[0] * scale, x1[1] * scale, x1[2] * scale, x1[3] * scale};
    out[s] = split8(x);
  }
}

// acc[kb][r] = sum_c img[t0 + 16 kb + 4 g + r][c] * owner[lane & 15][c]  (kb = 0, 1; r = 0..3): the partner rows are the MFMA rows
template <int HD>
__device__ __forceinline__ void rows_times_owner(const unsigned char* img, int t0, int n, const P3 (&own)[HD / 32], int i, int g,
                                                 f32x4 (&acc)[2], int abl = 0) {
  using L = Lay<HD>;
  if (abl & 16) { acc[0] = acc[1] = f32x4{0.5f, 0.25f, 0.125f, 0.0625f}; return; }
  f32x4 part[2][HD / 32];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int r = min(t0 + 16 * kb + i, n);
    const unsigned char* base = img + r * L::ROW3;
    const unsigned x = L::swz(r);
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) {
      const unsigned char* p = base + ((((unsigned)(4 * s + g)) ^ x) << 4);
      P3 A;
      A.h = *reinterpret_cast<const bf16x8*>(p);
      A.m = *reinterpret_cast<const bf16x8*>(p + L::ROWB);
      A.l = *reinterpret_cast<const bf16x8*>(p + 2 * L::ROWB);
      part[kb][s] = mfma6(A, own[s], f32x4{0.f, 0.f, 0.f, 0.f});
    }
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    acc[kb] = part[kb][0];
#pragma unroll
    for (int s = 1; s < HD / 32; ++s) acc[kb] += part[kb][s];
  }
}

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((RT_LDS s16x4*)(p));
}

// acc[cb][r] += sum over the 32 partner rows of img[row][16 cb + 4 g + r] * slots(row), where the lane's 8 slots are the rows
// t0 + 16 kb + 4 g + e (slot 4 kb + e): the partner rows are the REDUCTION index (transpose read)
template <int HD>
__device__ __forceinline__ void cols_times_slots(const unsigned char* img, int t0, int n, const P3& slots, int i, int g,
                                                 f32x4 (&acc)[HD / 16], int abl = 0) {
  using L = Lay<HD>;
  if (abl & 16) { acc[0] += f32x4{1.f, 1.f, 1.f, 1.f} * (float)slots.h[0]; return; }
  const int j = i >> 2, t = i & 3;
  const unsigned char* rb[2]; unsigned xs[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = min(t0 + 16 * kb + 4 * g + j, n);
    rb[kb] = img + k * L::ROW3;
    xs[kb] = L::swz(k) << 1;
  }
  P3 A[HD / 16];
#pragma unroll
  for (int cb = 0; cb < HD / 16; ++cb) {
    const unsigned c = (unsigned)(4 * cb + t);
    const unsigned char* p0 = rb[0] + ((c ^ xs[0]) << 3);
    const unsigned char* p1 = rb[1] + ((c ^ xs[1]) << 3);
    s16x8 vh, vm, vl;
    { const s16x4 a = tr_read(p0), b = tr_read(p1); vh = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
    { const s16x4 a = tr_read(p0 + L::ROWB), b = tr_read(p1 + L::ROWB); vm = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
    { const s16x4 a = tr_read(p0 + 2 * L::ROWB), b = tr_read(p1 + 2 * L::ROWB); vl = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
    A[cb].h = __builtin_bit_cast(bf16x8, vh); A[cb].m = __builtin_bit_cast(bf16x8, vm); A[cb].l = __builtin_bit_cast(bf16x8, vl);
  }
  // the HD / 16 accumulators take turns inside a term: no back-to-back dependent MFMAs
#define RT_V2_TERM(PA, PB)                                                                                   \
  _Pragma("unroll") for (int cb = 0; cb < HD / 16; ++cb)                                                     \
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb].PA, slots.PB, acc[cb], 0, 0, 0);
  RT_V2_TERM(l, h) RT_V2_TERM(h, l) RT_V2_TERM(m, m) RT_V2_TERM(m, h) RT_V2_TERM(h, m) RT_V2_TERM(h, h)
#undef RT_V2_TERM
}

__device__ __forceinline__ float quad_max(float v) {   // over the 4 lanes (lane & 15 equal) that share an owner row
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// o-th heaviest owner tile -> wave, in a zigzag of period 2 NW: wave w takes o = w, 2 NW - 1 - w, 2 NW + w, ...
// HEAVY_LAST: the tile with the largest index is the heaviest (query tiles: they see every earlier key); else tile 0 is (key tiles).
template <int NW, bool HEAVY_LAST, typename F>
__device__ __forceinline__ void for_my_tiles(int wave, int n_tiles, F&& body) {
  for (int base = 0; base < n_tiles; base += 2 * NW) {
    const int o1 = base + wave, o2 = base + 2 * NW - 1 - wave;
    if (o1 < n_tiles) body(HEAVY_LAST ? n_tiles - 1 - o1 : o1);
    if (o2 < n_tiles) body(HEAVY_LAST ? n_tiles - 1 - o2 : o2);
  }
}

// How many of a query's n_pad pad keys (key numbers n .. n + n_pad - 1) survive the attention dropout, counted by the FOUR lanes that
// share the query: the mask is one 32-bit mix per PAIR of adjacent keys (rt_varlen.h), so the lanes walk pairs — one mix per two keys
// instead of one per key (the round-5 loops re-mixed every pair twice).  Every lane of the four receives the total.
__device__ __forceinline__ int pads_kept_quad(unsigned long long seed, unsigned bh, unsigned q, int n, int n_pad, unsigned thr16, int g) {
  int kept = 0;
  const int first = n >> 1, last = (n + n_pad - 1) >> 1;            // key pairs that hold pad keys
  for (int pr = first + g; pr <= last; pr += 4) {
    const unsigned hsh = drop_hash(seed, bh, q, (unsigned)pr);
    const int k0 = 2 * pr, k1 = 2 * pr + 1;
    kept += (k0 >= n && k0 < n + n_pad && (hsh & 0xFFFFu) >= thr16) ? 1 : 0;
    kept += (k1 >= n && k1 < n + n_pad && (hsh >> 16) >= thr16) ? 1 : 0;
  }
  kept += __shfl_xor(kept, 16, 64);
  kept += __shfl_xor(kept, 32, 64);
  return kept;
}

// ---- HSTU's pointwise attention (hstu.py:84-128, 270-288): bucket search, silu, the bias gradients' register reductions — shared by K6v2
// (rt_attention_v2.hip) and K6v3 (rt_attention_v3.hip) ---------------------------------------------------------------------------------
constexpr int NBUCK = 147;       // time buckets (every bucket an int64 difference can reach), as rt_attention.hip

// rt_attention.hip's bucket search, bit for bit (largest b with thr[b] <= |dt|: fast-log estimate, two neighbouring thresholds settle it)
__device__ __forceinline__ int hstu_bucket(const long long* thr, long long dt) {
  const long long x = dt < 0 ? -dt : dt;
  int b = (int)(__logf(fmaxf((float)x, 1.f)) * (1.0f / 0.301f));
  b = b < 0 ? 0 : (b > NBUCK - 1 ? NBUCK - 1 : b);
  const long long t0 = thr[b], t1 = thr[b < NBUCK - 1 ? b + 1 : b];
  if (t0 > x) b -= 1;
  else if (b < NBUCK - 1 && t1 <= x) b += 1;
  return b < 0 ? 0 : b;
}
// The buckets of a lane's eight partner rows of one tile (local rows kl(e) = t0 + 16 (e >> 2) + 4 g + (e & 3), ascending in e).  Timestamps
// are nondecreasing inside a session (the preparator sorts them: data_preparator.py:73-99), so |dt| against ONE owner is monotone along
// the partners and so is the bucket: when the first and the last of the eight agree, all eight do — two searches instead of eight
// wherever the tile is far from the diagonal (bucket widths grow by a factor 1.35).  `owner_is_query`: dt = t_owner - ts_p[kl]; else
// dt = ts_p[kl] - t_owner (the dK/dV pass: the partners are the queries).
// `span_valid`: all eight pairs are causal pairs inside the session — only then is dt of one sign over the span (a masked partner on the
// other side of the diagonal has the opposite sign, and |dt| is not monotone across it): otherwise every element is searched.
__device__ __forceinline__ void hstu_buckets8(const long long* thr, const long long* ts_p, long long t_owner, bool owner_is_query, int t0, int g,
                                              int len, bool span_valid, int (&bk)[8]) {
  auto dt_of = [&](int e) {
    const int kl = min(t0 + 16 * (e >> 2) + 4 * g + (e & 3), len - 1);
    return owner_is_query ? t_owner - ts_p[kl] : ts_p[kl] - t_owner;
  };
  auto one = [&](int e) { return hstu_bucket(thr, dt_of(e)); };
  const long long dt0 = dt_of(0), dt7 = dt_of(7);
  const int b0 = hstu_bucket(thr, dt0), b7 = hstu_bucket(thr, dt7);
  bk[0] = b0; bk[7] = b7;
  // the shortcut needs |dt| monotone over the span: both ends on the expected side of zero (a context time BEFORE the last history
  // stamp, or a preparator that does not sort by time, gives a V-shaped |dt| — then every element is searched, as the reference's
  // per-element bucketize does: hstu.py:99-113)
  if (span_valid && b0 == b7 && dt0 >= 0 && dt7 >= 0) {
#pragma unroll
    for (int e = 1; e < 7; ++e) bk[e] = b0;
  } else {
#pragma unroll
    for (int e = 1; e < 7; ++e) bk[e] = one(e);
  }
}
__device__ __forceinline__ float hstu_silu(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float hstu_silu_d(float z) { const float s = 1.f / (1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }
// both from ONE sigmoid (the dK/dV pass needs the probability and the derivative of the same element); p is hstu_silu's value up to the
// rounding of z * s against z / (1 + e)
__device__ __forceinline__ void hstu_silu_both(float z, float& p, float& d) {
  const float s = 1.f / (1.f + __expf(-z));
  p = z * s;
  d = s * (1.f + z * (1.f - s));
}

// Run-length accumulator of one lane's time-bias gradient (rt_attention.hip's TimeGradRun): along a lane's keys the bucket is monotone
// (timestamps are), equal buckets come in runs: one LDS atomic per run instead of one per score element.
struct BucketRun {
  int cur; float acc;
  __device__ __forceinline__ void init() { cur = -1; acc = 0.f; }
  __device__ __forceinline__ void add(float* dtw, int b, float v) {
    if (b != cur) { if (cur >= 0) atomicAdd(dtw + cur, acc); cur = b; acc = v; } else acc += v;
  }
  __device__ __forceinline__ void flush(float* dtw) { if (cur >= 0) atomicAdd(dtw + cur, acc); cur = -1; acc = 0.f; }
};

// Position-bias gradient of one 16-query x 4-key block of a lane row (queries i = 0 .. 15 across the row's lanes, a lane's four
// consecutive keys j = 0 .. 3): element (i, j) belongs to slot base + j - i of d_pos_w, so the 64 values fall on 19 diagonals.  Summed along
// the diagonals in registers with row shifts (DPP, no LDS): `main` of lane i = the diagonal through its element 3 (slot base + 3 - i, whole
// for every lane: what a shift drops in at the row's start is zero), `wrap` of lanes 0 .. 2 = the three diagonals (-13, -14, -15) whose
// elements the shifts push out of the row's end (slot base - 13 - i).  Two LDS atomics per block instead of four, 16 + 3 live lanes per row
// instead of 64 values — the atomics were the dQ pass's largest non-matrix cost (one per valid score element).
__device__ __forceinline__ float row_shr1(float v) {   // lane i of a 16-lane row takes lane i - 1's value, lane 0 takes zero
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_ror1(float v) {   // ... lane 0 takes lane 15's
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
}
__device__ __forceinline__ void diagonal_sums(const float* d4, int i, float& main, float& wrap) {
  main = row_shr1(row_shr1(row_shr1(d4[0]) + d4[1]) + d4[2]) + d4[3];
  wrap = row_ror1(row_ror1(row_ror1(i >= 13 ? d4[0] : 0.f) + (i >= 14 ? d4[1] : 0.f)) + (i >= 15 ? d4[2] : 0.f));
}

}  // namespace rt_planes
