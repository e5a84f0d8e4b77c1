// K7w — fp32-accurate GEMM with a PRE-SPLIT weight operand:  C[M,N] = A[M,K] . W' (+ bias) (+ residual) (relu)
//
// The bf16x6 GEMM of rt_gemm.hip splits every fp32 fragment into three bf16 planes in registers, in every wave that consumes it: 352 VALU
// instructions per k-step and wave beside 48 MFMAs — the split, not the matrix pipe, bounds it (186 TF at 8192^3 =
// 0.45 of the 2,500 / 6 TF the arithmetic allows).  Half of those fragments are WEIGHTS: a few hundred KB that every tile of every
// product of a step re-reads and re-splits.  Here the weights are split ONCE per forward pass by `rt_split_planes` (three bf16 planes,
// 6 bytes per element, exact: x = h + m + l) and the kernel streams the planes through the LDS-DMA ring as they are; only the
// activation operand is split in registers (once per wave that owns its rows).  Same six bf16 products per fp32 product, same
// accumulation order per tile as rt_gemm's loop, so results agree with it to fp32 rounding.
//
//   w_tr = 0   W'(n, k) = W[n * ldw + k]   y  = x W^T   (nn.Linear forward: net_blocks.py:63-64, sasrec.py:191,221-229)
//              plane tile [128 n][32 k] bf16, B fragments by ds_read_b128
//   w_tr = 1   W'(n, k) = W[k * ldw + n]   dx = dy W    (its data gradient)
//              plane tile [32 k][128 n] bf16 — the reduction index runs over ROWS: B fragments by ds_read_b64_tr_b16 (transpose read)
// Tiles: 128 x 128 per 256-thread workgroup, 2 x 2 waves of 2 x 2 v_mfma_f32_32x32x16_bf16 tiles, BK = 32, two stages of
// (16 KB fp32 activations + 3 x 8 KB weight planes) = 80 KB: two workgroups per CU.  Exact tile grids only (M, N % 128, K % 32,
// 16-byte aligned operands): other shapes answer RT_ERR_UNSUPPORTED and the caller takes rt_gemm.
#include <stdlib.h>

#include "rt_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, GT = 256;
constexpr int A_TILE_B = BM * BK * 4;          // 16 KB
constexpr int P_TILE_B = BN * BK * 2;          // 8 KB per plane
constexpr int STAGE_B = A_TILE_B + 3 * P_TILE_B;
constexpr int NS = 2;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define RT_LDS __attribute__((address_space(3)))

struct WpArgs {
  const float* A; long long lda;
  const unsigned short* W; long long plane_stride, ldw;   // planes p at W + p * plane_stride (elements)
  float* C; long long ldc;
  const float* bias; const float* R; long long ldr;
  int M, N, K, relu;
};
struct WpGroup { WpArgs g[4]; int tile_end[4]; int n; };

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Split3 { bf16x8 h, m, l; };
__device__ __forceinline__ Split3 split_bf16x3(const f32x4& x0, const f32x4& x1) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = q < 2 ? x0[2 * q] : x1[2 * q - 4], b = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    const float la = ra - __uint_as_float(va & 0xFFFF0000u), lb = rb - __uint_as_float(vb & 0xFFFF0000u);
    ph[q] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    pm[q] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    pl[q] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
  }
  Split3 r;
  r.h = __builtin_bit_cast(bf16x8, ph); r.m = __builtin_bit_cast(bf16x8, pm); r.l = __builtin_bit_cast(bf16x8, pl);
  return r;
}

// one float4 (4 consecutive k) of activation row `row`, 16-byte chunk c of the [128][8 x 16 B] image (chunk stored at c ^ ((row>>1)&7))
__device__ __forceinline__ f32x4 read_a(const unsigned char* S, int row, int c) {
  return *reinterpret_cast<const f32x4*>(S + row * (BK * 4) + ((c ^ ((row >> 1) & 7)) << 4));
}

// Wave layout.  RI x CJ = the 32 x 32 MFMA tiles of a wave.  2 x 2 (waves as a 2 x 2 grid): every activation fragment is read — and SPLIT —
// by the two waves of its row pair.  1 x 4 (waves stacked along M, each across the whole tile width): every activation row belongs to ONE
// wave, so the split — the only VALU work left in this kernel — is done once per element; the weight planes are read by all four waves
// instead (LDS reads only: they need no arithmetic).
// NLD = 2: two extra LOADER waves issue every LDS-DMA piece of the ring; the four compute waves only meet them at the barrier.  A
// global_load_lds piece occupies its wave's issue port for 60 - 185 cycles (MI355X_MICROARCH.md): 10 pieces per k-step and compute wave
// were ~40 % of that step's 48 MFMAs of issue time on a SIMD that holds one compute wave (the ablation of DESIGN.md K7w: DMA and MFMA
// time add up).  NLD = 0: every wave loads its own share (the round-3 kernel).
template <bool BTR, int RI, int NLD = 0>
__global__ __launch_bounds__(GT + 64 * NLD) void gemm_wp_kernel(WpGroup gg) {
  constexpr int CJ = 4 / RI;
  constexpr int NISS = NLD ? NLD : 4;          // issuing waves
  constexpr int APW = 16 / NISS;               // activation pieces (8 rows x 128 B = 1 KB) per issuing wave and stage
  constexpr int BPW = 8 / NISS;                // weight pieces per issuing wave, plane and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int t = blockIdx.x;
  const int p = (t >= gg.tile_end[0]) + (t >= gg.tile_end[1]) + (t >= gg.tile_end[2]);
  t -= p > 0 ? gg.tile_end[p - 1] : 0;
  const WpArgs& g = gg.g[p];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: branches on it are scalar
  const int col = lane & 31, half = lane >> 5;
  const int wm = RI == 2 ? wave >> 1 : wave, wn = RI == 2 ? wave & 1 : 0;
  const int n_tn = g.N / BN;
  const int n_tiles = (g.M / BM) * n_tn;
  {   // consecutive tiles of one XCD share weight planes and neighbouring activation rows in that XCD's L2
    const int nx = 8, q = n_tiles / nx, r = n_tiles % nx, xcd = t % nx, idx = t / nx;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (t / n_tn) * BM, n0 = (t % n_tn) * BN;
  const int n_steps = g.K / BK;

  // ---- DMA sources (k0 = 0).  Activations: APW pieces per issuing wave, chunk c of row r fetched into slot c ^ ((r>>1)&7).
  const bool computes = NLD ? wave < 4 : true;
  const bool issuer = NLD ? wave >= 4 : true;
  const int iw = NLD ? (wave >= 4 ? wave - 4 : 0) : wave;       // index among the issuing waves
  const float* a_src[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int row = (iw * APW + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    a_src[j] = g.A + (long long)(m0 + row) * g.lda + c * 4;
  }
  // Weight planes: BPW pieces per issuing wave and plane (8 per plane tile of 8 KB); the LDS image is lane-linear, the swizzle is in the source
  const unsigned short* b_src[BPW];
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int q = (iw * BPW + j) * 64 + lane;                  // 16-byte unit of the plane tile
    if (!BTR) {
      const int row = q >> 2, c = (q & 3) ^ ((row >> 2) & 3);   // [128 n][4 units]: unit c of row n stored at c ^ ((n>>2)&3)
      b_src[j] = g.W + (long long)(n0 + row) * g.ldw + c * 8;
    } else {
      const int row = q >> 4, u = (q & 15) ^ ((row & 3) << 2);  // [32 k][16 units]: unit u of row k stored at u ^ ((k&3)<<2)
      b_src[j] = g.W + (long long)row * g.ldw + n0 + u * 8;
    }
  }
  const long long b_step = BTR ? (long long)BK * g.ldw : BK;
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const unsigned a_ofs = __builtin_amdgcn_readfirstlane((unsigned)(iw * APW * 1024));
  const unsigned b_ofs = __builtin_amdgcn_readfirstlane((unsigned)(A_TILE_B + iw * BPW * 1024));

  int issued = 0, iss_stage = 0;
  auto issue_next = [&]() {
    const unsigned sb = smem_base + (unsigned)(iss_stage * STAGE_B);
#pragma unroll
    for (int j = 0; j < APW; ++j) { dma16(a_src[j], sb + a_ofs + j * 1024); a_src[j] += BK; }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < BPW; ++j) dma16(b_src[j] + pl * g.plane_stride, sb + b_ofs + pl * P_TILE_B + j * 1024);
#pragma unroll
    for (int j = 0; j < BPW; ++j) b_src[j] += b_step;
    iss_stage ^= 1;
    ++issued;
  };
  if (issuer && issued < n_steps) issue_next();
  if (NLD && !computes) {   // loader waves: keep the two-stage ring full, one barrier per k-step in step with the compute waves
#pragma unroll 1
    for (int st = 0; st < n_steps; ++st) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (issued < n_steps) issue_next();
    }
    return;
  }

  f32x16 acc[RI][CJ];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int cons_stage = 0;
#pragma unroll 1
  for (int st = 0; st < n_steps; ++st) {
    if (NLD == 0) wait_vmcnt<0>();                    // stage st has landed (two stages: nothing else is in flight here)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NLD == 0 && issued < n_steps) issue_next();   // refills the buffer consumed at step st - 1, under this step's MFMAs
    const unsigned char* Ab = smem + cons_stage * STAGE_B;
    const unsigned char* Bb = Ab + A_TILE_B;
    cons_stage ^= 1;
#pragma unroll
    for (int u = 0; u < BK / 16; ++u) {
      Split3 as[RI], bs[CJ];
#pragma unroll
      for (int i = 0; i < RI; ++i) {                  // k = 16 u + 8 half + (0..7): chunks 4u + 2 half, 4u + 2 half + 1
        const int row = wm * (32 * RI) + i * 32 + col;
        as[i] = split_bf16x3(read_a(Ab, row, 4 * u + 2 * half), read_a(Ab, row, 4 * u + 2 * half + 1));
      }
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        if (!BTR) {
          const int n = wn * (32 * CJ) + j * 32 + col;
          const unsigned char* q = Bb + n * (BK * 2) + ((((unsigned)(2 * u + half)) ^ ((n >> 2) & 3)) << 4);
          bs[j].h = *reinterpret_cast<const bf16x8*>(q);
          bs[j].m = *reinterpret_cast<const bf16x8*>(q + P_TILE_B);
          bs[j].l = *reinterpret_cast<const bf16x8*>(q + 2 * P_TILE_B);
        } else {
          // 16-lane group G reads [4 k rows][16 n columns]: lane i supplies row (i >> 2), 4 columns 4 (i & 3) and receives column i
          const int i16 = lane & 15, G = lane >> 4;
          const int ncol = wn * (32 * CJ) + j * 32 + (G & 1) * 16 + 4 * (i16 & 3);   // first of this lane's 4 columns
          s16x8 v[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            s16x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int k = 16 * u + 8 * half + 4 * e + (i16 >> 2);
              const unsigned char* q = Bb + pl * P_TILE_B + k * (BN * 2) + ((((unsigned)(ncol >> 3)) ^ ((k & 3) << 2)) << 4) + ((ncol & 7) << 1);
              const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RT_LDS s16x4*)(q));
              if (e == 0) lo = x; else hi = x;
            }
            v[pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
          bs[j].h = __builtin_bit_cast(bf16x8, v[0]); bs[j].m = __builtin_bit_cast(bf16x8, v[1]); bs[j].l = __builtin_bit_cast(bf16x8, v[2]);
        }
      }
#define RT_WP_TERM(PA, PB)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < RI; ++i) _Pragma("unroll") for (int j = 0; j < CJ; ++j)                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[i].PA, bs[j].PB, acc[i][j], 0, 0, 0);
      RT_WP_TERM(l, h) RT_WP_TERM(h, l) RT_WP_TERM(m, m) RT_WP_TERM(m, h) RT_WP_TERM(h, m) RT_WP_TERM(h, h)
#undef RT_WP_TERM
    }
  }

  // epilogue (tiles are exact): bias, residual, relu
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      const int n = n0 + wn * (32 * CJ) + j * 32 + col;
      const int mb = m0 + wm * (32 * RI) + i * 32 + 4 * half;
      const float bv = g.bias != nullptr ? g.bias[n] : 0.f;
      float rv[16];
      if (g.R != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = g.R[(long long)(mb + (r & 3) + 8 * (r >> 2)) * g.ldr + n];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv + rv[r];
        if (g.relu) v = fmaxf(v, 0.f);
        g.C[(long long)(mb + (r & 3) + 8 * (r >> 2)) * g.ldc + n] = v;
      }
    }
}

// (A second form — four loader waves feeding a ring that runs ahead across tile boundaries, four compute waves without a vector-memory
// instruction in the loop, persistent tiles — was built in round 4 and measured equal to the loop above within 0 - 7 % at every shape of
// the step and of the recommend encoder: what bounds these products is the CU's memory pipe (activation rows in, C rows out, the weight
// planes through the L2, ~13 B/clk), not the issue cost of the DMA.  Removed in round 5 together with its switches; the lever is fewer
// bytes per row: rt_ffn.hip.)

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long long n4, long long n,
                                                           unsigned short* __restrict__ planes, long long plane_stride) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(src + 4 * q);
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float a = x[2 * e], b = x[2 * e + 1];
      const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
      const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
      const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
      const float la = ra - __uint_as_float(va & 0xFFFF0000u), lb = rb - __uint_as_float(vb & 0xFFFF0000u);
      h[e] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
      m[e] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
      l[e] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(planes + 4 * q) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(planes + plane_stride + 4 * q) = u32x2{m[0], m[1]};
    *reinterpret_cast<u32x2*>(planes + 2 * plane_stride + 4 * q) = u32x2{l[0], l[1]};
  }
}

bool wp_ok(const WpArgs& a) {
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  return a.M > 0 && a.N > 0 && a.K > 0 && a.M % BM == 0 && a.N % BN == 0 && a.K % BK == 0 && (a.lda & 3) == 0 && (a.ldw & 7) == 0 &&
         (a.plane_stride & 7) == 0 && !mis(a.A) && !mis(a.W) && a.A != nullptr && a.W != nullptr && a.C != nullptr;
}

}  // namespace

extern "C" {

// planes[p * plane_stride + i], p = 0, 1, 2 <- the exact three-way bf16 split of src[i] (h, m, l), i < n; n % 4 == 0, plane_stride % 8 == 0,
// 16-byte aligned pointers.  One launch covers a whole contiguous range of parameters (every weight of a layer stack).
int rt_split_planes(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, hipStream_t stream) {
  (void)hipGetLastError();
  if (n < 0 || (n & 3) || (plane_stride & 7) || plane_stride < n) return RT_ERR_INVALID_ARG;
  if (n == 0) return RT_OK;
  if (src == nullptr || planes == nullptr || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return RT_ERR_INVALID_ARG;
  const long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8LL * rt_num_cus()) blocks = 8LL * rt_num_cus();
  split_planes_kernel<<<(int)blocks, 256, 0, stream>>>(src, n4, n, planes, plane_stride);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

// Up to 4 products C = A . W' (+ bias) (+ R) (relu) with pre-split weights in ONE launch (all with the same w_tr).
struct rt_gemm_wp_problem {
  const float* A; int64_t lda;
  const uint16_t* W; int64_t plane_stride, ldw;     // planes of the weight (rt_split_planes); ldw: row stride of W in elements
  float* C; int64_t ldc;
  const float* bias; const float* R; int64_t ldr;
  int32_t M, N, K, relu;
};
int rt_gemm_wp(const rt_gemm_wp_problem* problems, int32_t n, int32_t w_tr, hipStream_t stream) {
  (void)hipGetLastError();
  if (problems == nullptr || n < 1 || n > 4) return RT_ERR_INVALID_ARG;
  WpGroup gg{};
  gg.n = n;
  int tiles = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      const rt_gemm_wp_problem& q = problems[i];
      WpArgs a{};
      a.A = q.A; a.lda = q.lda; a.W = q.W; a.plane_stride = q.plane_stride; a.ldw = q.ldw; a.C = q.C; a.ldc = q.ldc; a.bias = q.bias; a.R = q.R;
      a.ldr = q.ldr; a.M = q.M; a.N = q.N; a.K = q.K; a.relu = q.relu;
      if (!wp_ok(a)) return RT_ERR_UNSUPPORTED;
      gg.g[i] = a;
      tiles += (a.M / BM) * (a.N / BN);
    }
    gg.tile_end[i] = tiles;
  }
  const size_t lds = (size_t)NS * STAGE_B;
  auto go = [&](auto kern, int threads) -> int {
    RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<tiles, threads, lds, stream>>>(gg);
    RT_CHECK_LAUNCH();
    return RT_OK;
  };
  return w_tr ? go(&gemm_wp_kernel<true, 1>, GT) : go(&gemm_wp_kernel<false, 1>, GT);
}

}  // extern "C"
