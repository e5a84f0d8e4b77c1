// K7f — the feed-forward half of a SASRec block as ONE launch per direction (sasrec.py:225-229: `seqs = LN2(seqs); seqs = FFN(seqs) + seqs`,
// net_blocks.py:63-64: `fc2(dropout1(relu(fc1(x))))` + the block's dropout2), on packed rows.
//
// Before (per block and direction): LayerNorm, GEMM, dropout, GEMM, dropout — five launches of 7 - 25 us that each ramp up and drain on a
// 256-CU machine, and nine passes of [M, 256] fp32 arrays through HBM.  Here a workgroup owns 64 rows from the LayerNorm input to the
// block output:
//
//   forward   f = LN2(y)                      (prologue: one wave per row, the arithmetic of layernorm_fwd_kernel; f, mean, rstd are saved)
//             hdrop = drop(relu(f W1^T + b1)) (first product; the mask is applied in the epilogue — only hdrop is kept: relu'(h) = [hdrop != 0]
//                                              wherever the mask kept the element, and elsewhere the gradient is zero anyway)
//             out = f + drop(hdrop W2^T + b2) (second product)
//   backward  g_o = drop'(g_out)              (prologue)
//             g_h = [hdrop != 0] / keep * (g_o W2)
//             g_f = g_h W1 + g_out
//
// What the first form of this kernel taught (visit v4b/v4c of round 4: 62 us per launch — no faster than the five launches): a 256-thread
// workgroup that issues its own LDS-DMA, waits `vmcnt(0)` at every k-step and streams its activations back from L2 spends ~1 us per
// 32-wide k-step on issue + landing latency with the matrix pipe idle, and its stores gate the ring.  This form separates the roles:
//   * the 64 activation rows of the current product live in the LDS as fp32 ([64][K], 16-byte units XOR-swizzled by row & 15: operand
//     reads and epilogue writes are bank-conflict free) — written by the prologue / by the first product's epilogue straight from the
//     accumulators; no activation ever comes back from memory;
//   * four LOADER waves stream the weight planes — the only DMA traffic left — through a three-stage ring that runs ahead by two k-steps
//     across pass and product boundaries (counted `vmcnt`: a loader's counter sees only its own pieces);
//   * four COMPUTE waves (2 x 2 over a 64 x 128 pass tile) issue no vector-memory instruction inside the loop; their epilogue stores drain
//     under the next pass (their own `vmcnt` is never waited on until the kernel ends).
// Both products run K7w's arithmetic (rt_gemm_wp.hip: six v_mfma_f32_32x32x16_bf16 terms per fp32 product, same k order and term order —
// results equal the unfused path's bit for bit).  The matrix instruction is issued with its operands SWAPPED (weights as the row operand):
// a lane then holds, for one activation row, four groups of four CONSECUTIVE output features — the float4 groups the dropout hash is keyed
// by, a 16-byte store, and a 16-byte LDS write of the next product's operand.  LDS: 64 KB of activations + 72 KB of ring at d = dff = 256.
#include <stdlib.h>

#include "rt_common.h"

namespace {

constexpr int BM = 64, BN = 128, BK = 32, GT = 512;
constexpr int P_TILE_B = BN * BK * 2;          // 8 KB per plane
constexpr int W_STAGE_B = 3 * P_TILE_B;        // 24 KB: the three planes of a [128 n][32 k] weight tile
constexpr int NSTG = 3;
constexpr int PIECES = 6;                      // DMA pieces (1 KB each) per loader wave and stage

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define RT_LDS __attribute__((address_space(3)))

struct FfnArgs {
  int M, d, dff;
  float p, eps;
  unsigned long long seed_h, sid_h, seed_o, sid_o;
  const unsigned short *w1p, *w2p;   // bf16 planes of W1 [dff, d] and W2 [d, dff] (plane q at + q * plane_stride elements)
  long long plane_stride;
  // forward
  const float *y, *ln_w, *ln_b, *b1, *b2;
  float *f, *mean, *rstd, *hdrop, *out, *y_out;
  const float *attn, *q, *bo;        // first = 0: the out-projection's operand rows, its skip branch and bias
  const unsigned short* wop;         // bf16 planes of Wo [d, d]
  int first, last;
  // backward
  const float *g_out, *hd;
  float *g_o, *g_h, *g_f, *g_y, *g_A, *ln_partial;
  int probe;      // ablation builds only (-DRT_ABLATION_BUILD, RT_FFN_PROBE): 1 no prologue stores, 2 no first-epilogue stores, 4 no second epilogue, 8 no MFMA, 16 no DMA, 32 no operand reads
};

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
__device__ __forceinline__ Split3 split_bf16x3(const f32x4& x0, const f32x4& x1) {      // rt_gemm_wp.hip's split, bit for bit
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

// The activation image: row r (0..63) of the current product's operand, K floats, 16-byte unit u stored at u ^ (r & 15).
__device__ __forceinline__ unsigned a_off(int row, int unit, int rs) { return (unsigned)(row * rs + ((unit ^ (row & 15)) << 4)); }

// ---- loader side: the weight stream of the chain's products as ONE sequence of stages ----------------------------------------------------
struct WStream {                     // the CURRENT product's sources only: a table of all three would be indexed by the compiler
  //                                    through scratch memory, and scratch traffic counts in the vmcnt the ring's run-ahead is metered with
  const unsigned short *sa, *sb;     // this lane's 16-byte unit of pieces 2 lw, 2 lw + 1 at (pass 0, k 0)
  long long ldw;
  int KS, T;                         // ring stages per pass / of the whole product
  long long plane_stride;
  unsigned base;                     // LDS byte address of this loader wave's first piece in ring slot 0
  int g, kk, pass, slot, issued;
  bool no_dma;
};
template <bool BTR>
__device__ __forceinline__ const unsigned short* wsrc(const unsigned short* W, long long ldw, int lw, int lane, int j) {
  const int q = (lw * 2 + j) * 64 + lane;                    // 16-byte unit of the plane tile
  if (!BTR) {
    const int row = q >> 2, c = (q & 3) ^ ((row >> 2) & 3);   // [128 n][4 units]: unit c of row n stored at c ^ ((n>>2)&3)
    return W + (long long)row * ldw + c * 8;
  }
  const int row = q >> 4, u = (q & 15) ^ ((row & 3) << 2);    // [32 k][16 units]: unit u of row k stored at u ^ ((k&3)<<2)
  return W + (long long)row * ldw + u * 8;
}
// weight, row stride, K and N of product g of the chain (forward: Wo, W1, W2; backward: W2, W1, Wo — read as [k][n])
template <bool FWD>
__device__ __forceinline__ void product_of(const FfnArgs& a, int g, const unsigned short*& W, long long& ldw, int& K, int& N) {
  const int d = a.d, dff = a.dff;
  if (g == 0) { W = FWD ? a.wop : a.w2p; ldw = FWD ? d : dff; K = d; N = FWD ? d : dff; }
  else if (g == 1) { W = a.w1p; ldw = d; K = FWD ? d : dff; N = FWD ? dff : d; }
  else { W = FWD ? a.w2p : a.wop; ldw = FWD ? dff : d; K = FWD ? dff : d; N = d; }
}
template <bool BTR>
__device__ __forceinline__ void wstream_open(WStream& w, const FfnArgs& a, int g, int lw, int lane) {
  const unsigned short* W; long long ldw; int K, N;
  product_of<!BTR>(a, g, W, ldw, K, N);
  w.sa = wsrc<BTR>(W, ldw, lw, lane, 0);
  w.sb = wsrc<BTR>(W, ldw, lw, lane, 1);
  w.ldw = ldw; w.KS = K / BK; w.T = w.KS * (N / BN);
  w.g = g; w.kk = 0; w.pass = 0;
}
template <bool BTR>
__device__ __forceinline__ void wstream_issue(WStream& w, const FfnArgs& a, int lw, int lane) {
  const long long bo = BTR ? (long long)w.kk * BK * w.ldw + (long long)w.pass * BN : (long long)w.pass * BN * w.ldw + (long long)w.kk * BK;
  const unsigned sb = w.base + (unsigned)(w.slot * W_STAGE_B);
  if (!w.no_dma) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      dma16(w.sa + bo + pl * w.plane_stride, sb + pl * P_TILE_B);
      dma16(w.sb + bo + pl * w.plane_stride, sb + pl * P_TILE_B + 1024);
    }
  }
  ++w.issued;
  if (++w.slot == NSTG) w.slot = 0;
  if (++w.kk == w.KS) {
    w.kk = 0;
    if ((w.pass + 1) * w.KS == w.T) wstream_open<BTR>(w, a, w.g + 1, lw, lane); else ++w.pass;
  }
}

// ---- compute side: one pass (64 x 128 tile, this wave's 32 x 64 piece) over KS ring stages ------------------------------------------------
#ifdef RT_ABLATION_BUILD
#define RT_PROBE(bit) ((probe & (bit)) != 0)
#else
#define RT_PROBE(bit) false
#endif
template <bool BTR>
__device__ __forceinline__ void compute_pass(const unsigned char* A, int rs, const unsigned char* ring, int& slot, int KS, int lane, int wm, int wn,
                                             f32x16 (&acc)[2], int probe) {
  const int col = lane & 31, half = lane >> 5;
  const int row = wm * 32 + col;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
  for (int kk = 0; kk < KS; ++kk) {
    __builtin_amdgcn_s_barrier();                       // the loaders have seen this stage land; everyone is done with the previous one
    asm volatile("" ::: "memory");
    const unsigned char* Bb = ring + slot * W_STAGE_B;
    if (++slot == NSTG) slot = 0;
    if (RT_PROBE(32)) continue;
#pragma unroll
    for (int u = 0; u < BK / 16; ++u) {
      const int unit = kk * 8 + 4 * u + 2 * half;       // k = 32 kk + 16 u + 8 half + (0..7): two 16-byte units
      const Split3 as = split_bf16x3(*reinterpret_cast<const f32x4*>(A + a_off(row, unit, rs)),
                                     *reinterpret_cast<const f32x4*>(A + a_off(row, unit + 1, rs)));
      Split3 bs[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!BTR) {
          const int n = wn * 64 + j * 32 + col;
          const unsigned char* q = Bb + n * (BK * 2) + ((((unsigned)(2 * u + half)) ^ ((n >> 2) & 3)) << 4);
          bs[j].h = *reinterpret_cast<const bf16x8*>(q);
          bs[j].m = *reinterpret_cast<const bf16x8*>(q + P_TILE_B);
          bs[j].l = *reinterpret_cast<const bf16x8*>(q + 2 * P_TILE_B);
        } else {
          // 16-lane group G reads [4 k rows][16 n columns]: lane i supplies row (i >> 2), 4 columns 4 (i & 3) and receives column i
          const int i16 = lane & 15, G = lane >> 4;
          const int ncol = wn * 64 + j * 32 + (G & 1) * 16 + 4 * (i16 & 3);   // first of this lane's 4 columns
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
      // K7w's six terms in K7w's order (activation plane, weight plane): (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); operands swapped
#define RT_FFN_TERM(PA, PB) \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bs[j].PB, as.PA, acc[j], 0, 0, 0);
      if (RT_PROBE(8)) {     // keep the operand reads and the split alive without the matrix work
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j][0] += (float)(as.h[0] + as.m[1] + as.l[2]) + (float)(bs[j].h[0] + bs[j].m[3] + bs[j].l[5]);
        continue;
      }
      RT_FFN_TERM(l, h) RT_FFN_TERM(h, l) RT_FFN_TERM(m, m) RT_FFN_TERM(m, h) RT_FFN_TERM(h, m) RT_FFN_TERM(h, h)
#undef RT_FFN_TERM
    }
  }
}

#ifdef RT_ABLATION_BUILD
#define RT_PROBEA(bit) ((a.probe & (bit)) != 0)
#else
#define RT_PROBEA(bit) false
#endif
constexpr int NPMAX = 2;      // passes per product the register budget covers (d, dff <= 256)
constexpr int RED_B = 4 * 2 * 256 * 4;   // LayerNorm-backward partials of the four compute waves: [4][2][256] floats

// LayerNorm of 8 rows held as one float4 per lane (layernorm_fwd_kernel's arithmetic: sum -> mean -> centred squares -> rstd ->
// (v - mu) rs w + b), the rows' reductions interleaved; v <- LN(v), mu / rs per row.
__device__ __forceinline__ void ln8(f32x4 (&v)[8], bool on, int d, float eps, const f32x4& ww, const f32x4& bb, float (&mu)[8], float (&rs)[8]) {
  float q[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) mu[r] = 0.f + (v[r][0] + v[r][1] + v[r][2] + v[r][3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < 8; ++r) mu[r] += __shfl_xor(mu[r], o, 64);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    mu[r] = mu[r] / d;
    q[r] = 0.f;
    if (on) {
      const f32x4 t = v[r] - mu[r];
      q[r] += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < 8; ++r) q[r] += __shfl_xor(q[r], o, 64);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    rs[r] = 1.0f / sqrtf(q[r] / d + eps);
    if (on) v[r] = (v[r] - mu[r]) * rs[r] * ww + bb;
  }
}

// MODE 0: forward (training: every intermediate the backward needs is written), 1: backward, 2: forward without the saves (inference)
// a.first: 0 = the chain starts at the out-projection (three products), 1 = at the feed-forward input (two products; MODE 0 / 2)
// a.last (MODE 1): 2 = stop at g_f, 3 = through the LayerNorm backward and the out-projection's data gradient
template <int MODE>
__global__ __launch_bounds__(GT) void ffn_kernel(FfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int cw = wave & 3, wm = cw >> 1, wn = cw & 1, col = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int d = a.d, dff = a.dff;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const int KA = d > dff ? d : dff;
  unsigned char* A = smem;                                   // [64][K] fp32 operand rows of the current product
  unsigned char* ring = smem + BM * KA * 4;
  float* red = reinterpret_cast<float*>(ring + NSTG * W_STAGE_B);
  constexpr bool BTR = MODE == 1;
  constexpr bool FWD = MODE != 1;
  // products  forward : [0] y = q + attn Wo^T + bo (K = d, N = d)   [1] hdrop = drop(relu(f W1^T + b1)) (K = d, N = dff)   [2] out = f + drop(hdrop W2^T + b2)
  //           backward: [0] g_h = mask (g_o W2) (K = d, N = dff)    [1] g_f = g_h W1 + g_out (K = dff, N = d)            [2] g_A = g_y Wo (K = d, N = d)
  const int first = FWD ? a.first : 0, last = FWD ? 3 : a.last;
  const int K0 = d, K1 = FWD ? d : dff, K2 = FWD ? dff : d, N0 = FWD ? d : dff, N1 = FWD ? dff : d, N2 = d;
  const int c4 = lane * 4;
  const bool on = c4 < d;                                    // d = 128: half of the lanes hold a row's columns
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  WStream ws;
  if (loader) {    // the ring runs ahead of the prologue: the weights do not depend on it
    ws.plane_stride = a.plane_stride;
    ws.base = __builtin_amdgcn_readfirstlane(lds_addr(ring) + (unsigned)(cw * 2 * 1024));
    ws.slot = 0; ws.issued = 0; ws.no_dma = false;
    wstream_open<BTR>(ws, a, first, cw, lane);
#ifdef RT_ABLATION_BUILD
    ws.no_dma = (a.probe & 16) != 0;
#endif
    wstream_issue<BTR>(ws, a, cw, lane);
    wstream_issue<BTR>(ws, a, cw, lane);
  }

  // ---- prologue over the workgroup's 64 rows (8 per wave): the first product's operand rows into the LDS image
  {
    const int rs = d * 4;
    f32x4 v[8];
    const float* src = !FWD ? a.g_out : (first == 0 ? a.attn : a.y);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = on ? *reinterpret_cast<const f32x4*>(src + (long long)(m0 + wave * 8 + r) * d + c4) : zero4;
    if (FWD && first == 1) {                                 // f = LN2(y)
      float mu[8], rsd[8];
      f32x4 ww = zero4, bb = zero4;
      if (on) { ww = *reinterpret_cast<const f32x4*>(a.ln_w + c4); bb = *reinterpret_cast<const f32x4*>(a.ln_b + c4); }
      ln8(v, on, d, a.eps, ww, bb, mu, rsd);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int lr = wave * 8 + r, m = m0 + lr;
        if (on) {
          if (!RT_PROBEA(1)) *reinterpret_cast<f32x4*>(a.f + (long long)m * d + c4) = v[r];
          *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = v[r];
        }
        if (MODE == 0 && lane == 0) { a.mean[m] = mu[r]; a.rstd[m] = rsd[r]; }
      }
    } else if (FWD) {                                        // the attention's output rows as they are
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (on) *reinterpret_cast<f32x4*>(A + a_off(wave * 8 + r, lane, rs)) = v[r];
    } else {
      // g_o = drop'(g_out) with the mask of the forward's output dropout (stream seed_o / sid_o, keyed by the float4 group of [M, d])
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int lr = wave * 8 + r, m = m0 + lr;
        if (on) {
          f32x4 gv = v[r];
          if (a.p > 0.f) {
            gv = rt_drop4(gv, a.seed_o, a.sid_o, ((unsigned long long)m * d + c4) >> 2, a.p, inv_keep);
            if (!RT_PROBEA(1)) *reinterpret_cast<f32x4*>(a.g_o + (long long)m * d + c4) = gv;
          }
          *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = gv;
        }
      }
    }
  }
  if (loader) wait_vmcnt<0>();       // the loaders' counters start clean: from here on they hold ring pieces only (stages 0 and 1 have landed)
  __syncthreads();                   // B0: the operand rows of the first product are in the LDS

  if (loader) {
    // ---- weight stream: stage gi is published by barrier S_gi; stage gi + 2 is issued right behind it into the slot stage gi - 1 left.
    // Between two products the compute waves rewrite the operand image: two barriers, three where a LayerNorm pass runs on the image
    // (forward: behind the out-projection; backward: behind g_f) — the loaders only keep count; the ring keeps flying
    const int T0 = (K0 / BK) * (N0 / BN), T1 = (K1 / BK) * (N1 / BN), T2 = (K2 / BK) * (N2 / BN);
    const int T = (first == 0 ? T0 : 0) + T1 + (last == 3 ? T2 : 0);
    int bound = first == 0 ? T0 : T1, seg = first;
#pragma unroll 1
    for (int gi = 0; gi < T; ++gi) {
      if (gi + 1 < ws.issued) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();      // everything but the youngest stage has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ws.issued < T) wstream_issue<BTR>(ws, a, cw, lane);
      if (gi == bound - 1 && gi + 1 < T) {
        const int nb = (FWD ? seg == 0 : seg == 1) ? 3 : 2;
#pragma unroll 1
        for (int i = 0; i < nb; ++i) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ++seg;
        bound += seg == 1 ? T1 : T2;
      }
    }
    return;
  }

  // ---- compute waves -------------------------------------------------------------------------------------------------------------------
  const int m = m0 + wm * 32 + col;     // the activation row of this lane in every epilogue
  const int lrow = wm * 32 + col;
  int slot = 0;
  f32x16 acc[2];
  f32x4 keep[NPMAX][2][4];              // a product's results (all passes): the next product's operand rows
  auto lds_sync = [&]() {               // LDS-only synchronisation: the epilogues' global stores keep draining
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto publish = [&](int K_next, int np) {      // X1: every wave has read its last fragment; the kept rows become the image; X2
    lds_sync();
    const int rs = K_next * 4;
#pragma unroll
    for (int pass = 0; pass < NPMAX; ++pass)
      if (pass < np) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
            *reinterpret_cast<f32x4*>(A + a_off(lrow, n >> 2, rs)) = keep[pass][j][g];
          }
      }
    lds_sync();
  };
#define RT_FOR_TILE(body)                                                                      \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int g = 0; g < 4; ++g) { \
    const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;                             \
    body                                                                                       \
  }
#define RT_ACC4 f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]}

  if (FWD) {
    if (first == 0) {
      // ---- [0] y = q + attn Wo^T + bo, then f = LN2(y) on the image
#pragma unroll
      for (int pass = 0; pass < NPMAX; ++pass)
        if (pass < N0 / BN) {
          compute_pass<BTR>(A, K0 * 4, ring, slot, K0 / BK, lane, wm, wn, acc, a.probe);
          f32x4 qv[2][4];
          RT_FOR_TILE(qv[j][g] = *reinterpret_cast<const f32x4*>(a.q + (long long)m * d + n);)
          RT_FOR_TILE(
            f32x4 v = RT_ACC4;
            v += *reinterpret_cast<const f32x4*>(a.bo + n);
            v += qv[j][g];
            if (MODE == 0) *reinterpret_cast<f32x4*>(a.y_out + (long long)m * d + n) = v;
            keep[pass][j][g] = v;)
        }
      publish(K1, N0 / BN);
      {   // LayerNorm over the image: 16 rows per compute wave, in place; f (and the statistics) also go to memory
        const int rs = K1 * 4;
        f32x4 ww = zero4, bb = zero4;
        if (on) { ww = *reinterpret_cast<const f32x4*>(a.ln_w + c4); bb = *reinterpret_cast<const f32x4*>(a.ln_b + c4); }
#pragma unroll 1
        for (int r0 = 0; r0 < 16; r0 += 8) {
          f32x4 v[8];
          float mu[8], rsd[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = on ? *reinterpret_cast<const f32x4*>(A + a_off(cw * 16 + r0 + r, lane, rs)) : zero4;
          ln8(v, on, d, a.eps, ww, bb, mu, rsd);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int lr = cw * 16 + r0 + r, mm = m0 + lr;
            if (on) {
              *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = v[r];
              *reinterpret_cast<f32x4*>(a.f + (long long)mm * d + c4) = v[r];
            }
            if (MODE == 0 && lane == 0) { a.mean[mm] = mu[r]; a.rstd[mm] = rsd[r]; }
          }
        }
        lds_sync();                     // X3
      }
    }
    // ---- [1] hdrop = drop(relu(f W1^T + b1))
#pragma unroll
    for (int pass = 0; pass < NPMAX; ++pass)
      if (pass < N1 / BN) {
        compute_pass<BTR>(A, K1 * 4, ring, slot, K1 / BK, lane, wm, wn, acc, a.probe);
        RT_FOR_TILE(
          f32x4 v = RT_ACC4;
          v += *reinterpret_cast<const f32x4*>(a.b1 + n);
          _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          if (a.p > 0.f) v = rt_drop4(v, a.seed_h, a.sid_h, ((unsigned long long)m * dff + n) >> 2, a.p, inv_keep);
          if (MODE == 0 && !RT_PROBEA(2)) *reinterpret_cast<f32x4*>(a.hdrop + (long long)m * dff + n) = v;
          keep[pass][j][g] = v;)
      }
    publish(K2, N1 / BN);
    // ---- [2] out = f + drop(hdrop W2^T + b2)
#pragma unroll 1
    for (int pass = 0; pass < N2 / BN; ++pass) {
      compute_pass<BTR>(A, K2 * 4, ring, slot, K2 / BK, lane, wm, wn, acc, a.probe);
      if (RT_PROBEA(4)) { if (acc[0][0] == 12345.f) a.out[0] = acc[1][3]; continue; }
      f32x4 fv[2][4];
      RT_FOR_TILE(fv[j][g] = *reinterpret_cast<const f32x4*>(a.f + (long long)m * d + n);)
      RT_FOR_TILE(
        f32x4 v = RT_ACC4;
        v += *reinterpret_cast<const f32x4*>(a.b2 + n);
        if (a.p > 0.f) v = rt_drop4(v, a.seed_o, a.sid_o, ((unsigned long long)m * d + n) >> 2, a.p, inv_keep);
        v += fv[j][g];
        *reinterpret_cast<f32x4*>(a.out + (long long)m * d + n) = v;)
    }
  } else {
    // ---- [0] g_h = [hdrop != 0] / keep * (g_o W2)
#pragma unroll
    for (int pass = 0; pass < NPMAX; ++pass)
      if (pass < N0 / BN) {
        compute_pass<BTR>(A, K0 * 4, ring, slot, K0 / BK, lane, wm, wn, acc, a.probe);
        f32x4 hv[2][4];
        RT_FOR_TILE(hv[j][g] = *reinterpret_cast<const f32x4*>(a.hd + (long long)m * dff + n);)
        RT_FOR_TILE(
          f32x4 v;
          _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = hv[j][g][e] != 0.f ? acc[j][4 * g + e] * inv_keep : 0.f;
          if (!RT_PROBEA(2)) *reinterpret_cast<f32x4*>(a.g_h + (long long)m * dff + n) = v;
          keep[pass][j][g] = v;)
      }
    publish(K1, N0 / BN);
    // ---- [1] g_f = g_h W1 + g_out
#pragma unroll
    for (int pass = 0; pass < NPMAX; ++pass)
      if (pass < N1 / BN) {
        compute_pass<BTR>(A, K1 * 4, ring, slot, K1 / BK, lane, wm, wn, acc, a.probe);
        if (RT_PROBEA(4)) { if (acc[0][0] == 12345.f) a.g_f[0] = acc[1][3]; continue; }
        f32x4 rv[2][4];
        RT_FOR_TILE(rv[j][g] = *reinterpret_cast<const f32x4*>(a.g_out + (long long)m * d + n);)
        RT_FOR_TILE(
          f32x4 v = RT_ACC4;
          v += rv[j][g];
          if (last == 2) *reinterpret_cast<f32x4*>(a.g_f + (long long)m * d + n) = v;
          keep[pass][j][g] = v;)
      }
    if (last == 3) {
      publish(K2, N1 / BN);
      {   // LayerNorm backward over the image (layernorm_bwd_kernel's arithmetic): g_y in place and to memory, this workgroup's
          // share of d ln_w / d ln_b to partial[blockIdx][2][d] (summed by rt_layernorm_bwd_reduce)
        const int rs = K2 * 4;
        const f32x4 wv = on ? *reinterpret_cast<const f32x4*>(a.ln_w + c4) : zero4;
        f32x4 pw = zero4, pb = zero4;
#pragma unroll 1
        for (int r0 = 0; r0 < 16; r0 += 8) {
          f32x4 gq[8], xh[8];
          float mu[8], rsd[8], s1[8], s2[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int lr = cw * 16 + r0 + r, mm = m0 + lr;
            mu[r] = a.mean[mm]; rsd[r] = a.rstd[mm];
            gq[r] = on ? *reinterpret_cast<const f32x4*>(A + a_off(lr, lane, rs)) : zero4;
            xh[r] = on ? *reinterpret_cast<const f32x4*>(a.y + (long long)mm * d + c4) : zero4;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            xh[r] = on ? (xh[r] - mu[r]) * rsd[r] : zero4;
            const f32x4 gw = gq[r] * wv;
            s1[r] = 0.f + (gw[0] + gw[1] + gw[2] + gw[3]);
            s2[r] = 0.f + (gw[0] * xh[r][0] + gw[1] * xh[r][1] + gw[2] * xh[r][2] + gw[3] * xh[r][3]);
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 8; ++r) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int lr = cw * 16 + r0 + r, mm = m0 + lr;
            s1[r] = s1[r] / d; s2[r] = s2[r] / d;
            if (on) {
              const f32x4 o4 = (gq[r] * wv - s1[r] - xh[r] * s2[r]) * rsd[r];
              *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = o4;
              *reinterpret_cast<f32x4*>(a.g_y + (long long)mm * d + c4) = o4;
              pw += gq[r] * xh[r];
              pb += gq[r];
            }
          }
        }
        if (on) {
          *reinterpret_cast<f32x4*>(red + (cw * 2 + 0) * d + c4) = pw;
          *reinterpret_cast<f32x4*>(red + (cw * 2 + 1) * d + c4) = pb;
        }
        lds_sync();                     // X3: g_y rows and the four waves' partials are in the LDS
        float* part = a.ln_partial + (long long)blockIdx.x * 2 * d;
        for (int c = cw * 64 + lane; c < d; c += 256) {
          part[c] = (red[0 * d + c] + red[2 * d + c]) + (red[4 * d + c] + red[6 * d + c]);
          part[d + c] = (red[1 * d + c] + red[3 * d + c]) + (red[5 * d + c] + red[7 * d + c]);
        }
      }
      // ---- [2] g_A = g_y Wo
#pragma unroll 1
      for (int pass = 0; pass < N2 / BN; ++pass) {
        compute_pass<BTR>(A, K2 * 4, ring, slot, K2 / BK, lane, wm, wn, acc, a.probe);
        RT_FOR_TILE(*reinterpret_cast<f32x4*>(a.g_A + (long long)m * d + n) = RT_ACC4;)
      }
    }
  }
#undef RT_FOR_TILE
#undef RT_ACC4
}

size_t lds_bytes(int d, int dff) { return (size_t)BM * (d > dff ? d : dff) * 4 + (size_t)NSTG * W_STAGE_B + RED_B; }
bool shape_ok(int M, int d, int dff, long long plane_stride) {
  return M > 0 && M % BM == 0 && d % BN == 0 && dff % BN == 0 && d <= 256 && dff <= NPMAX * BN && lds_bytes(d, dff) <= 160 * 1024 &&
         (plane_stride & 7) == 0;
}
bool mis(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

template <int MODE>
int launch(const FfnArgs& a, hipStream_t stream) {
  const size_t lds = lds_bytes(a.d, a.dff);
#ifdef RT_ABLATION_BUILD
  { const char* e = getenv("RT_FFN_PROBE"); const_cast<FfnArgs&>(a).probe = e ? atoi(e) : 0; }
#endif
  auto kern = &ffn_kernel<MODE>;
  RT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<a.M / BM, GT, lds, stream>>>(a);
  RT_CHECK_LAUNCH();
  return RT_OK;
}

}  // namespace

extern "C" {

// 1 when the row-resident chain kernels serve the shape (rows a multiple of 64, d and dff 128 or 256: the operand rows of a product stay
// in the LDS): the block executor asks once per call and takes the separate launches otherwise.
int rt_ffn_fused_supported(int32_t M, int32_t d, int32_t dff) { return shape_ok(M, d, dff, 0) ? 1 : 0; }

// Forward of the feed-forward half on M packed rows: f = LN(y; ln_w, ln_b, eps) (+ mean, rstd [M]), hdrop [M, dff] = drop(relu(f W1^T +
// b1)) (stream seed_h / sid_h), out [M, d] = f + drop(hdrop W2^T + b2) (stream seed_o / sid_o).  w1_planes / w2_planes: the bf16 planes
// of W1 [dff, d] / W2 [d, dff] (rt_split_planes; plane q at + q * plane_stride elements).  p = 0: no dropout.  All arrays contiguous,
// 16-byte aligned.  RT_ERR_UNSUPPORTED for other shapes.
int rt_ffn_fused_fwd(const float* y, const float* ln_w, const float* ln_b, float eps, float* f, float* mean, float* rstd,
                     const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, const float* b1, const float* b2, float* hdrop,
                     float* out, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_h, uint64_t sid_h, uint64_t seed_o, uint64_t sid_o,
                     hipStream_t stream) {
  (void)hipGetLastError();
  if (y == nullptr || ln_w == nullptr || ln_b == nullptr || f == nullptr || mean == nullptr || rstd == nullptr || w1_planes == nullptr ||
      w2_planes == nullptr || b1 == nullptr || b2 == nullptr || hdrop == nullptr || out == nullptr || p < 0.f || p >= 1.f)
    return RT_ERR_INVALID_ARG;
  if (!shape_ok(M, d, dff, plane_stride)) return RT_ERR_UNSUPPORTED;
  if (mis(y) || mis(ln_w) || mis(ln_b) || mis(f) || mis(w1_planes) || mis(w2_planes) || mis(b1) || mis(b2) || mis(hdrop) || mis(out))
    return RT_ERR_INVALID_ARG;
  FfnArgs a{};
  a.M = M; a.d = d; a.dff = dff; a.p = p; a.eps = eps; a.first = 1; a.last = 3;
  a.seed_h = seed_h; a.sid_h = sid_h; a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.wop = w1_planes; a.plane_stride = plane_stride;
  a.y = y; a.ln_w = ln_w; a.ln_b = ln_b; a.b1 = b1; a.b2 = b2; a.f = f; a.mean = mean; a.rstd = rstd; a.hdrop = hdrop; a.out = out;
  return launch<0>(a, stream);
}

// Backward: g_o [M, d] = drop'(g_out) (written only when p > 0; with p = 0 the caller's g_out IS g_o), g_h [M, dff] =
// [hdrop != 0] / (1 - p) * (g_o W2), g_f [M, d] = g_h W1 + g_out — the gradient with respect to f = LN2(y) including the skip branch.
// The weight gradients (dW2 = g_o^T hdrop, dW1 = g_h^T f) and the LayerNorm backward stay with the caller.
int rt_ffn_fused_bwd(const float* g_out, const float* hdrop, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride,
                     float* g_o, float* g_h, float* g_f, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o, uint64_t sid_o,
                     hipStream_t stream) {
  (void)hipGetLastError();
  if (g_out == nullptr || hdrop == nullptr || w1_planes == nullptr || w2_planes == nullptr || g_h == nullptr || g_f == nullptr ||
      (p > 0.f && g_o == nullptr) || p < 0.f || p >= 1.f)
    return RT_ERR_INVALID_ARG;
  if (!shape_ok(M, d, dff, plane_stride)) return RT_ERR_UNSUPPORTED;
  if (mis(g_out) || mis(hdrop) || mis(w1_planes) || mis(w2_planes) || mis(g_o) || mis(g_h) || mis(g_f)) return RT_ERR_INVALID_ARG;
  FfnArgs a{};
  a.M = M; a.d = d; a.dff = dff; a.p = p; a.first = 0; a.last = 2;
  a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.wop = w1_planes; a.plane_stride = plane_stride;
  a.g_out = g_out; a.hd = hdrop; a.g_o = g_o; a.g_h = g_h; a.g_f = g_f;
  return launch<1>(a, stream);
}

// The whole tail of a packed SASRec block behind its attention (sasrec.py:224-229) as ONE launch:
//   y = q + attn Wo^T + bo;  f = LN2(y);  hdrop = drop(relu(f W1^T + b1));  out = f + drop(hdrop W2^T + b2)
// training != 0: y, f, mean, rstd, hdrop are written (what rt_block_tail_bwd reads).  training == 0 (recommend(): p must be 0): only `out`
// and the scratch rows `f` are written — y, mean, rstd, hdrop may be NULL; a row then crosses the memory pipe three times (attn and q in,
// out out) instead of twelve.  wo_planes: the planes of the out-projection's weight [d, d], same plane_stride as w1 / w2.
int rt_block_tail_fwd(const float* attn, const float* q, const uint16_t* wo_planes, const float* bo, const float* ln_w, const float* ln_b,
                      float eps, float* y, float* f, float* mean, float* rstd, const uint16_t* w1_planes, const uint16_t* w2_planes,
                      int64_t plane_stride, const float* b1, const float* b2, float* hdrop, float* out, int32_t M, int32_t d, int32_t dff, float p,
                      uint64_t seed_h, uint64_t sid_h, uint64_t seed_o, uint64_t sid_o, int32_t training, hipStream_t stream) {
  (void)hipGetLastError();
  if (attn == nullptr || q == nullptr || wo_planes == nullptr || bo == nullptr || ln_w == nullptr || ln_b == nullptr || f == nullptr ||
      w1_planes == nullptr || w2_planes == nullptr || b1 == nullptr || b2 == nullptr || out == nullptr || p < 0.f || p >= 1.f)
    return RT_ERR_INVALID_ARG;
  if (training ? (y == nullptr || mean == nullptr || rstd == nullptr || hdrop == nullptr) : p != 0.f) return RT_ERR_INVALID_ARG;
  if (!shape_ok(M, d, dff, plane_stride)) return RT_ERR_UNSUPPORTED;
  if (mis(attn) || mis(q) || mis(wo_planes) || mis(bo) || mis(ln_w) || mis(ln_b) || mis(y) || mis(f) || mis(w1_planes) || mis(w2_planes) ||
      mis(b1) || mis(b2) || mis(hdrop) || mis(out))
    return RT_ERR_INVALID_ARG;
  FfnArgs a{};
  a.M = M; a.d = d; a.dff = dff; a.p = p; a.eps = eps; a.first = 0; a.last = 3;
  a.seed_h = seed_h; a.sid_h = sid_h; a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.wop = wo_planes; a.plane_stride = plane_stride;
  a.attn = attn; a.q = q; a.bo = bo; a.y_out = y; a.ln_w = ln_w; a.ln_b = ln_b; a.b1 = b1; a.b2 = b2; a.f = f; a.mean = mean; a.rstd = rstd;
  a.hdrop = hdrop; a.out = out;
  return training ? launch<0>(a, stream) : launch<2>(a, stream);
}

size_t rt_block_tail_partial_floats(int32_t M, int32_t d) { return M > 0 && d > 0 ? (size_t)(M / BM) * 2 * (size_t)d : 0; }

// Its backward down to the attention's output: g_o (p > 0 only), g_h as rt_ffn_fused_bwd; then, with the rows still in the LDS, the
// LayerNorm backward g_y = LN2'(g_f; y, mean, rstd, ln_w) — g_f itself never leaves the chip — and g_A [M, d] = g_y Wo.  ln_partial
// [(M / 64) * 2 * d floats]: every workgroup's share of d ln_w / d ln_b; rt_layernorm_bwd_reduce(ln_partial, M / 64, d, ...) sums them.
// The three weight gradients (g_o^T hdrop, g_h^T f, g_y^T attn) stay with the caller.
int rt_block_tail_bwd(const float* g_out, const float* hdrop, const float* y, const float* mean, const float* rstd, const float* ln_w,
                      const uint16_t* wo_planes, const uint16_t* w1_planes, const uint16_t* w2_planes, int64_t plane_stride, float* g_o,
                      float* g_h, float* g_y, float* g_A, float* ln_partial, int32_t M, int32_t d, int32_t dff, float p, uint64_t seed_o,
                      uint64_t sid_o, hipStream_t stream) {
  (void)hipGetLastError();
  if (g_out == nullptr || hdrop == nullptr || y == nullptr || mean == nullptr || rstd == nullptr || ln_w == nullptr || wo_planes == nullptr ||
      w1_planes == nullptr || w2_planes == nullptr || g_h == nullptr || g_y == nullptr || g_A == nullptr || ln_partial == nullptr ||
      (p > 0.f && g_o == nullptr) || p < 0.f || p >= 1.f)
    return RT_ERR_INVALID_ARG;
  if (!shape_ok(M, d, dff, plane_stride)) return RT_ERR_UNSUPPORTED;
  if (mis(g_out) || mis(hdrop) || mis(y) || mis(ln_w) || mis(wo_planes) || mis(w1_planes) || mis(w2_planes) || mis(g_o) || mis(g_h) ||
      mis(g_y) || mis(g_A) || mis(ln_partial))
    return RT_ERR_INVALID_ARG;
  FfnArgs a{};
  a.M = M; a.d = d; a.dff = dff; a.p = p; a.first = 0; a.last = 3;
  a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.wop = wo_planes; a.plane_stride = plane_stride;
  a.g_out = g_out; a.hd = hdrop; a.y = y; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.ln_w = ln_w;
  a.g_o = g_o; a.g_h = g_h; a.g_y = g_y; a.g_A = g_A; a.ln_partial = ln_partial;
  return launch<1>(a, stream);
}

}  // extern "C"
