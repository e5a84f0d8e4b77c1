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
  float *f, *mean, *rstd, *hdrop, *out;
  // backward
  const float *g_out, *hd;
  float *g_o, *g_h, *g_f;
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

// ---- loader side: the weight stream of both products as ONE sequence of stages ----------------------------------------------------------
struct WStream {
  const unsigned short* src[2][2];   // [product][piece]: this lane's 16-byte unit of pieces 2 lw, 2 lw + 1 at (pass 0, k 0)
  long long ldw[2];
  int KS[2], T[2];
  long long plane_stride;
  unsigned base;                     // LDS byte address of this loader wave's first piece in ring slot 0
  int g, kk, pass, slot, issued;
  bool no_dma;
};
template <bool BTR>
__device__ __forceinline__ void wstream_init(WStream& w, int lw, int lane, const unsigned short* Wa, long long ldwa, int Ka, int Na,
                                             const unsigned short* Wb, long long ldwb, int Kb, int Nb, long long plane_stride, unsigned ring_lds) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const unsigned short* W = g == 0 ? Wa : Wb;
    const long long ldw = g == 0 ? ldwa : ldwb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = (lw * 2 + j) * 64 + lane;                    // 16-byte unit of the plane tile
      if (!BTR) {
        const int row = q >> 2, c = (q & 3) ^ ((row >> 2) & 3);   // [128 n][4 units]: unit c of row n stored at c ^ ((n>>2)&3)
        w.src[g][j] = W + (long long)row * ldw + c * 8;
      } else {
        const int row = q >> 4, u = (q & 15) ^ ((row & 3) << 2);  // [32 k][16 units]: unit u of row k stored at u ^ ((k&3)<<2)
        w.src[g][j] = W + (long long)row * ldw + u * 8;
      }
    }
    w.ldw[g] = ldw;
  }
  w.KS[0] = Ka / BK; w.T[0] = w.KS[0] * (Na / BN);
  w.KS[1] = Kb / BK; w.T[1] = w.KS[1] * (Nb / BN);
  w.plane_stride = plane_stride;
  w.base = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(lw * 2 * 1024));
  w.g = 0; w.kk = 0; w.pass = 0; w.slot = 0; w.issued = 0; w.no_dma = false;
}
template <bool BTR>
__device__ __forceinline__ void wstream_issue(WStream& w) {
  const int g = w.g;
  const long long ldw = g == 0 ? w.ldw[0] : w.ldw[1];
  const int KSg = g == 0 ? w.KS[0] : w.KS[1], Tg = g == 0 ? w.T[0] : w.T[1];
  const long long bo = BTR ? (long long)w.kk * BK * ldw + (long long)w.pass * BN : (long long)w.pass * BN * ldw + (long long)w.kk * BK;
  const unsigned sb = w.base + (unsigned)(w.slot * W_STAGE_B);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (!w.no_dma) dma16((g == 0 ? w.src[0][j] : w.src[1][j]) + bo + pl * w.plane_stride, sb + pl * P_TILE_B + j * 1024);
  ++w.issued;
  if (++w.slot == NSTG) w.slot = 0;
  (void)0;
  if (++w.kk == KSg) {
    w.kk = 0;
    if ((w.pass + 1) * KSg == Tg) { w.pass = 0; w.g = 1; } else ++w.pass;
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

template <int MODE>   // 0: forward, 1: backward
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
  // products: forward  (A = f, K = d, N = dff, W1 as [n][k]) then (A = hdrop, K = dff, N = d, W2 as [n][k])
  //           backward (A = g_o, K = d, N = dff, W2 [d, dff] as [k][n]) then (A = g_h, K = dff, N = d, W1 [dff, d] as [k][n])
  constexpr bool BTR = MODE == 1;
  const int K1 = d, N1 = dff, K2 = dff, N2 = d;
  const int NP1 = N1 / BN, NP2 = N2 / BN, KS1 = K1 / BK, KS2 = K2 / BK;

  WStream ws;
  if (loader) {    // the ring runs ahead of the prologue: the weights do not depend on it
    if (MODE == 0) wstream_init<BTR>(ws, cw, lane, a.w1p, d, K1, N1, a.w2p, dff, K2, N2, a.plane_stride, lds_addr(ring));
    else wstream_init<BTR>(ws, cw, lane, a.w2p, dff, K1, N1, a.w1p, d, K2, N2, a.plane_stride, lds_addr(ring));
#ifdef RT_ABLATION_BUILD
    ws.no_dma = (a.probe & 16) != 0;
#endif
    wstream_issue<BTR>(ws);
    wstream_issue<BTR>(ws);
  }

  // ---- prologue over the workgroup's 64 rows (8 per wave): the rows go to memory (saved for the other direction) AND into the LDS image
  {
    const int rs = K1 * 4;
    f32x4 v[8];
    const int c = lane * 4;
    const bool on = c < d;                                   // d = 128: half of the lanes hold a row's columns
    const float* src = MODE == 0 ? a.y : a.g_out;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v[r] = on ? *reinterpret_cast<const f32x4*>(src + (long long)(m0 + wave * 8 + r) * d + c) : z;
    }
    if (MODE == 0) {
      // f = LN2(y): layernorm_fwd_kernel's arithmetic (sum -> mean -> centred squares -> rstd -> (v - mu) rs w + b), 8 rows interleaved
      float s[8], q[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) s[r] = 0.f + (v[r][0] + v[r][1] + v[r][2] + v[r][3]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < 8; ++r) s[r] += __shfl_xor(s[r], o, 64);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        s[r] = s[r] / d;                                     // mu
        q[r] = 0.f;
        if (on) {
          const f32x4 t = v[r] - s[r];
          q[r] += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < 8; ++r) q[r] += __shfl_xor(q[r], o, 64);
      f32x4 ww = {0.f, 0.f, 0.f, 0.f}, bb = ww;
      if (on) { ww = *reinterpret_cast<const f32x4*>(a.ln_w + c); bb = *reinterpret_cast<const f32x4*>(a.ln_b + c); }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int lr = wave * 8 + r, m = m0 + lr;
        const float rs_ = 1.0f / sqrtf(q[r] / d + a.eps);
        if (on) {
          const f32x4 fv = (v[r] - s[r]) * rs_ * ww + bb;
          if (!RT_PROBEA(1)) *reinterpret_cast<f32x4*>(a.f + (long long)m * d + c) = fv;
          *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = fv;
        }
        if (lane == 0) { a.mean[m] = s[r]; a.rstd[m] = rs_; }
      }
    } else {
      // g_o = drop'(g_out) with the mask of the forward's output dropout (stream seed_o / sid_o, keyed by the float4 group of [M, d])
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int lr = wave * 8 + r, m = m0 + lr;
        if (on) {
          f32x4 gv = v[r];
          if (a.p > 0.f) {
            gv = rt_drop4(gv, a.seed_o, a.sid_o, ((unsigned long long)m * d + c) >> 2, a.p, inv_keep);
            if (!RT_PROBEA(1)) *reinterpret_cast<f32x4*>(a.g_o + (long long)m * d + c) = gv;
          }
          *reinterpret_cast<f32x4*>(A + a_off(lr, lane, rs)) = gv;
        }
      }
    }
  }
  if (loader) wait_vmcnt<0>();       // the loaders' counters start clean: from here on they hold ring pieces only (stages 0 and 1 have landed)
  __syncthreads();                   // B0: the operand rows of the first product are in the LDS

  if (loader) {
    // ---- weight stream: stage gi is published by barrier S_gi; stage gi + 2 is issued right behind it into the slot stage gi - 1 left
    const int T1 = ws.T[0], T = ws.T[0] + ws.T[1];
#pragma unroll 1
    for (int gi = 0; gi < T; ++gi) {
      if (gi + 1 < ws.issued) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();      // everything but the youngest stage has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ws.issued < T) wstream_issue<BTR>(ws);
      if (gi == T1 - 1) {            // between the products the compute waves rewrite the operand image (barriers X1, X2); the ring keeps flying
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    return;
  }

  // ---- compute waves -------------------------------------------------------------------------------------------------------------------
  const int m = m0 + wm * 32 + col;     // the activation row of this lane in every epilogue
  const int lrow = wm * 32 + col;
  int slot = 0;
  f32x16 acc[2];
  f32x4 keep[NPMAX][2][4];              // the first product's results (all passes): the second product's operand rows
#pragma unroll
  for (int pass = 0; pass < NPMAX; ++pass) {
    if (pass < NP1) {
      compute_pass<BTR>(A, K1 * 4, ring, slot, KS1, lane, wm, wn, acc, a.probe);
      f32x4 x4[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
          x4[j][g] = MODE == 0 ? *reinterpret_cast<const f32x4*>(a.b1 + n) : *reinterpret_cast<const f32x4*>(a.hd + (long long)m * dff + n);
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
          f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
          if (MODE == 0) {              // hdrop = drop(relu(f W1^T + b1))
            v += x4[j][g];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (a.p > 0.f) v = rt_drop4(v, a.seed_h, a.sid_h, ((unsigned long long)m * dff + n) >> 2, a.p, inv_keep);
            if (!RT_PROBEA(2)) *reinterpret_cast<f32x4*>(a.hdrop + (long long)m * dff + n) = v;
          } else {                      // g_h = [hdrop != 0] / keep * (g_o W2)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x4[j][g][e] != 0.f ? v[e] * inv_keep : 0.f;
            if (!RT_PROBEA(2)) *reinterpret_cast<f32x4*>(a.g_h + (long long)m * dff + n) = v;
          }
          keep[pass][j][g] = v;
        }
    }
  }
  // X1: every compute wave has read its last operand fragment of the first product.  LDS-only synchronisation: the epilogue's global
  // stores keep draining under the second product (a __syncthreads would wait for them)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    const int rs = K2 * 4;
#pragma unroll
    for (int pass = 0; pass < NPMAX; ++pass)
      if (pass < NP1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
            *reinterpret_cast<f32x4*>(A + a_off(lrow, n >> 2, rs)) = keep[pass][j][g];
          }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // X2: the second product's operand rows are in the LDS
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int pass = 0; pass < NP2; ++pass) {
    compute_pass<BTR>(A, K2 * 4, ring, slot, KS2, lane, wm, wn, acc, a.probe);
    if (RT_PROBEA(4)) { if (acc[0][0] == 12345.f) a.mean[0] = acc[1][3]; continue; }
    const float* res = MODE == 0 ? a.f : a.g_out;      // the skip branch: f (forward), g_out (backward)
    f32x4 rv[2][4], bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
        rv[j][g] = *reinterpret_cast<const f32x4*>(res + (long long)m * d + n);
        if (MODE == 0) bv[j][g] = *reinterpret_cast<const f32x4*>(a.b2 + n);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = pass * BN + wn * 64 + j * 32 + 8 * g + 4 * half;
        f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
        if (MODE == 0) {                // out = f + drop(hdrop W2^T + b2)
          v += bv[j][g];
          if (a.p > 0.f) v = rt_drop4(v, a.seed_o, a.sid_o, ((unsigned long long)m * d + n) >> 2, a.p, inv_keep);
          v += rv[j][g];
          *reinterpret_cast<f32x4*>(a.out + (long long)m * d + n) = v;
        } else {                        // g_f = g_h W1 + g_out
          v += rv[j][g];
          *reinterpret_cast<f32x4*>(a.g_f + (long long)m * d + n) = v;
        }
      }
  }
}

size_t lds_bytes(int d, int dff) { return (size_t)BM * (d > dff ? d : dff) * 4 + (size_t)NSTG * W_STAGE_B; }
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

// 1 when rt_ffn_fused_fwd / _bwd serve the shape (rows a multiple of 64, d and dff 128 or 256: the operand rows of a product stay in the LDS): the block executor
// asks once per call and takes the five-launch sequence otherwise.
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
  a.M = M; a.d = d; a.dff = dff; a.p = p; a.eps = eps;
  a.seed_h = seed_h; a.sid_h = sid_h; a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.plane_stride = plane_stride;
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
  a.M = M; a.d = d; a.dff = dff; a.p = p;
  a.seed_o = seed_o; a.sid_o = sid_o;
  a.w1p = w1_planes; a.w2p = w2_planes; a.plane_stride = plane_stride;
  a.g_out = g_out; a.hd = hdrop; a.g_o = g_o; a.g_h = g_h; a.g_f = g_f;
  return launch<1>(a, stream);
}

}  // extern "C"
