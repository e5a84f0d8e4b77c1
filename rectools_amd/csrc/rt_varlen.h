// Shared by the packed (variable-length) attention kernels: rt_attention_varlen.hip (f32-input MFMA, first form) and
// rt_attention_v2.hip (bf16 planes, 16x16x32 MFMA).  See rt_attention_varlen.hip for the layout of packed sessions.
#pragma once
#include "rt_common.h"

namespace rt_varlen {

struct VarlenArgs {
  const float* q; const float* k; const float* v; long long ldq, ldk, ldv;
  float* o; long long ldo;
  const long long* cu;                 // [B+1] first packed row of every session
  const float* bk; const float* bv;    // [H*hd] key / value projection biases = the pad key / value row; null: no pad keys
  int B, H, hd, window;                // window = the reference's session_max_len: n_pad = window - n_b
  float scale;                         // 1 / sqrt(hd)
  // training
  float p_drop; unsigned long long seed;
  float* lse;                          // [N, H] log-sum-exp of every query (written by the forward when not null)
  const float* dout; long long lddo;   // backward input
  float* delta;                        // [N, H] rowsum(dO * O): written by the dQ kernel, read by the dK/dV kernel
  float* dq; float* dk; float* dv; long long lddq, lddk, lddv;
  float* dbv_part;                     // [B, H*hd] per-session partial of the value-bias gradient from the pad keys (or null)
  int ablate;                          // read by -DRT_ABLATION_BUILD builds of rt_attention_v2.hip only (RT_V2_ABLATE), else 0
  int uniform_len;                     // cu == NULL (rt_attention_v3.hip only): every session is `uniform_len` rows, session b starts at b * uniform_len
  int prefix_sessions;                 // > 0 (rt_attention_v3.hip, causal): sessions 0 .. prefix_sessions - 1 sit behind the shared pad prefix, session
                                       // number prefix_sessions (`window` rows); a session of n rows sees the prefix's first window - n rows as keys
  float* prefix_ws;                    // backward: [groups][window][2 * H * hd] partials of the prefix rows' dK | dV from the sessions behind it
};

// Attention dropout mask, same construction as rt_attention.hip: ONE 32-bit mix per (head, query, PAIR of adjacent keys), 16 bits
// per key compared with p * 65536.  Queries and keys are numbered inside their session; the window's pad keys take the key
// numbers n .. n + n_pad - 1 behind the real ones.  The three kernels regenerate the same masks.
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, unsigned bh, unsigned q, unsigned key_pair) {
  unsigned x = (unsigned)seed ^ (q * 0x9E3779B1u) ^ (key_pair * 0x85EBCA77u) ^ (bh * 0xC2B2AE3Du) ^ (unsigned)(seed >> 32);
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned drop_thr16(float p) { return (unsigned)(p * 65536.0f); }
__device__ __forceinline__ bool drop_kept(unsigned long long seed, unsigned bh, unsigned q, unsigned key, unsigned thr16) {
  return ((drop_hash(seed, bh, q, key >> 1) >> (16u * (key & 1u))) & 0xFFFFu) >= thr16;
}
// how many of the n_pad pad keys of query q survive the dropout
__device__ __forceinline__ int pads_kept(unsigned long long seed, unsigned bh, unsigned q, int n, int n_pad, unsigned thr16) {
  int kept = 0;
  for (int kk = n; kk < n + n_pad; ++kk) kept += drop_kept(seed, bh, q, (unsigned)kk, thr16) ? 1 : 0;
  return kept;
}


// K6v2 (rt_attention_v2.hip): HSTU's pointwise attention silu(q k^T + rab) / L over packed sessions on the bf16-plane geometry of K4v2.
struct HstuV2Args {
  const float* q; const float* k; const float* v; long long ldq, ldk, ldv;
  float* o; long long ldo;
  const long long* cu;                 // [B+1] first packed row of every session
  const long long* ts;                 // session b's n_b + 1 timestamps start at ts[cu[b] + b] (null: no time bias)
  const float* time_w; const long long* time_thr; const float* pos_w;   // [n], [148] (thresholds, then n), [2 Lw - 1] (pos_w null: no position bias)
  int B, H, hd, Lw;                    // Lw = the window (session_max_len): 1 / Lw and the position table's centre
  const float* dout; long long lddo;
  float* dq; float* dk; float* dv; long long lddq, lddk, lddv;
  float* d_time_w; float* d_pos_w;     // accumulators (atomics; the caller zero-fills)
};

}  // namespace rt_varlen

// K6v2: RT_ERR_UNSUPPORTED for head sizes other than 32 / 64 or a window whose tables do not fit the LDS beside two 192-row images.
int rt_v2_hstu_fwd(const rt_varlen::HstuV2Args& a, hipStream_t stream);
int rt_v2_hstu_bwd(const rt_varlen::HstuV2Args& a, hipStream_t stream);

// rt_attention_v2.hip: the bf16-plane kernels behind the same entry points (hd 32 / 64, one (session, head) image within 160 KB of LDS).
// Each returns RT_ERR_UNSUPPORTED when the shape does not fit (the caller then takes the first-form kernels).
int rt_v2_varlen_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream);
int rt_v2_varlen_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream);
int rt_v2_bidir_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream);
int rt_v2_bidir_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream);

// rt_attention_v3.hip: the same kernels' arithmetic on STREAMED 64-row chunks (a workgroup = 64 owner rows of a (session, head)); hd 32 /
// 64, any session length.
int rt_v3_varlen_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream);
int rt_v3_varlen_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream);
int rt_v3_bidir_fwd(const rt_varlen::VarlenArgs& a, int max_len, bool train, hipStream_t stream);
int rt_v3_bidir_bwd(const rt_varlen::VarlenArgs& a, int max_len, hipStream_t stream);
int rt_v3_hstu_fwd(const rt_varlen::HstuV2Args& a, hipStream_t stream);
int rt_v3_hstu_bwd(const rt_varlen::HstuV2Args& a, hipStream_t stream);
size_t rt_v3_prefix_workspace_floats(int window, int H, int hd);      // VarlenArgs.prefix_ws of rt_v3_varlen_bwd with prefix_sessions > 0
