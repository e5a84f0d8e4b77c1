"""ctypes binding of `librectools_hip.so` (C ABI: include/rectools_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module raises.
PyTorch is used for device memory and streams only: tensors are passed as raw `data_ptr()`s and kernels
are enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import typing as tp

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH: another build of the SAME library (diagnostic builds: -DRT_ABLATION_BUILD / -DRT_ATTN_TRACE, scripts/gpu/ablate.sh)
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(PKG_DIR, "librectools_hip.so")

RT_OK, RT_ERR_INVALID_ARG, RT_ERR_WORKSPACE, RT_ERR_LAUNCH, RT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4

c_i32, c_i64, c_f32, c_sz, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
c_u64 = ctypes.c_uint64
c_f64 = ctypes.c_double

# name -> (restype, argtypes); MUST list every symbol include/rectools_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
SIGNATURES: tp.Dict[str, tp.Tuple[tp.Any, tp.List[tp.Any]]] = {
    "rt_version": (c_i32, []),
    "rt_device_cu_count": (c_i32, []),
    "rt_last_error": (ctypes.c_char_p, []),
    "rt_topk_workspace_bytes": (c_sz, [c_i32, c_i64, c_i32, c_i32]),
    "rt_filter_hash_bytes": (c_sz, [c_i32, c_i64]),
    "rt_filter_hash_build": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "rt_topk_score": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "rt_to_hm_rows": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "rt_one_plane_to_fragments": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "rt_topk_two_stage_workspace_bytes": (c_sz, [c_i32, c_i64, c_i32, c_i32, c_i32]),
    "rt_topk_score_two_stage": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "rt_gemm_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "rt_gemm": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "rt_wgrad_grouped_workspace_bytes": (c_sz, [c_vp, c_i32, c_i32, c_i32]),
    "rt_wgrad_grouped": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "rt_split_planes": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    "rt_gemm_wp": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "rt_ffn_fused_supported": (c_i32, [c_i32, c_i32, c_i32]),
    "rt_ffn_fused_fwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_u64, c_u64, c_vp]),
    "rt_ffn_fused_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp]),
    "rt_block_tail_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_u64, c_u64, c_i32, c_vp]),
    "rt_block_tail_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp]),
    "rt_block_tail_partial_floats": (c_sz, [c_i32, c_i32]),
    "rt_layernorm_bwd_reduce": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rt_gemm_grouped": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    "rt_colsum": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "rt_collate": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_f32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_collate_packed": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_collate_packed_ts": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "rt_collate_packed_bert": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_sample_negatives": (c_i32, [c_i64, c_i64, c_i64, c_u64, c_u64, c_vp, c_vp]),
    "rt_bag_sum_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_vp]),
    "rt_bag_sum_bwd_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "rt_bag_sum_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_vp, c_sz, c_vp]),
    "rt_embed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_vp]),
    "rt_embed_bwd_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "rt_embed_bwd": (c_i32, [c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_i32, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "rt_embed_bwd_prepare": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "rt_embed_packed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_vp]),
    "rt_embed_packed_bwd": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_u64, c_vp, c_i32, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "rt_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rt_layernorm_fwd_masked": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_layernorm_bwd_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "rt_layernorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "rt_layernorm_bwd_fused": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "rt_layernorm_bwd_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "rt_layernorm_bwd_rows_scaled": (c_i32, [c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "rt_layernorm_bwd_combine": (c_i32, [c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rt_layernorm_fwd_cols": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_layernorm_bwd_cols": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "rt_act_dropout_fwd": (c_i32, [c_vp, c_i32, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp, c_vp]),
    "rt_act_dropout_bwd": (c_i32, [c_vp, c_vp, c_i32, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp]),
    "rt_swiglu_fwd": (c_i32, [c_vp, c_vp, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp]),
    "rt_swiglu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp, c_vp]),
    "rt_gate_fwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp]),
    "rt_gate_bwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_u64, c_u64, c_i64, c_vp, c_vp, c_vp]),
    "rt_axpy": (c_i32, [c_vp, c_f32, c_vp, c_i64, c_vp, c_vp]),
    "rt_mul_mask": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "rt_mul_mask_ld": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "rt_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "rt_adam_step_segments": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "rt_mha_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_last_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_mha_fwd_scaled": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_bwd_scaled": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_last_fwd_scaled": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_mha_varlen_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_mha_varlen_train_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_varlen_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rt_mha_varlen_bidir_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_varlen_bidir_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_varlen_prefix_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_vp]),
    "rt_mha_varlen_prefix_bwd_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "rt_mha_varlen_prefix_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_u64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "rt_mha_varlen_last_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_mha_varlen_last_x_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "rt_mha_last_x_expand": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "rt_sasrec_block_saved_floats": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "rt_sasrec_block_bwd_scratch_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "rt_sasrec_block_grad_offsets": (None, [c_i32, c_i32, c_vp]),
    "rt_sasrec_block_packed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_sasrec_block_packed_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp]),
    "rt_sasrec_block_infer_scratch_floats": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "rt_sasrec_block_packed_infer": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_embed_block1_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rt_embed_block1_preln_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rt_sasrec_step_arena_bytes": (c_sz, [c_vp]),
    "rt_sasrec_step_run": (c_i32, [c_vp, c_i32, c_vp]),
    "rt_sasrec_step_grad_ptrs": (c_i32, [c_vp, c_vp]),
    "rt_preln_block_saved_floats": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "rt_preln_block_bwd_scratch_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "rt_preln_block_packed_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rt_preln_block_packed_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp]),
    "rt_side_join": (c_i32, [c_vp]),
    "rt_side_reach": (c_i32, [c_vp]),
    "rt_side_stream": (c_i32, [c_vp]),
    "rt_side_fork": (c_i32, [c_vp, c_vp]),
    "rt_side_mark": (c_i32, []),
    "rt_side_wait_mark": (c_i32, [c_vp]),
    "rt_timing_enable": (c_i32, [c_i32]),
    "rt_timing_collect": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp]),
    "rt_hstu_attn_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_hstu_attn_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rt_hstu_attn_varlen_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_hstu_attn_varlen_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rt_hstu_attn_last_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_hstu_attn_varlen_last_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_sampled_loss_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f64, c_vp, c_vp, c_vp]),
    "rt_sampled_loss_bwd_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "rt_sampled_loss_fwd_train": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f64, c_vp, c_vp, c_vp, c_i64, c_vp, c_sz, c_i32, c_vp]),
    "rt_sampled_loss_prepare": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "rt_sampled_loss_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "rt_loss_reduce": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "rt_softmax_ce_rows": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_i32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "rt_l2norm_fwd": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "rt_l2norm_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_gather_rows": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_scatter_rows": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "rt_dp_unique_id": (c_i32, [c_vp]),
    "rt_dp_init": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "rt_dp_allreduce": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rt_dp_broadcast": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "rt_dp_reduce_scatter": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rt_dp_allgather": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rt_dp_finalize": (c_i32, [c_vp]),
    "rt_dp_last_error": (ctypes.c_char_p, []),
}

_lib: tp.Optional[ctypes.CDLL] = None


class GemmProblem(ctypes.Structure):
    """`rt_gemm_problem` of include/rectools_hip.h (one product of an rt_gemm_grouped launch)."""

    _fields_ = [("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64), ("C", c_vp), ("ldc", c_i64), ("bias", c_vp), ("R", c_vp),
                ("ldr", c_i64), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("relu", c_i32)]


class SasrecBlock(ctypes.Structure):
    """`rt_sasrec_block` of include/rectools_hip.h (geometry, dropout streams and parameter pointers of one packed SASRec block)."""

    _fields_ = [(n, c_i32) for n in ("rows", "rows_real", "B", "H", "d", "dff", "window", "pad_keys")] + \
               [(n, c_f32) for n in ("p_drop", "eps1", "eps2")] + \
               [(n, c_u64) for n in ("seed_attn", "seed_h", "sid_h", "seed_o", "sid_o")] + \
               [(n, c_vp) for n in ("cu", "ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2",
                                    "in_wp", "out_wp", "w1_wp", "w2_wp")] + [("wp_stride", c_i64)]


class SasrecStep(ctypes.Structure):
    """`rt_sasrec_step` of include/rectools_hip.h (one packed SASRec training step: batch, parameters, arena, Adam segments)."""

    _fields_ = [(n, c_i32) for n in ("n_blocks", "rows", "rows_real", "B", "B_attn", "V", "d", "dff", "H", "window", "pad_keys", "n_neg", "loss",
                                     "cosine", "wgrad_splits", "pos_rows")] + \
               [(n, c_f32) for n in ("p_emb", "p_blk", "emb_scale", "eps_last", "logits_t")] + [("gbce_beta", c_f64)] + \
               [(n, c_u64) for n in ("seed_emb", "sid_emb")] + \
               [(n, c_vp) for n in ("ids", "dist", "y", "neg", "cu", "cu_attn", "yw", "table", "pos", "lnf_w", "lnf_b", "blocks", "planes_src")] + \
               [("planes_n", c_i64), ("planes", c_vp), ("planes_stride", c_i64), ("upstream", c_vp), ("loss_out", c_vp), ("arena", c_vp),
                ("arena_bytes", c_sz), ("flat_p", c_vp), ("adam_m", c_vp), ("adam_v", c_vp), ("n_seg", c_i32), ("seg_offsets", c_vp),
                ("seg_lens", c_vp), ("seg_role", c_vp), ("adam_step", c_i32), ("lr", c_f32), ("beta1", c_f32), ("beta2", c_f32), ("adam_eps", c_f32)]


class PreLNBlock(ctypes.Structure):
    """`rt_preln_block` of include/rectools_hip.h."""

    _fields_ = [(n, c_i32) for n in ("rows", "rows_real", "B", "H", "d", "dff", "window", "causal")] + \
               [(n, c_f32) for n in ("p_drop", "eps1", "eps2")] + \
               [(n, c_u64) for n in ("seed_attn", "seed1", "sid1", "seed_h", "sid_h", "seed2", "sid2", "seed3", "sid3")] + \
               [(n, c_vp) for n in ("cu", "ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2",
                                    "in_wp", "out_wp", "w1_wp", "w2_wp")] + [("wp_stride", c_i64)]


class GemmWpProblem(ctypes.Structure):
    """`rt_gemm_wp_problem` of include/rectools_hip.h."""

    _fields_ = [("A", c_vp), ("lda", c_i64), ("W", c_vp), ("plane_stride", c_i64), ("ldw", c_i64), ("C", c_vp), ("ldc", c_i64),
                ("bias", c_vp), ("R", c_vp), ("ldr", c_i64), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("relu", c_i32)]


class HipLibraryError(RuntimeError):
    """Raised when the HIP extension is missing or reports a failure."""


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Fails loudly: there is no CPU / eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found. Build it with `python -m rectools_amd.build` "
            "(or `__graft_entry__.build()`); the MI355X engine has no fallback path."
        )
    # One HIP runtime per process: torch bundles its own libamdhip64 and owns the memory and streams we are
    # handed, so it must be loaded first; our library then binds to the already loaded runtime by SONAME.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HipLibraryError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Map C status codes to the reference's exception types (Errors)."""
    if status == RT_OK:
        return
    if status == RT_ERR_INVALID_ARG:
        raise ValueError(f"{what}: invalid argument")
    if status == RT_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: unsupported configuration")
    if status == RT_ERR_WORKSPACE:
        raise HipLibraryError(f"{what}: workspace missing or too small")
    detail = ""
    if _lib is not None:
        try:
            detail = (_lib.rt_last_error() or b"").decode(errors="replace")
        except Exception:  # pragma: no cover
            detail = ""
    raise HipLibraryError(f"{what}: HIP launch failed (status {status}) {detail}")


def ptr(t) -> tp.Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_RAW_STREAM = None


def current_stream() -> int:
    """Raw handle of torch's current HIP stream on the current device.  `torch.cuda.current_stream().cuda_stream` builds a Stream
    object per call (8 us of the 17 us a ctypes launch costs on the host; ~100 launches per training step); the private raw
    getter is two orders of magnitude cheaper and is what torch's own extensions use."""
    global _RAW_STREAM
    import torch

    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)   # pylint: disable=protected-access
    if _RAW_STREAM:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    import torch

    if not torch.cuda.is_available():
        raise HipLibraryError("no HIP device visible: the MI355X engine cannot run (no CPU fallback)")
