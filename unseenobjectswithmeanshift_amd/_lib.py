"""ctypes binding of libmsm_hip.so (C ABI declared in include/msm_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the reference silently falls back to a PyTorch path on ANY exception,
ops/modules/ms_deform_attn.py:116-121 -- deliberately not reproduced).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsm_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "msm_hip.h")

_lib = None
ABI_VERSION = 22     # must equal MSM_ABI_VERSION of include/msm_hip.h (checked when the library is loaded)

c_f = ctypes.c_void_p      # float* (device)
c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_fl = ctypes.c_float

_SIGNATURES = {
    "msm_abi_version": (c_i, []),
    "msm_last_error_string": (ctypes.c_char_p, []),
    "msm_set_option": (c_i, [c_i, c_i]),
    "msm_get_option": (c_i, [c_i]),
    "msm_gemm_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i,
                           c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l, c_l,
                           c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_layernorm_f32": (c_i, [c_f, c_f, c_i, c_l, c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_p]),
    "msm_groupnorm_stats_f32": (c_i, [c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_groupnorm_apply_f32": (c_i, [c_f, c_p, c_f, c_f, c_f, c_i, c_i, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_p]),
    "msm_groupnorm_apply_split": (c_i, [c_f, c_p, c_f, c_f, c_f, c_i, c_i, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_p]),
    "msm_groupnorm_apply_f16": (c_i, [c_f, c_p, c_f, c_f, c_f, c_i, c_i, c_l, c_p, c_i, c_i, c_i, c_i, c_i, c_fl, c_i, c_p]),
    "msm_groupnorm_apply_nchw_f32": (c_i, [c_f, c_p, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_fl, c_i, c_p]),
    "msm_pos_embed_sine": (c_i, [c_f, c_i, c_i, c_i, c_l, c_l, c_f, c_fl, c_fl, c_p]),
    "msm_transpose_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_p]),
    "msm_l2_normalize_nchw_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_fl, c_p]),
    "msm_msda_locations": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_l, c_i, c_i, c_i, c_p]),
    "msm_mask_logits_fwd": (c_i, [c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_f, c_l, c_p]),
    "msm_pool_mask_taps": (c_i, [c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_l, c_p]),
    "msm_attn_mask_pooled": (c_i, [c_f, c_l, c_f, c_l, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_pack_mask_features_bf16": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p]),
    "msm_pack_mask_features_f16": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p]),
    "msm_mask_logits_bf16_fwd": (c_i, [c_f, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_f, c_l, c_p]),
    "msm_pack_mask_features_split": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p]),
    "msm_mask_logits_split_fwd": (c_i, [c_f, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_f, c_l, c_p]),
    "msm_hypersphere_attn_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "msm_hypersphere_attn_fwd": (c_i, [c_f, c_f, c_f, c_p, c_p, c_f, c_i, c_i, c_i, c_i,
                                       c_l, c_l, c_l, c_l, c_l, c_l, c_fl, c_f, c_l, c_p]),
    "msm_hypersphere_attn_lp_fwd": (c_i, [c_f, c_p, c_p, c_i, c_p, c_p, c_f, c_i, c_i, c_i, c_i,
                                          c_l, c_l, c_l, c_l, c_l, c_l, c_fl, c_f, c_l, c_p]),
    "msm_hypersphere_attn_bwd_workspace": (c_l, [c_i, c_i, c_i]),
    "msm_hypersphere_attn_bwd": (c_i, [c_f, c_f, c_f, c_p, c_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i,
                                       c_l, c_l, c_l, c_l, c_l, c_l, c_fl, c_f, c_l, c_p]),
    "msm_msdeform_attn_fwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_bwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_fwd_f64": (c_i, [c_f, c_p, c_p, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_bwd_f64": (c_i, [c_f, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_enc_fwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_encoder_block_stream_floats": (c_l, [c_i, c_i]),
    "msm_value_to_head_major_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_enc_hm_fwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msda_pack_proj": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_enc_fused_fwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_encoder_block_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_p]),
    "msm_encoder_block_split_stream_bytes": (c_l, [c_i, c_i]),
    "msm_encoder_block_split_fwd": (c_i, [c_f, c_f, c_p, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_p]),
    "msm_encoder_block_lp_stream_bytes": (c_l, [c_i, c_i]),
    "msm_encoder_block_lp_fwd": (c_i, [c_f, c_f, c_p, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_p]),
    "msm_encoder_block_hm_stream_bytes": (c_l, [c_i, c_i]),
    "msm_encoder_block_hm_small_floats": (c_i, [c_i]),
    "msm_encoder_block_hm_fwd": (c_i, [c_p, c_f, c_p, c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_i, c_fl, c_i, c_p]),
    "msm_msdeform_attn_enc_lp_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_msdeform_attn_enc_lp_fused_fwd": (c_i, [c_p, c_p, c_p, c_f, c_f, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_f32_to_f16": (c_i, [c_f, c_p, c_l, c_p]),
    "msm_bias_act_nhwc": (c_i, [c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_p]),
    "msm_nhwc_to_nchw_f32": (c_i, [c_p, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_ucn_embedding_tail": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_p]),
    "msm_f32_to_f16_rows": (c_i, [c_f, c_p, c_i, c_l, c_l, c_p]),
    "msm_kv_project_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_p]),
    "msm_kv_project_multi_f32": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "msm_kv_project_multi_bf16": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_kv_project_multi_split": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "msm_tokens_proj_nchw_f32": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_i, c_fl, c_i, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_dec_pack_weight": (c_i, [c_f, c_f, c_i, c_i, c_p]),
    "msm_dec_post_cross": (c_i, [c_f] * 12 + [c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_post_self": (c_i, [c_f] * 9 + [c_i, c_f, c_f, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_heads": (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, c_i] + [c_f] * 15 + [c_p, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_pack_weight_bf16": (c_i, [c_f, c_p, c_i, c_i, c_p]),
    "msm_dec_post_cross_bf16": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f] + [c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_post_self_bf16": (c_i, [c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_p] + [c_i, c_f, c_f, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_heads_bf16": (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_f] + [c_f] * 4 +
                           [c_p, c_i, c_i, c_i, c_fl, c_p]),
    "msm_nchw_to_tokens_f16": (c_i, [c_f, c_p, c_i, c_i, c_i, c_p]),
    "msm_mask_conv3x3_folded": (c_i, [c_p, c_f, c_l, c_l, c_p, c_p, c_i, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_attn_pack_kv_weights": (c_i, [c_f, c_p, c_i, c_p]),
    "msm_attn_mask_bits_bytes": (c_l, [c_i, c_i, c_i]),
    "msm_attn_pack_mask_bits": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "msm_hypersphere_attn_fused_kv_fwd": (c_i, [c_f, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_fl, c_f, c_l, c_p]),
    "msm_dec_pack_weight_f16": (c_i, [c_f, c_p, c_i, c_i, c_p]),
    "msm_dec_post_cross_f16": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f] + [c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_post_self_f16": (c_i, [c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_p] + [c_i, c_f, c_f, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_heads_f16": (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_f] + [c_f] * 4 +
                          [c_p, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_heads_mask": (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_f] + [c_f] * 4 +
                           [c_f, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_pack_weight_bf16x2": (c_i, [c_f, c_p, c_i, c_i, c_p]),
    "msm_dec_post_cross_bf16x2": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_f] + [c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_post_self_bf16x2": (c_i, [c_f, c_f, c_p, c_f, c_f, c_f, c_p, c_f, c_p] + [c_i, c_f, c_f, c_i, c_i, c_i, c_fl, c_p]),
    "msm_dec_heads_bf16x2": (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_p, c_f, c_f] + [c_f] * 4 +
                             [c_p, c_i, c_i, c_i, c_fl, c_p]),
    "msm_l2_prefetch": (c_i, [c_p, c_p, c_i, c_p]),
    "msm_dec_set_prefetch": (c_i, [c_p, c_p, c_i]),
    "msm_ms_seed_workspace": (c_l, [c_i]),
    "msm_ms_select_seeds": (c_i, [c_f, c_i, c_i, c_i, c_l, c_f, c_p, c_f, c_l, c_i, c_p]),
    "msm_ms_hill_climb_workspace": (c_l, [c_i, c_i]),
    "msm_ms_hill_climb": (c_i, [c_f, c_i, c_i, c_f, c_i, c_fl, c_i, c_f, c_l, c_p]),
    "msm_ms_hill_climb_split_workspace": (c_l, [c_i, c_i]),
    "msm_ms_hill_climb_split": (c_i, [c_f, c_i, c_i, c_f, c_i, c_fl, c_i, c_f, c_l, c_p]),
    "msm_ms_bf16_rows": (c_l, [c_i]),
    "msm_ms_pack_bf16": (c_i, [c_f, c_i, c_i, c_p, c_p]),
    "msm_ms_select_seeds_bf16": (c_i, [c_p, c_f, c_i, c_i, c_i, c_l, c_f, c_p, c_f, c_l, c_i, c_p]),
    "msm_ms_hill_climb_bf16": (c_i, [c_p, c_i, c_i, c_f, c_i, c_fl, c_i, c_f, c_l, c_p]),
    "msm_ms_assign": (c_i, [c_f, c_i, c_i, c_f, c_i, c_p, c_p, c_p, c_i, c_p]),
    "msm_ms_connected_components": (c_i, [c_f, c_i, c_i, c_fl, c_p, c_p, c_p]),
    "msm_ms_relabel_largest_zero": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p]),
    "msm_topk_class_scores": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    "msm_topk_class_scores_gather": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f, c_l, c_i, c_f, c_p]),
    "msm_conv1x1_in_f32": (c_i, [c_f, c_f, c_f, c_f, c_l, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv1x1_in_multi_f32": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_i, c_i, c_p]),
    "msm_conv1x1_in_multi_lp": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_i, c_i, c_p]),
    "msm_conv1x1_in_multi_wide": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_l, c_p, c_i, c_i, c_p]),
    "msm_conv1x1_in_lp": (c_i, [c_f, c_p, c_f, c_f, c_l, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_f32": (c_i, [c_f, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_bf16": (c_i, [c_f, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_f16": (c_i, [c_f, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_f16h": (c_i, [c_p, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_nchw_f16": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_split": (c_i, [c_f, c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_nchw_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_conv3x3_c64_nchw_bf16": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_encoder_prologue_stream_floats": (c_l, [c_i]),
    "msm_encoder_prologue_fwd": (c_i, [c_f, c_p, c_f, c_p, c_i, c_i, c_fl, c_f, c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "msm_encoder_prologue_hm_weight_bytes": (c_l, []),
    "msm_encoder_prologue_hm_fwd": (c_i, [c_f, c_p, c_f, c_p, c_i, c_i, c_fl, c_p, c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_p]),
    "msm_label_stats": (c_i, [c_f, c_f, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    "msm_label_image": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_crop_resize": (c_i, [c_f, c_f, c_f, c_p, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_paste_labels": (c_i, [c_f, c_p, c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p]),
    "msm_instance_postprocess_workspace": (c_l, [c_i, c_i, c_i, c_i]),
    "msm_instance_postprocess": (c_i, [c_f, c_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
}


def declared_symbols():
    """Function names declared in include/msm_hip.h (used by the symbol-export test)."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_[a-z0-9_]+)\s*\(", src)))


def lib():
    """Load libmsm_hip.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.msm_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI {L.msm_abi_version()}, these bindings need {ABI_VERSION}: rebuild it")
        _lib = L
    return _lib


# kernel-selection overrides of include/msm_hip.h (enum order), for tools/ and tests/ only
OPTIONS = ("MASK_NC", "MASKB_TARGET", "GEMM_TILE", "GEMM_SHALLOW", "ATTN_TARGET", "ATTN_KERNEL", "ATTN_QK_MAX", "ATTN_QKCFG",
           "CONVIN_NT", "POST_GENERIC", "ENC_NO_COOP", "MSDA_GENERIC", "MS_CHUNK", "MS_NO_PERSISTENT", "ATTN_FUSED_KV", "KV_PIPE", "MASK_KERNEL",
           "MS_SPLIT_KERNEL", "CONV3_WIDE", "DEC_TILE32")
OPT_AUTO = -1


def set_option(name, value=OPT_AUTO):
    """msm_set_option(MSM_OPT_<name>, value); value OPT_AUTO restores the library's own choice.  Returns the old value."""
    key = OPTIONS.index(name)
    old = lib().msm_get_option(key)
    check(lib().msm_set_option(key, int(value)), "msm_set_option")
    if old != int(value):
        from ._plan import bump_plan_epoch
        bump_plan_epoch()          # captured graphs hold the kernels the old option selected: they re-capture (graphs.py)
    return old


class option:
    """``with option("MASK_NC", 1): ...`` -- scoped override, restored on exit."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


class CallTimer:
    """Measurement aid for bench.py: while active, every launch entry point of the library (``msm_*`` functions taking a
    stream) is bracketed by HIP events recorded on the stream it launches on (torch's current stream -- the one ops._stream()
    hands to the library).  ``durations()`` -> {entry point: [ms per call]} after a synchronize.  Eager launches only:
    events cannot time nodes inside a HIP-graph replay."""

    def __init__(self):
        self.records = []
        self._saved = {}

    def __enter__(self):
        import torch
        L = lib()
        for name, (_, args) in _SIGNATURES.items():
            if not args or args[-1] is not c_p or name.endswith("_workspace"):
                continue
            fn = getattr(L, name)
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _name=name):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*a)
                e1.record()
                self.records.append((_name, e0, e1))
                return rc

            setattr(L, name, wrapped)
        return self

    def __exit__(self, *exc):
        L = lib()
        for name, fn in self._saved.items():
            setattr(L, name, fn)
        self._saved = {}
        return False

    def durations(self):
        out = {}
        for name, e0, e1 in self.records:
            out.setdefault(name, []).append(e0.elapsed_time(e1))
        return out


def check(rc, what):
    if rc != 0:
        msg = lib().msm_last_error_string()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
