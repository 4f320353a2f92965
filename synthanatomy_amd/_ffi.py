"""ctypes binding of libsynthanatomy_hip.so (include/synthanatomy_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a launch fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SA_HIP_LIB") or os.path.join(HERE, "libsynthanatomy_hip.so")   # (SA_HIP_LIB: dev A/B builds)

SA_F32, SA_BF16, SA_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU = 0, 1, 2, 3
MASK_NONE, MASK_POS, MASK_LRELU, MASK_GELU = 0, 1, 2, 3
MAX_TAPS = 64


class ConvGeom(Structure):
    _fields_ = [
        ("N", c_int32), ("Dm", c_int32), ("Hm", c_int32), ("Wm", c_int32),
        ("Di", c_int32), ("Hi", c_int32), ("Wi", c_int32), ("Cin", c_int32),
        ("Do", c_int32), ("Ho", c_int32), ("Wo", c_int32), ("Cout", c_int32),
        ("cin_valid", c_int32), ("cout_valid", c_int32),
        ("KT", c_int32 * 3),
        ("in_mult", c_int32 * 3), ("tap_step", c_int32 * 3), ("in_off", c_int32 * 3),
        ("out_mult", c_int32 * 3), ("out_off", c_int32 * 3),
        ("Kpad", c_int32), ("CoutPad", c_int32),
    ]


class PackDesc(Structure):
    """sa_pack_desc (include/synthanatomy_hip.h)."""
    _fields_ = [("w", c_void_p), ("wpk", c_void_p), ("tap_lut", c_int32 * 64), ("s_row", c_int64), ("s_red", c_int64), ("dtype", c_int32), ("rows", c_int32),
                ("red", c_int32), ("ntaps", c_int32), ("rows_pad", c_int32), ("red_stride", c_int32), ("Kpad", c_int32), ("reserved", c_int32)]


class Epilogue(Structure):
    _fields_ = [
        ("bias", c_void_p), ("addend", c_void_p), ("mask", c_void_p), ("alpha", c_void_p),
        ("act", c_int32), ("mask_mode", c_int32), ("add_before_act", c_int32),
        ("out_dtype", c_int32), ("add_dtype", c_int32), ("mask_dtype", c_int32),
        ("slope", c_float),
        ("out_pre", c_void_p), ("out_lp", c_void_p),
    ]


class LocalAttnArgs(Structure):
    """sa_local_attn_args (include/synthanatomy_hip.h): local-window heads co-launched with the FAVOR+ heads."""
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p),
                ("q_stride", c_int32), ("q_off", c_int32), ("k_stride", c_int32), ("k_off", c_int32), ("v_stride", c_int32), ("v_off", c_int32),
                ("o_stride", c_int32), ("o_off", c_int32),
                ("o", c_void_p), ("lse", c_void_p), ("o_lp", c_void_p),
                ("out", c_void_p), ("dout", c_void_p), ("lse_in", c_void_p),
                ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p), ("Dbuf", c_void_p), ("dv_lp", c_void_p),
                ("L", c_int32), ("W", c_int32)]


_SIGS = {
    "sa_abi_version": (c_int, []),
    "sa_last_error": (c_char_p, []),
    "sa_last_conv_kernel": (c_char_p, []),
    "sa_kernel_log_begin": (None, []),
    "sa_kernel_log_read": (c_int, [ctypes.c_char_p, c_int, c_int]),
    "sa_bench_mfma_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sa_bench_mfma_bf16_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_get_debug_flags": (ctypes.c_uint32, []),
    "sa_set_debug_flags": (ctypes.c_uint32, [ctypes.c_uint32]),
    "sa_conv1_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_conv1_fwd_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_conv1_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_convt1_fused_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_convt1_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_favor_features_project_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                              c_int, c_int, c_void_p]),
    "sa_favor_project_features": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "sa_favor_project": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "sa_favor_project_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "sa_pack_weights_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sa_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int32), c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "sa_conv_fprop": (c_int, [POINTER(ConvGeom), c_int, c_void_p, c_void_p, c_void_p, POINTER(Epilogue), c_void_p]),
    "sa_conv_fprop_classes": (c_int, [POINTER(ConvGeom), c_int, c_int, c_void_p, POINTER(c_void_p), c_void_p, POINTER(Epilogue), c_void_p]),
    "sa_resblock_fprop": (c_int, [POINTER(ConvGeom), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Epilogue), c_void_p]),
    "sa_conv_wgrad": (c_int, [POINTER(ConvGeom), c_int, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int32), c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "sa_conv_wgrad_workspace_bytes": (c_int64, [POINTER(ConvGeom), c_int]),
    "sa_conv1x1_backward": (c_int, [POINTER(ConvGeom), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p]),
    "sa_colsum": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sa_vq_assign": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sa_vq_ema_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "sa_vq_perplexity": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "sa_vq_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "sa_vq_embed": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sa_cast_pad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "sa_mse": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p]),
    "sa_mse_det": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "sa_bn_sums_ws_floats": (c_int64, [c_int]),
    "sa_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]),
    "sa_embed_sum": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_int, c_int, c_int64, c_void_p, c_void_p]),
    "sa_sample_step": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sa_embed_step": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "sa_favor_step": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sa_attn_step": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sa_gemv_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_void_p, c_int, c_int, c_void_p,
                             c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sa_local_attn_step": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "sa_embed_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "sa_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "sa_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sa_gelu": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "sa_rezero_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "sa_rezero_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "sa_axpy": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "sa_favor_features_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "sa_favor_features_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int64, c_int, c_int, c_void_p]),
    "sa_favor_projection": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_favor_scan_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "sa_favor_scan_a": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_void_p, c_void_p]),
    "sa_favor_scan_b": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sa_favor_scan_a_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_int, c_void_p]),
    "sa_favor_scan_b_cum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sa_favor_scan_a_state": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p, c_int, c_void_p]),
    "sa_cumsum_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sa_favor_den": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "sa_favor_dden": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "sa_rotary": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "sa_comm_unique_id": (c_int, [c_void_p]),
    "sa_comm_init": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "sa_comm_destroy": (c_int, [c_void_p]),
    "sa_comm_rank": (c_int, [c_void_p]),
    "sa_comm_world": (c_int, [c_void_p]),
    "sa_comm_all_reduce_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sa_comm_reduce_scatter_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sa_comm_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sa_comm_last_error": (c_char_p, []),
    "sa_subpixel_pool_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_subpixel_pool_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_rotary_pairs": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int64, c_int64, c_void_p]),
    "sa_rotary_groups": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int64,
                                 c_int64, c_void_p, c_void_p]),
    "sa_local_attn_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p, c_void_p]),
    "sa_local_attn_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sa_bn_forward": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_float, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_void_p]),
    "sa_bn_backward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sa_lrelu_mask": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_float, c_void_p]),
    "sa_convt1_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_convt1_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_convt1_im2col": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_convt1_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_favor_fused_state_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "sa_favor_fused_proj_tiles": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "sa_favor_fused_prepass": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "sa_favor_fused_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                   c_int, c_int, c_int, c_int, c_void_p, c_void_p, POINTER(LocalAttnArgs), c_void_p]),
    "sa_favor_fused_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   POINTER(LocalAttnArgs), c_void_p]),
    "sa_sum_det": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "sa_dot_det": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "sa_cross_entropy_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "sa_layernorm_dwprod": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sa_colsum_det_workspace_bytes": (c_int64, [c_int]),
    "sa_colsum_det": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "sa_vq_stats_det": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sa_embed_scatter_det": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "sa_cross_entropy": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load the library (once).  Raises HipLibraryError when it is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -m synthanatomy_amd.build` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.sa_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"ABI version mismatch: {LIB_PATH} reports {l.sa_abi_version()}, this package binds version {ABI_VERSION} "
                                  "(include/synthanatomy_hip.h: SA_ABI_VERSION) -- rebuild with `python -m synthanatomy_amd.build`")
        _lib = l
    return _lib


ABI_VERSION = 4   # include/synthanatomy_hip.h: SA_ABI_VERSION
SA_EINVAL, SA_EUNSUPPORTED, SA_ENOGPU = -1, -2, -3   # include/synthanatomy_hip.h


def check(rc: int, what: str = ""):
    if rc != 0:
        err = lib().sa_last_error()
        raise RuntimeError(f"synthanatomy_hip: {what} failed with code {rc} ({err.decode() if err else ''})")


def dtype_id(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return SA_F32
    if dt == torch.bfloat16:
        return SA_BF16
    if dt == torch.float16:
        return SA_F16      # forward operand / activation type of an f16 forward chain only
    raise TypeError(f"unsupported dtype {dt}")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "synthanatomy_hip kernels need device tensors (no CPU fallback)"
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise HipLibraryError("no HIP device visible: the synthanatomy_amd product path only runs on MI355X (no CPU fallback)")


class kernel_log:
    """``with kernel_log() as names: ...`` -- afterwards ``names`` holds the distinct kernel names the launches dispatched
    (sa_kernel_log_begin / sa_kernel_log_read; process-wide: the backward launches come from autograd's thread)."""

    def __enter__(self):
        self.names = []
        lib().sa_kernel_log_begin()
        return self.names

    def __exit__(self, *exc):
        need = lib().sa_kernel_log_read(None, 0, 0)
        buf = ctypes.create_string_buffer(need)
        lib().sa_kernel_log_read(buf, need, 1)
        self.names.extend(n for n in buf.value.decode().split("\n") if n)
        return False
