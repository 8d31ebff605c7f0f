"""ctypes signatures of the model-side entry points of include/py4cast_hip.h (see _lib.py)."""

import ctypes
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

P, I, L, F = c_void_p, c_int, c_int64, c_float


class HalfUNetDesc(ctypes.Structure):
    """struct p4c_halfunet_desc"""

    _fields_ = [
        ("B", c_int32), ("H", c_int32), ("W", c_int32),
        ("cin", c_int32), ("cin_pad", c_int32), ("cout", c_int32), ("dx_channels", c_int32),
        ("dtype", c_int32), ("norm", c_int32), ("groups", c_int32), ("has_bias", c_int32),
        ("eps", c_float), ("momentum", c_float), ("compute", c_int32),
        ("weights_prepared", c_int32), ("skip_out_conv", c_int32),
    ]


DP = ctypes.POINTER(HalfUNetDesc)


class RowMlpDesc(ctypes.Structure):
    """struct p4c_row_mlp_desc"""

    _fields_ = [
        ("rows", c_int64), ("x", c_void_p), ("k", c_int32), ("k_real", c_int32), ("w1", c_void_p), ("ldw1", c_int32),
        ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("o_real", c_int32), ("gamma", c_void_p), ("beta", c_void_p),
        ("eps", c_float), ("gather_a", c_void_p), ("index_a", c_void_p), ("gather_b", c_void_p), ("index_b", c_void_p),
        ("res", c_void_p), ("out", c_void_p), ("out_res", c_void_p), ("prepared", c_void_p), ("dy", c_void_p), ("dy_res", c_void_p), ("dx", c_void_p),
        ("dpre", c_void_p), ("dx_plus_dy_res", c_int32),
    ]


MP = ctypes.POINTER(RowMlpDesc)


class RowMlpGradSinks(ctypes.Structure):
    """struct p4c_row_mlp_grad_sinks"""

    _fields_ = [("dw1", c_void_p), ("ld_dw1", c_int32), ("db1", c_void_p), ("dw2", c_void_p), ("db2", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p)]


SP = ctypes.POINTER(RowMlpGradSinks)

SIGNATURES = {
    "p4c_prep_weights": [P, I, I, I, I, I, I, P, I, P],
    "p4c_conv_fwd": [P, I, I, I, P, I, P, P, I, P, P, I, P, I, I, I, I, P],
    "p4c_conv_wgrad": [P, I, I, I, I, P, P, I, P, I, I, P, P, I, I, I, P],
    "p4c_conv_fwd_compact": [P, I, P, I, P, I, I, I, I, P],
    "p4c_conv_wgrad_compact": [P, I, I, P, I, I, I, P, P, I, I, I, P],
    "p4c_conv_wgrad_nb": [P, P, P, I, P, P, P, P, P, P, P, P, P, I, I, P, P, I, I, I, P],
    "p4c_conv_wgrad_kernel_kind": [I, I, I, I],
    "p4c_out_conv_bwd": [P, P, P, P, P, P, P, P, P, I, P, P, I, L, P],
    "p4c_upsample_sum_bwd_x": [I, P, I, I, I, P, P, P, P, P],
    "p4c_first_conv_tail": [P, I, I, P, P, P, I, I, I, P],
    "p4c_halfunet_workspace_bytes": [DP, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)],
    "p4c_halfunet_prepare_weights": [DP, P, P, P],
    "p4c_halfunet_forward": [DP, P, P, P, P, P, P, I, P],
    "p4c_halfunet_tail": [DP, P, P, P, P, P, P],
    "p4c_halfunet_backward": [DP, P, P, P, P, P, P, P, I, P],
    "p4c_side_stream_defer": [I],
    "p4c_side_stream_enable": [I],
    "p4c_side_stream_join": [P],
    "p4c_edge_gather_add_fwd": [P, P, P, P, P, P, L, I, I, I, P],
    "p4c_edge_gather_add_bwd": [P, P, P, P, P, P, P, L, I, I, I, P],
    "p4c_segment_sum": [P, P, P, P, P, L, L, I, I, I, P],
    "p4c_segment_sum_pair": [P, P, P, P, L, P, P, P, L, L, I, I, P],
    "p4c_row_layernorm_fwd": [P, P, P, P, F, P, L, I, I, P],
    "p4c_row_layernorm_bwd": [P, P, P, F, P, P, P, P, L, I, I, P],
    "p4c_row_layernorm_fwd_masked": [P, P, P, P, F, P, L, I, I, I, I, I, I, P],
    "p4c_row_layernorm_bwd_masked": [P, P, P, F, P, P, P, P, L, I, I, I, I, I, I, P],
    "p4c_row_linear_wgrad": [P, P, P, P, L, I, I, I, P],
    "p4c_row_gemm": [P, L, P, I, I, P, P, L, L, I, I, P, L, I, P, L, P],
    "p4c_row_gemm_wgrad": [P, L, P, L, P, P, L, I, I, I, P],
    "p4c_row_mlp_fwd": [MP, P],
    "p4c_row_mlp_prepare": [MP, P, P],
    "p4c_row_mlp_bwd": [MP, P, P, P],
    "p4c_row_mlp_bwd_accumulate": [MP, SP, P, P],
    "p4c_node_proj_fwd": [P, L, I, P, P, P, P],
    "p4c_node_proj_dgrad": [P, L, I, P, P, P, P, P],
    "p4c_node_proj_wgrad": [P, P, L, I, P, P, P, P],
    "p4c_grad_reduce_defer": [I],
    "p4c_grad_reduce_flush": [P],
    "p4c_window_attn_fwd": [P, P, P, I, I, I, I, I, I, I, F, I, P],
    "p4c_window_attn_bwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P],
    "p4c_row_add_layernorm_fwd": [P, P, L, P, P, F, P, P, L, I, I, P],
    "p4c_row_add_layernorm_bwd": [P, P, P, P, F, P, P, P, P, L, I, I, P],
    "p4c_sum_leading": [P, I, I, L, P, I, P],
    "p4c_upsample_bilinear_fwd": [P, P, P, I, I, I, I, I, P],
    "p4c_upsample_bilinear_bwd": [P, P, I, I, I, I, I, P],
    "p4c_gemm_prep_weight": [P, I, I, I, P, P, P],
    "p4c_gemm_prep_weight_scaled": [P, P, P, P, I, I, I, P, P, P],
    "p4c_gemm_prep_weight_batch": [I, P, P, P, P, P, P, P, P, P, P],
    "p4c_gemm_scale_fold_bwd": [P, P, P, P, P, I, I, P, P, P, I, P],
    "p4c_gemm_nt": [P, L, P, I, I, I, I, I, I, I, P, P, L, I, P, P, L, P, L, P, P, P],
    "p4c_gemm_tn": [P, L, P, L, I, I, I, I, I, I, P, P, I, P, P],
    "p4c_bnorm_finalize": [P, I, ctypes.c_double, I, P, P, F, F, P, P, P, P, P, P, P, P],
}
OTHER = {
    "p4c_conv_wgrad_workspace_bytes": ([I, I], c_size_t),
    "p4c_out_conv_bwd_slots": ([I, L], c_int),
    "p4c_first_conv_tail_slots": ([I, I, I], c_int),
    "p4c_out_conv_bwd_workspace_bytes": ([I, L], c_size_t),
    "p4c_conv_stat_tiles": ([I, I, I, I, I, I], c_int),
    "p4c_conv_stat_tiles_ks": ([I, I, I, I, I, I, I], c_int),
    "p4c_conv_kernel_kind": ([I, I, I, I, I, I, I], c_int),
    "p4c_conv_compact_supported": ([I, I, I, I, I, I], c_int),
    "p4c_halfunet_param_count": ([DP], c_int64),
    "p4c_window_attn_bwd_workspace_bytes": ([I, I, I, I, I], c_size_t),
    "p4c_row_layernorm_bwd_workspace_bytes": ([L, I, I], c_size_t),
    "p4c_row_linear_wgrad_workspace_bytes": ([L, I], c_size_t),
    "p4c_row_gemm_supported": ([I, I], c_int),
    "p4c_row_gemm_wgrad_supported": ([I, I, I], c_int),
    "p4c_row_gemm_wgrad_workspace_bytes": ([L, I, I, I], c_size_t),
    "p4c_row_mlp_bwd_workspace_bytes": ([L, I], c_size_t),
    "p4c_row_mlp_prepared_bytes": ([I], c_size_t),
    "p4c_node_proj_wgrad_workspace_bytes": ([L, I], c_size_t),
    "p4c_grad_reduce_pending": ([], c_int),
    "p4c_side_stream_launch_count": ([], ctypes.c_longlong),
    "p4c_row_add_layernorm_bwd_workspace_bytes": ([L, I], c_size_t),
    "p4c_gemm_nt_workspace_bytes": ([I, I, I], c_size_t),
    "p4c_gemm_nt_stat_blocks": ([I, I, I], c_int),
    "p4c_gemm_tn_workspace_bytes": ([I, I, I], c_size_t),
}
