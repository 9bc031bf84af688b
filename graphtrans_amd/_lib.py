"""ctypes binding of libgraphtrans_hip.so (the C ABI declared in include/graphtrans_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
The product never routes through oracle/ or any CPU/eager re-implementation.
"""
import ctypes as C
import os

import torch  # noqa: F401  must come first: PyTorch-ROCm bundles its own libamdhip64 (soname
#               libamdhip64.so.7); loading it before our library makes both share ONE HIP runtime
#               (same device context, streams and allocations).

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GT_LIB_PATH") or os.path.join(_HERE, "libgraphtrans_hip.so")   # GT_LIB_PATH: A/B builds

GT_F32, GT_BF16 = 0, 1
GT_CONV_GCN, GT_CONV_GIN = 0, 1
GT_EDGE_NONE, GT_EDGE_LINEAR, GT_EDGE_TABLES, GT_EDGE_DENSE = 0, 1, 2, 3

_p = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t
_u64 = C.c_uint64
_d = C.c_double

# name -> (restype, argtypes); mirrors include/graphtrans_hip.h exactly
SIGNATURES = {
    "gt_version": (_i, []),
    "gt_last_error": (C.c_char_p, []),
    "gt_option_set": (_i, [C.c_char_p, _i]),
    "gt_option_get": (_i, [C.c_char_p]),
    "gt_profile_enable": (_i, [C.c_uint]),
    "gt_profile_resume": (_i, [C.c_uint]),
    "gt_profile_count": (_i64, []),
    "gt_profile_get": (_i, [_i64, C.c_char_p, _i64, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "gt_attr_rank": (_i, [_p, _i64, _p, _p]),
    "gt_collate_workspace_bytes": (_sz, [_i64]),
    "gt_collate": (_i, [_p, _p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "gt_graph_prep_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gt_graph_prep": (_i, [_p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_aggregate_fwd": (_i, [_i, _i, _i, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p,
                              C.POINTER(C.c_int32), _i64, _p, _p, _p]),
    "gt_aggregate_bwd_workspace_bytes": (_sz, [_i, _i, _i64, _i64, _i64]),
    "gt_aggregate_bwd": (_i, [_i, _i, _i, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p,
                              C.POINTER(C.c_int32), _i64, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_pna_aggregate_fwd": (_i, [_p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    "gt_pna_aggregate_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    "gt_scale_combine_fwd": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p]),
    "gt_scale_combine_bwd": (_i, [_p, _p, _i64, _i, _i, _i, _p, _p]),
    "gt_embed_sum_fwd": (_i, [_i, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "gt_embed_sum_bwd_workspace_bytes": (_sz, [_i, _p, _i64]),
    "gt_embed_sum_bwd": (_i, [_i, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _sz, _p]),
    "gt_embed_sort_plan_bytes": (_sz, [_i, _p, _i64]),
    "gt_embed_sort_workspace_bytes": (_sz, [_i, _p, _i64]),
    "gt_embed_sort": (_i, [_i, _p, _p, _p, _p, _i64, _p, _sz, _p, _sz, _p]),
    "gt_embed_sum_bwd_sorted_workspace_bytes": (_sz, [_i, _i64, _i64]),
    "gt_embed_sum_bwd_sorted": (_i, [_i, _p, _p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "gt_segment_bcast_add": (_i, [_i, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "gt_segment_sum": (_i, [_i, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "gt_seq_layout_packed": (_i, [_p, _i64, _i64, _i, _p, _p, _p, _i64, _p, _p]),
    "gt_segment_sum_workspace_bytes": (_sz, [_i64, _i64]),
    "gt_segment_sum_ws": (_i, [_i, _p, _p, _p, _i64, _i64, _i64, _p, _p, _sz, _p]),
    "gt_seq_gather": (_i, [_i, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i64, _p, _p, _p]),
    "gt_seq_scatter": (_i, [_i, _p, _p, _p, _p, _p, _i64, _i64, _i, _i64, _i64, _p, _p, _p]),
    "gt_batchnorm_workspace_bytes": (_sz, [_i64, _i64]),
    "gt_batchnorm_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _i64, _i64, _p, _p, _p, _f, _u64, _p, _sz, _p]),
    "gt_batchnorm_apply": (_i, [_i, _p, _p, _p, _p, _p, _i, _p, _i64, _i64, _p, _f, _u64, _p]),
    "gt_batchnorm_bwd_apply": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _d, _i, _i64, _i64, _p, _f, _u64, _p]),
    "gt_batchnorm_fwd_bcast": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _f, _u64, _p, _sz, _p]),
    "gt_batchnorm_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i64, _p, _p, _p, _f, _u64, _p, _sz, _p]),
    "gt_layernorm_fwd": (_i, [_i, _p, _p, _p, _p, _f, _f, _u64, _i64, _i64, _p, _p, _p, _p]),
    "gt_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "gt_layernorm_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _u64, _i64, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_linear_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _i, _f, _u64, _p]),
    "gt_colsum_rows_f32": (_i, [_i, _p, _p, _i64, _i64, _p, _p]),
    "gt_seq_token_rows_layernorm": (_i, [_i, _p, _p, _p, _p, _i64, _i64, _i, _i64, _i64, _p, _p, _p, _p, _f, _p, _p, _p, _p]),
    "gt_seq_token_rows": (_i, [_i, _p, _p, _p, _p, _i64, _i64, _i, _i64, _i64, _p, _p, _p]),
    "gt_defer_begin": (_i, [_p, _sz]),
    "gt_defer_limit": (_i, [_sz]),
    "gt_defer_take": (_p, [_sz]),
    "gt_defer_push": (_i, [_p, _i, _i64, _i64, _p, _p, _i64, _i64, _p]),
    "gt_defer_push_strided": (_i, [_p, _i, _i64, _i64, _p, _i64]),
    "gt_defer_room": (_i, [_i]),
    "gt_defer_flush": (_i, [_p]),
    "gt_defer_end": (_i, []),
    "gt_linear_rows_ok": (_i, [_i, _i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_set_rows": (_i, [_p]),
    "gt_linear_rows_layernorm_ok": (_i, [_i64]),
    "gt_linear_set_rows_layernorm": (_i, [_p, _p, _p, _f, _p, _p, _p]),
    "gt_linear_fwd_ld": (_i, [_i, _i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i, _f, _u64, _p]),
    "gt_linear_bwd_ld": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f, _p, _sz, _p]),
    "gt_adamw_chunk_elems": (_i, []),
    "gt_adamw_step": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _p, _f, _f, _f, _f, _f, _i64, _p, _p]),
    "gt_grad_sqnorm": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _p, _p, _p]),
    "gt_grad_clip_coef": (_i, [_p, _i64, _f, _p, _p]),
    "gt_linear_fwd_gelu": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _u64, _p]),
    "gt_linear_bwd_mul": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "gt_stream_create": (_p, [_i]),
    "gt_stream_destroy": (None, [_p]),
    "gt_stream_priority_range": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "gt_linear_bwd_dw_forked": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _p, _sz, _p]),
    "gt_linear_bwd_mul_dw_forked": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "gt_linear_bwd_wt": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _p, _sz, _p]),
    "gt_transpose": (_i, [_p, _p, _i64, _i64, _p]),
    "gt_bn_sync_set": (_i, [_p, _p, _i]),
    "gt_vn_update_bwd_dt0": (_p, [_p, _p]),
    "gt_linear_bwd_bcast_ok": (_i, [_i, _i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_bwd_bcast": (_i, [_p, _p]),
    "gt_linear_cat2_ok": (_i, [_i, _p, _i64, _i64, _i64, _i64]),
    "gt_linear_fwd_cat2": (_i, [_i, _i, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _p]),
    "gt_linear_bwd_cat2": (_i, [_i, _i, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "gt_w3_image_bytes": (_sz, [_i64, _i64]),
    "gt_w3_images": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "gt_w3_bind": (_i, [_i, _p, _p, _p, _p, _p]),
    "gt_w3_unbind": (_i, []),
    "gt_w1_image_bytes": (_sz, [_i64, _i64]),
    "gt_w1_images": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "gt_w1_bind": (_i, [_i, _p, _p, _p, _p, _p]),
    "gt_w1_unbind": (_i, []),
    "gt_linear_layernorm_fwd_ok": (_i, [_i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_bwd_dx_layernorm_ok": (_i, [_i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_bwd_dx_layernorm_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gt_linear_bwd_dx_layernorm": (_i, [_i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _f, _u64, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_layernorm_bwd_finish": (_i, [_p, _i, _i64, _p, _p, _p]),
    "gt_linear_layernorm_fwd": (_i, [_i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _f, _f, C.c_uint64, _p, _p, _p, _p]),
    "gt_linear_bwd_gate_out_ok": (_i, [_i, _i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_bwd_gate_out": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _p, _sz, _p]),
    "gt_dropout": (_i, [_i, _p, _p, _i64, _f, _u64, _p]),
    "gt_linear_bwd_bnstats_ok": (_i, [_i, _i, _i, _i64]),
    "gt_linear_bwd_bnstats_rows": (_i64, [_i64]),
    "gt_linear_bwd_bnstats_rows_for": (_i64, [_i, _i, _i, _p, _i64, _i64, _i64]),
    "gt_linear_bwd_bnstats": (_i, [_p, _i64, _p, _p, _p, _p, _i, _p]),
    "gt_batchnorm_bwd_parts": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "gt_overlap_dw_begin": (_i, [_p, _p]),
    "gt_overlap_dw_sync": (_i, []),
    "gt_overlap_dw_release": (_i, [_p, _sz]),
    "gt_overlap_dw_urgent": (_i, [_i]),
    "gt_overlap_dw_fork": (_p, [_p, C.c_uint]),
    "gt_overlap_dw_booked": (None, [_p, _sz]),
    "gt_overlap_dw_end": (_i, []),
    "gt_linear_fwd_ld2": (_i, [_i, _i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _f, _u64, _p]),
    "gt_linear_fwd_grouped": (_i, [_i, _i, _i, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _i64, _i64, _i, _f, _u64, _p]),
    "gt_linear_bwd_grouped_workspace_bytes": (_sz, [_i, _i64, _i64, _i64, _i]),
    "gt_linear_bwd_grouped": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _i64, _i64, _f,
                                   _p, _sz, _p]),
    "gt_linear_bwd_ld2": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f, _p, _sz, _p]),
    "gt_xent_fwd": (_i, [_p, _i64, _i64, _i64, _i64, _p, _i64, _p, _p, _p, _p, _p]),
    "gt_xent_bwd": (_i, [_p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "gt_bce_masked_fwd": (_i, [_p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p]),
    "gt_bce_masked_bwd": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "gt_linear_bwd_workspace_bytes": (_sz, [_i, _i64, _i64, _i64]),
    "gt_linear_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _p, _sz, _p]),
    # composite layers (descriptor structs are passed by pointer; see graphtrans_amd/layers.py)
    "gt_encoder_layer_saved_bytes": (_sz, [_p]),
    "gt_encoder_layer_workspace_bytes": (_sz, [_p]),
    "gt_encoder_layer_grad_elems": (_i64, [_p]),
    "gt_encoder_layer_fwd": (_i, [_p, _p, _p, _p, _p]),
    "gt_encoder_layer_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_gcn_layer_saved_bytes": (_sz, [_p]),
    "gt_gcn_layer_workspace_bytes": (_sz, [_p]),
    "gt_gcn_layer_grad_elems": (_i64, [_p]),
    "gt_gcn_layer_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_gcn_layer_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_vn_update_saved_bytes": (_sz, [_p]),
    "gt_vn_update_workspace_bytes": (_sz, [_p]),
    "gt_vn_update_grad_elems": (_i64, [_p]),
    "gt_vn_update_fwd": (_i, [_p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_vn_update_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_event_create": (_p, []),
    "gt_event_destroy": (None, [_p]),
    "gt_event_record": (_i, [_p, _p]),
    "gt_stream_wait_event": (_i, [_p, _p]),
    "gt_gin_layer_saved_bytes": (_sz, [_p]),
    "gt_gin_layer_workspace_bytes": (_sz, [_p]),
    "gt_gin_layer_grad_elems": (_i64, [_p]),
    "gt_gin_layer_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_gin_layer_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_add3": (_i, [_p, _p, _p, _i64, _p, _p]),
    "gt_copy2d": (_i, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "gt_repitch": (_i, [_p, _i64, _p, _i64, _i64, _i, _p]),
    "gt_rows_gather": (_i, [_i, _p, _p, _i64, _i64, _p, _p]),
    "gt_rows_scatter": (_i, [_i, _p, _p, _i64, _i64, _i64, _p, _p]),
    "gt_seq_layout_packed_host": (_i, [_p, _i64, _i64, _i, _p, _sz, _p]),
    "gt_pna_layer_saved_bytes": (_sz, [_p]),
    "gt_pna_layer_workspace_bytes": (_sz, [_p]),
    "gt_pna_layer_grad_elems": (_i64, [_p]),
    "gt_pna_layer_fwd": (_i, [_p, _p, _p, _p, _p, _sz, _p]),
    "gt_pna_layer_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_pna_scales": (_i, [_p, _i64, _i, _p, _f, _f, _p, _p]),
    "gt_gather_f32": (_i, [_p, _p, _p, _i64, _p]),
    "gt_pna_aggregate_fwd_uv": (_i, [_p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    "gt_pna_aggregate_bwd_uv": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _p]),
    "gt_rows_take": (_i, [_i, _p, _p, _i64, _i64, _p, _p]),
    "gt_rows_put": (_i, [_i, _p, _p, _i64, _i64, _i64, _p, _p]),
    "gt_rows_add": (_i, [_i, _p, _p, _i64, _i64, _p, _p]),
    "gt_attn_fwd_last": (_i, [_i, _p, _p, _p, _i64, _i64, _i, _p, _i64, _i64, _i64, _f, _f, _u64, _p]),
    "gt_attn_bwd_last": (_i, [_i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p, _i64, _i64, _i64, _p, _i64, _f, _f, _u64, _p]),
    "gt_encoder_layer_pooled_saved_bytes": (_sz, [_p]),
    "gt_encoder_layer_pooled_workspace_bytes": (_sz, [_p]),
    "gt_encoder_layer_pooled_fwd": (_i, [_p, _p, _p, _p, _p, _p]),
    "gt_encoder_layer_pooled_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "gt_model_ctx_bytes": (_sz, []),
    "gt_model_abi_sizes": (_i, [_p]),
    "gt_model_grad_ranges": (_i, [_p, _p, _p]),
    "gt_model_prepare": (_i, [_p, _p, _p, _p]),
    "gt_model_forward": (_i, [_p, _p, _p, _p, _p]),
    "gt_model_backward": (_i, [_p, _p, _p, _p, _p, _i, _p]),
    "gt_seq_gather_cls32": (_i, [_i, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i64, _p, _p]),
    "gt_colsum_f32": (_i, [_i, _p, _i64, _i64, _p, _p]),
    "gt_attn_fwd": (_i, [_i, _p, _p, _p, _i64, _i64, _i, _p, _i64, _i64, _i64, _p, _i64, _p, _p, _f, _f, _f, _u64, _p]),
    "gt_attn_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p, _i64, _i64, _i64, _p, _i64, _p, _p, _f, _f, _f,
                         _u64, _p]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m graphtrans_amd.build` "
                "(graphtrans_amd has no CPU or eager fallback)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(h, name)  # AttributeError if a declared symbol is not exported
            except AttributeError:
                if not os.environ.get("GT_LIB_PATH"):
                    raise
                continue   # an A/B build of another revision (GT_LIB_PATH, tools/ab.sh) may predate an entry point
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("GT_F32_GEMM", "split") == "exact":   # the parity yardstick everywhere: w3.ENABLED keeps the GEMMs' images
            h.gt_option_set(b"attn_f32_exact", 1)               # unbound, this keeps attention on the exact fp32 MFMA chains
        _lib = h
    return _lib


class KernelTimer:
    """Optional HIP-event timing of individual C-ABI launches (used by bench.py for the roofline
    object).  Events are recorded on torch's current stream = the stream the kernels are enqueued
    on.  `names`: entry points to time; every timed launch appends (name, start, end, meta)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []

    def summary(self):
        """name -> dict(calls, total_ms, avg_us, metas); synchronises the device."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, meta in self.records:
            d = out.setdefault(name, dict(calls=0, total_ms=0.0, metas=[]))
            d["calls"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["metas"].append(meta)
        for d in out.values():
            d["avg_us"] = 1e3 * d["total_ms"] / max(d["calls"], 1)
        return out


TIMER = None


def option_set(name, value):
    """named runtime option (include/graphtrans_hip.h "Named runtime options") -> the previous value"""
    rc = lib().gt_option_set(name.encode(), int(value))
    if rc < 0:
        check(rc, "gt_option_set")
    return rc


def option_get(name):
    rc = lib().gt_option_get(name.encode())
    if rc < 0:
        check(rc, "gt_option_get")
    return rc


def profile_enable(mask):
    """C-side launch profiler (common.hip): 1 aggregate | 2 attention | 4 linear; 0 = stop."""
    check(lib().gt_profile_enable(int(mask)), "gt_profile_enable")


def profile_records():
    """[(name, ms, dims[6])] of every launch recorded since profile_enable(mask); synchronises."""
    torch.cuda.synchronize()
    L = lib()
    out = []
    name = C.create_string_buffer(64)
    ms = C.c_float()
    dims = (C.c_int64 * 6)()
    for i in range(L.gt_profile_count()):
        check(L.gt_profile_get(i, name, 64, C.byref(ms), dims), "gt_profile_get")
        out.append((name.value.decode(), float(ms.value), list(dims)))
    return out


def launch(name, *args, meta=None):
    """Call entry point `name`, raise on a non-zero status; timed when TIMER selects it."""
    fn = getattr(lib(), name)
    t = TIMER
    if t is not None and name in t.names:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        t.records.append((name, s, e, meta))
    else:
        rc = fn(*args)
    check(rc, name)


# gt_bn_sync_fn (include/graphtrans_hip.h): int fn(void* user, int kind, float* buf, int64_t n, gt_stream_t stream)
BN_SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p)


def check(rc, what):
    if rc != 0:
        msg = lib().gt_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
