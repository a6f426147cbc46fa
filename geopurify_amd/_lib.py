"""ctypes binding of libgeopurify_hip.so (the C-ABI in include/geopurify_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this
module raises.  PyTorch is used only for device memory and streams; the library sees raw device
pointers, sizes and a hipStream_t.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GP_LIB") or os.path.join(_HERE, "libgeopurify_hip.so")      # (GP_LIB: another build of the library, for same-box A/B runs)

_P = c_void_p  # every device pointer travels as void*

# name -> (restype, [argtypes])  -- mirrors include/geopurify_hip.h one to one
SIGNATURES = {
    "gp_version": (c_int32, []),
    "gp_last_error": (c_char_p, []),
    "gp_debug_set": (c_int32, [c_int32, c_int32]),
    "gp_debug_ptr": (c_int32, [c_int32, _P, c_size_t]),
    "gp_voxelize_workspace_bytes": (c_size_t, [c_int64]),
    "gp_voxelize_f64": (c_int32, [_P, c_int64, POINTER(c_double), _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gp_fnv_hash_f64": (c_int32, [_P, c_int64, _P, _P]),
    "gp_project_points_f64": (c_int32, [_P, c_int64, POINTER(c_double), c_double, c_double, c_double,
                                        c_double, _P, c_int32, c_int32, c_int32, c_double, _P, _P, _P]),
    "gp_render_depth_f64": (c_int32, [_P, c_int64, POINTER(c_double), c_double, c_double, c_double, c_double,
                                      c_int32, c_int32, c_int32, _P, _P]),
    "gp_minmax_i32": (c_int32, [_P, c_int64, _P, _P]),
    "gp_morton_order_workspace_bytes": (c_size_t, [c_int64]),
    "gp_morton_order": (c_int32, [_P, c_int64, _P, _P, _P, c_size_t, _P]),
    "gp_grid_bytes": (c_size_t, [c_int64, POINTER(c_int32)]),
    "gp_grid_build": (c_int32, [_P, c_int64, POINTER(c_int32), POINTER(c_int32), _P, c_size_t, _P]),
    "gp_scatter_mean_csr": (c_int32, [_P, c_int64, c_int32, _P, _P, c_int64, _P, _P, c_int64, c_int32, _P]),
    "gp_gather_rows": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P, c_int64, _P]),
    "gp_kernel_map_build": (c_int32, [_P, _P, c_int64, _P, _P]),
    "gp_sparse_conv": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int32, c_int32, c_int32, _P, _P, _P,
                                 c_int64, c_int32, _P, c_int64, _P]),
    "gp_conv_pairs_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_conv_pairs_build": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gp_conv_chunk_plan_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_conv_chunk_plan": (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "gp_conv_weights_split": (c_int32, [_P, c_int32, c_int32, c_int32, c_float, _P, _P, _P]),
    "gp_split_f16": (c_int32, [_P, c_int64, c_int32, c_int64, _P, _P, c_int64, _P]),
    "gp_sparse_conv_f16x3": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int32, c_int64, c_int64, c_int32, _P, _P,
                                       c_int32, c_int32, _P, _P, _P, _P, c_int64, c_int32, _P, c_int64, _P, _P,
                                       c_int64, c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), _P, _P,
                                       _P, _P, c_int64, _P, c_int32, c_int32, _P]),
    "gp_conv_weights_split_blocked": (c_int32, [_P, c_int32, c_int32, c_int32, c_float, _P, _P, c_int32, _P]),
    "gp_pow2_scale": (c_int32, [_P, c_int64, c_int32, c_int64, _P, _P, c_size_t, _P]),
    "gp_split_f16_scaled": (c_int32, [_P, c_int64, c_int32, c_int64, _P, _P, c_int64, _P, _P, _P, _P]),
    "gp_rcb_order": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P, _P]),
    "gp_rows_renumber_i32": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P]),
    "gp_l2norm_rows": (c_int32, [_P, c_int64, c_int32, c_int64, _P]),
    "gp_embed_head_f16x3": (c_int32, [_P, _P, c_int64, _P, _P, _P, c_int64, c_int32, c_int32, c_float, c_int32, _P, c_int64, _P, _P, c_float, _P, _P]),
    "gp_knn_workspace_bytes": (c_size_t, [c_int64]),
    "gp_knn_lattice": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, c_size_t, _P]),
    "gp_affinity_softmax": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int64, c_float, _P, _P]),
    "gp_affinity_softmax_scatter": (c_int32, [_P, c_int64, c_int32, _P, c_int32, c_int64, c_float, _P, _P, _P, _P, _P]),
    "gp_pool_ell": (c_int32, [_P, c_int64, _P, _P, c_int32, c_int64, c_int32, _P, c_int64, _P]),
    "gp_pool_tiles_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_pool_tiles_count": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "gp_pool_tiles_fill": (c_int32, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P]),
    "gp_pool_tiles_apply": (c_int32, [_P, c_int64, _P, _P, _P, c_int32, c_int64, c_int32, _P, c_int64, _P]),
    "gp_pool_mfma_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_pool_mfma_count": (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "gp_pool_mfma_fill": (c_int32, [_P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, _P, _P, _P]),
    "gp_pool_mfma_apply": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "gp_pool_mfma_apply_persistent": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64,
                                                _P, c_int64, c_int64, _P, _P, _P]),
    "gp_pool_cs_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_pool_cs_count": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "gp_pool_cs_fill": (c_int32, [_P, _P, c_int64, c_int32, c_int32, _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "gp_pool_cs_structure": (c_int32, [_P, c_int64, c_int32, c_int32, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P]),
    "gp_pool_cs_structure_valid": (c_int32, [_P, c_int64, c_int32, c_int32, _P, c_int64, c_int32, _P, _P, _P, _P]),
    "gp_affinity_cs_fragments": (c_int32, [_P, _P, c_int64, c_int32, c_int32, c_float, _P, _P, _P, _P, c_int32, _P, _P, _P]),
    "gp_pool_cs_apply": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "gp_pool_cs_apply_engine": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "gp_pool_cs_apply_half": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "gp_pool_cs_deps": (c_int32, [_P, _P, c_int64, c_int32, _P, _P, _P]),
    "gp_pool_cs_chain_flag_words": (c_size_t, [c_int64, c_int32]),
    "gp_pool_cs_apply_chain": (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, _P, c_int64, _P, _P, _P,
                                         c_uint32, _P]),
    "gp_lift_dense_accum": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P, c_int64, _P, _P]),
    "gp_lift_dense_bilinear_accum": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P,
                                               c_int64, _P, _P]),
    "gp_lift_dense_finish": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "gp_lift_masks_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "gp_lift_masks_view": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32,
                                     _P, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "gp_pv_count": (c_int32, [_P, c_int64, _P, _P]),
    "gp_scan_workspace_bytes": (c_size_t, [c_int64]),
    "gp_exclusive_scan_i64": (c_int32, [_P, c_int64, _P, _P, c_size_t, _P]),
    "gp_pv_fill": (c_int32, [_P, _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "gp_segment_tables": (c_int32, [_P, c_int32, c_int32, _P, c_int32, c_float, _P, _P, _P]),
    "gp_fuse_views_top3": (c_int32, [_P, _P, _P, c_int64, _P, _P, c_int32, c_int32, c_int32, _P, c_int64, _P, _P]),
    "gp_nn1_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gp_nn1_f64": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, c_size_t, _P]),
    "gp_nn1_masked_workspace_bytes": (c_size_t, [c_int64]),
    "gp_nn1_masked_f64": (c_int32, [_P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "gp_visible_lists_workspace_bytes": (c_size_t, [c_int64]),
    "gp_visible_lists": (c_int32, [_P, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gp_views_visible_lists_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_views_visible_lists": (c_int32, [_P, c_int64, _P, _P, c_int32, c_int32, c_int32, c_int32, c_double, c_int64, c_int64,
                                         _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gp_lift_masks_views_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64]),
    "gp_lift_masks_views": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32, _P, c_int64,
                                      _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "gp_classify_argmax": (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_int32, c_float, _P, _P, _P]),
    "gp_gather_rows_classify": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P, c_int64, _P, c_int32, c_float, _P, _P, _P]),
    "gp_rows_argmax": (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_int64, c_int32, _P, _P, _P]),
    "gp_col_stats_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_col_stats": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_size_t, _P]),
    "gp_bn_train_apply": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_float, _P, c_int64, c_int32, _P, c_int64,
                                    _P, _P, c_int64, c_float, _P, _P, _P]),
    "gp_bn_train_backward_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_bn_train_backward": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_float, _P, _P, c_int64, c_int32, _P, c_int64,
                                       _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "gp_col_sums_f64": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_size_t, _P]),
    "gp_bn_bwd_sums_f64": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_float, _P, _P, c_int64, c_int32, _P, _P, c_size_t, _P]),
    "gp_bn_bwd_apply": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_float, _P, _P, _P, c_int64, c_int64, c_int32, _P, c_int64,
                                  _P, c_int64, _P, _P]),
    "gp_infonce_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "gp_infonce_fwd_bwd": (c_int32, [_P, c_int64, c_int64, c_int32, _P, c_int64, _P, c_int64, c_int32, c_float, _P, _P, c_int64,
                                     _P, c_size_t, _P]),
    "gp_conv_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "gp_conv_wgrad_f16x3": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int64, _P, c_int32, c_int32, c_int32, c_int32,
                                      _P, _P, _P, c_size_t, _P]),
    "gp_adamw_step": (c_int32, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "gp_knn_points_f32": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P]),
    "gp_sampler_select": (c_int32, [_P, c_int64, c_int64, c_int64, _P, c_int32, _P, _P, _P]),
    "gp_normalize_split_f16": (c_int32, [_P, c_int64, c_int32, c_int64, c_int64, c_float, _P, _P, c_int64, _P]),
    "gp_iou_hist_i64": (c_int32, [_P, _P, c_int64, c_int32, POINTER(c_int64), c_int32, _P, _P]),
    "gp_fused_decode_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "gp_fused_decode": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P, _P, _P, c_size_t, _P]),
}

_lib = None


class GeoPurifyHipError(RuntimeError):
    pass


def load(path=None):
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime; it must be the one already in the process when this library (linked
    # against libamdhip64 by soname) is loaded, otherwise two runtimes coexist and kernel launches report
    # "no ROCm-capable device" (seen when the library was loaded before the first `import torch`).
    import torch  # noqa: F401
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise GeoPurifyHipError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C geopurify_amd/csrc`).  There is no CPU fallback.")
    lib = ctypes.CDLL(p)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise GeoPurifyHipError(f"{p} does not export {missing}; rebuild the library")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gp_last_error()
        raise GeoPurifyHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def exported_symbols():
    return list(SIGNATURES.keys())
