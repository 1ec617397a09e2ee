"""ctypes binding of libpanoptic_hip.so (C-ABI declared in include/panoptic_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be resolved this
module raises, and every op built on it fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanoptic_hip.so")

PP_OK = 0
_STATUS = {1: "PP_ERR_INVALID", 2: "PP_ERR_RANGE", 3: "PP_ERR_HIP", 4: "PP_ERR_WORKSPACE", 5: "PP_UNSUPPORTED"}
PP_ERR_WORKSPACE = 4
PP_UNSUPPORTED = 5  # an optional fused form does not serve the shape: nothing was launched

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
f64 = C.c_double
sz = C.c_size_t

# name -> (restype, argtypes); kept in the order of include/panoptic_hip.h
SIGNATURES = {
    "pp_version": (C.c_char_p, []),
    "pp_last_error": (C.c_char_p, []),
    "pp_triad": (C.c_int, [vp, vp, vp, f32, i64, vp]),
    "pp_hash_capacity": (i64, [i64]),
    "pp_hash_build": (C.c_int, [vp, i64, vp, vp, i64, vp, vp]),
    "pp_stride_coords_workspace": (sz, [i64]),
    "pp_stride_coords": (C.c_int, [vp, i64, i32, vp, vp, i64, vp, vp, vp, vp, sz, vp, vp]),
    "pp_kernel_map": (C.c_int, [vp, i64, vp, vp, i64, i32, i32, i32, vp, vp, vp]),
    "pp_exclusive_scan_workspace": (sz, [i64]),
    "pp_exclusive_scan": (C.c_int, [vp, vp, i64, vp, vp, sz, vp]),
    "pp_select_workspace": (sz, [i64]),
    "pp_select_indices": (C.c_int, [vp, i64, vp, vp, vp, sz, vp]),
    "pp_run_lengths": (C.c_int, [vp, i64, vp, vp, vp, vp, vp, sz, vp]),
    "pp_sort_pairs_workspace_bytes": (sz, [i64]),
    "pp_sort_pairs": (C.c_int, [vp, vp, i32, vp, vp, i64, i32, vp, sz, vp]),
    "pp_kernel_map_transpose": (C.c_int, [vp, i64, i32, i64, vp, vp, vp]),
    "pp_block_index_workspace": (sz, [i64]),
    "pp_block_index_capacity": (i64, [i64]),
    "pp_block_index_count": (C.c_int, [vp, i64, i32, i32, vp, vp, vp, sz, vp]),
    "pp_block_index_fill": (C.c_int, [vp, i64, i32, i32, vp, i64, vp, vp, i64, vp, vp, vp, vp]),
    "pp_block_index_coarsen_workspace": (sz, [i64]),
    "pp_block_index_coarsen": (C.c_int, [vp, vp, i64, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, sz, vp]),
    "pp_block_index_coarsen_chain": (C.c_int, [vp, vp, i64, i64, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, sz, vp]),
    "pp_kernel_map_bi": (C.c_int, [vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "pp_proposals_unique_workspace": (sz, [i64]),
    "pp_proposals_unique": (C.c_int, [vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
    "pp_proposals_emit": (C.c_int, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "pp_proposal_pairs_capacity": (i64, [i32]),
    "pp_proposal_pairs_workspace": (sz, [i64, i64, i32]),
    "pp_proposal_pairs": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "pp_nms_paint_workspace": (sz, [i32, i32, i64]),
    "pp_nms_paint": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, i64, vp, i32, vp, f32, i32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "pp_histogram2d": (C.c_int, [vp, vp, i64, i32, i32, vp, vp, vp]),
    "pp_pair_counts_workspace": (sz, [i64]),
    "pp_pair_counts": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, sz, vp]),
    "pp_block_merge_workspace": (sz, [i64]),
    "pp_block_merge": (C.c_int, [vp, vp, i64, vp, i64, vp, vp, vp, sz, vp]),
    "pp_compose_perm": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "pp_map_window": (i32, []),
    "pp_map_mask": (C.c_int, [vp, i32, i64, vp, vp]),
    "pp_map_order": (C.c_int, [vp, i64, vp, vp]),
    "pp_map_order_window": (C.c_int, [vp, i64, i32, vp, vp]),
    "pp_map_compact_workspace": (sz, [i64]),
    "pp_map_compact_count": (C.c_int, [vp, i32, i64, vp, vp, vp, sz, vp]),
    "pp_map_compact_write": (C.c_int, [vp, i32, i64, vp, vp, vp, vp]),
    "pp_spconv_fwd_cmap": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, vp, vp, vp, i64, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp]),
    "pp_map_set_window": (C.c_int, [i32]),
    "pp_map_permute": (C.c_int, [vp, i32, i64, vp, vp, i64, i32, vp, vp]),
    "pp_level_permute": (C.c_int, [vp, i64, vp, vp, vp, vp]),
    "pp_morton_order_workspace": (sz, [i64]),
    "pp_morton_order": (C.c_int, [vp, i64, i32, i32, vp, vp, vp, sz, vp, vp]),
    "pp_packed_weight_floats": (sz, [i32, i32, i32]),
    "pp_kernel_map_transpose8": (C.c_int, [vp, i64, i64, vp, vp, vp, vp]),
    "pp_spconv_fwd_t8": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, i64, i32, vp, vp, i32, vp, vp, vp, i32, vp]),
    "pp_pack_weight": (C.c_int, [vp, i32, i32, i32, i32, vp, vp]),
    "pp_pack_weights_batched": (C.c_int, [vp, vp, i32, i64, vp]),
    "pp_spconv_fwd": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, i32, i64, i32, vp, vp, i32, vp, vp, vp, vp]),
    "pp_spconv_set_scratch": (C.c_int, [vp, sz]),
    "pp_spconv_kernel_family": (C.c_int, [i32, i32, i64, i32, i64, i32, i32]),
    "pp_spconv_x3_full_lines": (C.c_int, [i32]),
    "pp_spconv_fwd_shortcut": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, i32, i64, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp]),
    "pp_spconv_fwd_bf16": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, i32, i64, i32, vp, vp, i32, vp, vp, vp, vp]),
    "pp_spconv_fwd_ex": (C.c_int, [vp, i32, vp, i32, i64, vp, vp, i32, i64, i32, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "pp_spconv_bwd_weight": (C.c_int, [vp, i32, i64, vp, i32, vp, i32, i64, vp, vp]),
    "pp_spconv_bwd_weight_bf16": (C.c_int, [vp, i32, i64, vp, i32, vp, i32, i64, vp, vp]),
    "pp_wgrad_pairs_workspace": (sz, [i32, i64]),
    "pp_wgrad_pairs_build": (C.c_int, [vp, i32, i64, vp, vp, vp, vp, sz, vp]),
    "pp_spconv_bwd_weight_pairs": (C.c_int, [vp, i32, i64, vp, i32, i64, vp, vp, i32, i64, vp, i32, vp]),
    "pp_spconv_bwd_weight_pairs_det_workspace": (sz, [i32, i32, i32, i64]),
    "pp_spconv_bwd_weight_pairs_det": (C.c_int, [vp, i32, i64, vp, i32, i64, vp, vp, i32, i64, vp, i32, vp, sz, vp]),
    "pp_channel_stats": (C.c_int, [vp, i64, i32, vp, vp, vp]),
    "pp_affine_act": (C.c_int, [vp, i64, i32, vp, vp, i32, f32, vp, vp, vp]),
    "pp_bn_bwd_reduce": (C.c_int, [vp, vp, i64, i32, vp, vp, vp]),
    "pp_bn_train_workspace": (sz, [i64, i32]),
    "pp_bn_train_fwd": (C.c_int, [vp, i64, i32, vp, vp, f64, f64, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp]),
    "pp_bn_train_bwd": (C.c_int, [vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "pp_head_mlp": (C.c_int, [vp, i64, i32, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "pp_linear_rows": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    "pp_linear_wgrad_workspace": (C.c_size_t, [i64, i32, i32]),
    "pp_linear_wgrad": (C.c_int, [vp, vp, i64, i32, i32, vp, vp, vp, C.c_size_t, vp]),
    "pp_heads": (C.c_int, [vp, i64, i32, vp, i64, vp, i32, vp, vp]),
    "pp_region_grow_workspace": (sz, [i64, i32]),
    "pp_region_grow_workspace_for": (sz, [i64, i64, i32]),
    "pp_region_grow": (C.c_int, [vp, vp, vp, i64, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "pp_meanshift_workspace": (sz, [i64, i32, i32]),
    "pp_meanshift": (C.c_int, [vp, i64, i32, vp, i32, f32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "pp_hdbscan_workspace": (sz, [i64, i32]),
    "pp_hdbscan": (C.c_int, [vp, i64, i32, vp, i32, i32, i32, i32, i32, C.c_double, vp, vp, vp, sz, vp]),
    "pp_voxelize_workspace": (sz, [i64]),
    "pp_voxelize": (C.c_int, [vp, vp, i64, f32, vp, vp, vp, vp, vp, sz, vp]),
    "pp_cylinder_pairs_workspace": (sz, [i64]),
    "pp_cylinder_pairs": (C.c_int, [vp, i64, vp, i32, f32, vp, vp, i64, vp, vp, sz, vp]),
    "pp_group_by_key_workspace": (sz, [i64]),
    "pp_group_by_key": (C.c_int, [vp, vp, i64, i32, vp, vp, vp, vp, vp, sz, vp]),
    "pp_segment_reduce_workspace": (sz, [i64]),
    "pp_segment_reduce": (C.c_int, [vp, vp, i64, i32, i64, i32, vp, vp, vp, sz, vp]),
    "pp_segment_reduce_unchecked": (C.c_int, [vp, vp, i64, i32, i64, i32, vp, vp, vp, sz, vp]),
    "pp_segment_sum_ordered": (C.c_int, [vp, vp, vp, i64, i32, i32, vp, vp]),
    "pp_instance_iou": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp]),
    "pp_proposal_intersections_workspace": (sz, [i64, i64]),
    "pp_proposal_intersections": (C.c_int, [vp, vp, i32, i64, vp, vp, sz, vp]),
    "pp_gather_rows": (C.c_int, [vp, i64, i32, vp, i64, vp, vp, vp]),
    "pp_nearest_workspace": (sz, [i64]),
    "pp_nearest": (C.c_int, [vp, i64, vp, i64, i32, f32, f32, vp, vp, vp, sz, vp]),
}

_lib = None


class HeadDesc(C.Structure):  # pp_head_t of include/panoptic_hip.h
    _fields_ = [("w1", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("y", C.c_void_p), ("argmax", C.c_void_p), ("cout", C.c_int32), ("log_softmax", C.c_int32)]


class PanopticHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it is not built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("PP_HIP_LIB", LIB_PATH)  # profiling builds (profiles/ablate_conv.sh); never a fallback
    if not os.path.exists(LIB_PATH):
        raise PanopticHipError(
            "libpanoptic_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C panopticsegforlargescalepointcloud_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != PP_OK:
        msg = load().pp_last_error().decode("utf-8", "replace")
        raise PanopticHipError("%s failed: %s (%s)" % (what, _STATUS.get(rc, rc), msg))
