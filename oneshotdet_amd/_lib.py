"""ctypes binding of liboneshotdet_hip.so (C-ABI declared in include/oneshotdet_hip.h).

There is NO fallback: if the library is missing or fails to load, every op raises.  (The oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSD_LIB_PATH: another build of the SAME library (A/B timing of two kernel versions on one box); never a fallback
LIB_PATH = os.environ.get("OSD_LIB_PATH") or os.path.join(_HERE, "lib", "liboneshotdet_hip.so")

OSD_F32, OSD_BF16 = 0, 1
ABI_VERSION = 4      # include/oneshotdet_hip.h: osd_abi_version() of the library this binding was written against
ACT_NONE, ACT_RELU, ACT_EXP_SCALE = 0, 1, 2
RES_NONE, RES_SAME, RES_UP2X, RES_DOWN2X = 0, 1, 2, 3
GN_SPLITS = 64


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "n", "h", "w", "cin", "in_stride_n", "in_stride_h", "in_stride_w", "ho", "wo", "cout", "r", "s",
        "stride_h", "stride_w", "pad_h", "pad_w", "w_rows", "out_stride", "res_mode", "res_h", "res_w", "res_stride",
        "act")] + [("act_scale", C.c_float), ("relu_in", C.c_int32), ("algo", C.c_int32), ("reserved0", C.c_int32),
                   ("ordered_ws", C.c_void_p), ("ordered_ws_bytes", C.c_int64)]


_p, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); must list every symbol declared in include/oneshotdet_hip.h
SIGNATURES = {
    "osd_last_error_string": (C.c_char_p, []),
    "osd_abi_version": (_i, []),
    "osd_conv_algo_count": (_i, []),
    "osd_conv2d_fwd": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_fwd_grouped": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_fwd_multi": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_fwd_multi_gn": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "osd_pack_conv_weight": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_pack_stem_weight": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "osd_pack_image": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_nhwc_to_nchw_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_nchw_f32_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "osd_maxpool3x3s2_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_groupnorm_stats": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "osd_groupnorm_finalize": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "osd_groupnorm_relu_apply": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_roialign_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _p]),
    "osd_shot_mean": (_i, [_p, _p, _i, _i, _i, _p]),
    "osd_query_pool_levels": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "osd_query_pool_levels_bwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p]),
    "osd_correlate_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_correlate_levels": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "osd_correlate_bwd_query_levels": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "osd_fcos_score_decode": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "osd_fcos_score_decode_sizes": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _i, _p]),
    "osd_level_topk": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "osd_rank_sort_gather": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "osd_nms_sorted": (_i, [_p, _p, _p, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p, _p]),
    "osd_nms_workspace_bytes": (_i64, [_i, _i]),
    "osd_nms": (_i, [_p, _p, _i, _f, _i, _p, _p, _p, _p]),
    "osd_nms_single_workspace_bytes": (_i64, [_i]),
    "osd_sigmoid_focal_fwd": (_i, [_p, _p, _p, _i, _i, _f, _f, _p]),
    "osd_sigmoid_focal_bwd": (_i, [_p, _p, _p, _p, _i, _i, _f, _f, _p]),
    "osd_pack_conv_weight_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_pack_conv_weight_ex": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_conv2d_wgrad": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    "osd_groupnorm_relu_fwd_levels": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "osd_groupnorm_relu_fwd_levels_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, C.c_uint32, _p]),
    "osd_groupnorm_relu_bwd_levels": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_groupnorm_relu_bwd_levels_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, C.c_uint32, _p]),
    "osd_groupnorm_relu_bwd_levels_convbias": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_groupnorm_onepass_workspace_bytes": (_i64, [_i, _p, _i, _i, _i, _i]),
    "osd_groupnorm_onepass_sync_bytes": (_i64, [_i, _i]),
    "osd_groupnorm_relu_fwd_levels_onepass": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "osd_groupnorm_relu_bwd_levels_onepass": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_groupnorm_onepass_selftest_timeout": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p]),
    "osd_sgd_momentum_multi": (_i, [_p, _p, _i, _p, _p, _p, _f, _f, _i, _p]),
    "osd_sgd_momentum_pack_multi": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _f, _i, _i, _p]),
    "osd_pack_multi": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _p]),
    "osd_conv2d_wgrad_grouped": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_wgrad_pred": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_wgrad_pred_workspace_bytes": (_i64, [_i, _p, _p, _p, _i]),
    "osd_pred_dy_gather": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p]),
    "osd_pred_dgrad_pack": (_i, [_i, _p, _i, _i, _p, _p]),
    "osd_conv2d_wgrad_pred_gathered": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_wgrad_batched": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_wgrad_multi": (_i, [C.POINTER(ConvDesc), _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_conv2d_wgrad_mixed": (_i, [_i, C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    "osd_unpack_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "osd_bias_grad": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "osd_conv2d_dgrad_naive": (_i, [C.POINTER(ConvDesc), _p, _p, _p, _p, _p, _p]),
    "osd_scatter2x": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_add_mask": (_i, [_p, _p, _p, _p, _i64, _i, _p]),
    "osd_upsample2x_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "osd_correlate_bwd_query": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "osd_roialign_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _i, _p]),
    "osd_shot_mean_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "osd_cast_f32": (_i, [_p, _p, _i64, _i, _p]),
    "osd_grad_wire_cast": (_i, [_p, _p, _i64, _i, _p]),
    "osd_fcos_loss_level": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _p, _p, _p, _p, _i, _p, _i, _p]),
    "osd_fcos_loss_levels": (_i, [_i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _f, _f, _f, _p, _p, _p, _p, _i, _p, _i, _p]),
    "osd_fcos_loss_finalize": (_i, [_p, _p, _i, _p]),
    "osd_fcos_loss_finalize_scales": (_i, [_p, _p, _i, _p, _p, _p, _i, _p]),
    "osd_roi_pool_levels": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "osd_groupnorm_act_rois": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i, _p]),
    "osd_box_decode": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _f, _f, _p, _f, _i, _p]),
    "osd_append_gt_boxes": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "osd_proposals_sort_nms": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p]),
    "osd_proposals_sort_nms_hint": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _f, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "osd_proposals_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "osd_box_match_sample": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "osd_box_loss": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _f, _p, _p, _i, _i, _p]),
    "osd_voc_match": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "osd_voc_curves": (_i, [_p, _p, _p, _i, _p, _p, _p]),
    "osd_voc_ap": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "osd_coco_match": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
    "osd_groupnorm_act_rois_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _i, _i, _i, _p]),
    "osd_rois_sum": (_i, [_p, _p, _i, _i, _i64, _i, _p]),
    "osd_roi_pool_levels_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "osd_image_transform": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "osd_image_transform_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "osd_image_transform_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
}

_lib = None


class OsdError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  torch must be imported first so that libamdhip64.so.7 resolves to the HIP runtime
    PyTorch already loaded (one runtime per process: streams and device pointers are shared with torch)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OsdError("liboneshotdet_hip.so not built (%s missing): run `python -m oneshotdet_amd.build` or "
                       "__graft_entry__.build(); there is no CPU/eager fallback" % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 first)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.osd_abi_version() != ABI_VERSION:
        raise OsdError("%s reports ABI %d, this binding is written for ABI %d: rebuild it (`python -m oneshotdet_amd.build`)"
                       % (LIB_PATH, lib.osd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().osd_last_error_string()
        err = OsdError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
        err.code = rc
        raise err


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
