"""ctypes binding of libcvhip.so — the C ABI declared in include/cvhip.h.

The product path has NO fallback: if the shared object is missing, or an entry point refuses a call,
this module raises. (The pure-torch restatement under /oracle is test infrastructure only and is
never imported from here.)
"""
import ctypes as C
import os

# torch must be imported BEFORE libcvhip.so is dlopen'ed: the torch wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so, no SONAME) and loads it into the global symbol scope; libcvhip's hip* calls
# then resolve to that one runtime. Loaded the other way round, libcvhip binds /opt/rocm's copy and the
# process ends up with two HIP runtimes (second one reports "no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CVHIP_LIB: A/B of two BUILDS of the library in one gpurun call (tools/ab_env.sh); unset = the in-tree library
LIB_PATH = os.environ.get("CVHIP_LIB") or os.path.join(_HERE, "libcvhip.so")

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_LAUNCH = 0, -1, -2, -3
ACT_NONE, ACT_RELU, ACT_SILU, ACT_LEAKY, ACT_SIGMOID, ACT_HSWISH = 0, 1, 2, 3, 4, 5
DTYPE_F32, DTYPE_F64, DTYPE_I32, DTYPE_BF16, DTYPE_U8 = 0, 1, 2, 3, 4
RED_SUM, RED_MAX, RED_MIN = 0, 1, 2
DGRAD_CLASS_INTS = 12
REDUCE_SCRATCH_ROWS = 64  # CVHIP_REDUCE_SCRATCH_ROWS
YOLO_BIAS_ROWS = 256  # CVHIP_YOLO_BIAS_ROWS
BN_ACC_SHARDS = 16  # CVHIP_BN_ACC_SHARDS


class CvhipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """cvhip_conv_desc (include/cvhip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "N", "C", "H", "W", "K", "R", "S", "stride_h", "stride_w", "pad_h", "pad_w", "dil_h", "dil_w",
        "groups", "x_ld", "y_ld", "k_valid", "c_valid")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class BnTail(C.Structure):
    """cvhip_bn_tail (include/cvhip.h)."""
    _fields_ = [("y", C.c_void_p), ("y_ld", C.c_int32), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("act", C.c_int32), ("act_param", C.c_float), ("acc", C.c_void_p), ("acc_ld", C.c_int32)]


class ConvFuse(C.Structure):
    """cvhip_conv_fuse (include/cvhip.h): optional fused prologue / epilogue operands of cvhip_conv2d_fprop_fused."""
    _fields_ = [("bias", C.c_void_p), ("stats_partial", C.c_void_p), ("bn_acc", C.c_void_p), ("ep_scale", C.c_void_p),
                ("ep_shift", C.c_void_p), ("ep_act", C.c_int32), ("ep_act_param", C.c_float), ("pro_scale", C.c_void_p),
                ("pro_shift", C.c_void_p), ("pro_act", C.c_int32), ("pro_act_param", C.c_float), ("z_out", C.c_void_p),
                ("z_ld", C.c_int32), ("residual", C.c_void_p), ("residual_ld", C.c_int32), ("residual_pre", C.c_int32),
                ("x_image", C.c_void_p), ("x_image_planes", C.c_int32), ("pro_lo", C.c_int32), ("pro_hi", C.c_int32),
                ("y2", C.c_void_p), ("y2_ld", C.c_int32), ("y_split", C.c_int32)]


class LazyIn(C.Structure):
    """cvhip_lazy_in (include/cvhip.h): a lazily activated input operand of a backward kernel."""
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("act", C.c_int32), ("act_param", C.c_float), ("c_lo", C.c_int32),
                ("c_hi", C.c_int32)]


PATCH_CLASS_INTS = 30  # CVHIP_PATCH_CLASS_INTS
BAND_PLAN_INTS = 13    # CVHIP_BAND_PLAN_INTS
WGRAD_BAND_PLAN_INTS = 8   # include/cvhip.h CVHIP_WGRAD_BAND_PLAN_INTS


class PrepEntry(C.Structure):
    """cvhip_prep_entry (include/cvhip.h): one layer of a batched operand-preparation plan."""
    _fields_ = [("desc", ConvDesc), ("master", C.c_void_p), ("w_fprop", C.c_void_p), ("w_dgrad", C.c_void_p)]


class YoloLossDesc(C.Structure):
    """cvhip_yolo_loss_desc (include/cvhip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("N", "A", "NO", "H", "W", "ld", "T")] + [("anchor_t", C.c_float), ("anchors", C.c_float * 16)]


class OtaDesc(C.Structure):
    """cvhip_ota_desc (include/cvhip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("L", "N", "A", "NO", "T", "G")] + [("H", C.c_int32 * 4), ("W", C.c_int32 * 4), ("ld", C.c_int32 * 4),
                                                                            ("stride", C.c_float * 4), ("anchors", (C.c_float * 16) * 4),
                                                                            ("anchor_t", C.c_float), ("img_size", C.c_float)]


class SimotaDesc(C.Structure):
    """cvhip_simota_desc (include/cvhip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("L", "B", "A", "G", "nc")] + [("H", C.c_int32 * 4), ("W", C.c_int32 * 4), ("ld", C.c_int32 * 4),
                                                                       ("stride", C.c_float * 4)]


_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_dp = C.POINTER(ConvDesc)
_ylp = C.POINTER(YoloLossDesc)
_smp = C.POINTER(SimotaDesc)
_otp = C.POINTER(OtaDesc)
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/cvhip.h one-to-one
SIGNATURES = {
    "cvhip_version": (_i32, []),
    "cvhip_last_error": (C.c_char_p, []),
    "cvhip_conv2d_out_hw": (_i32, [_dp, C.POINTER(_i32), C.POINTER(_i32)]),
    "cvhip_conv2d_fprop_stats_rows": (_i32, [_dp]),
    "cvhip_conv1x1_stream_blocks": (_i32, [_i32, _i32, _i64, _i32]),
    "cvhip_conv_stem_blocks": (_i32, [_dp]),
    "cvhip_conv2d_dgrad_weight_elems": (_i64, [_dp]),
    "cvhip_conv2d_weight_image_elems": (_i64, [_dp, _i32]),
    "cvhip_conv2d_dgrad_plan": (_i32, [_dp, C.POINTER(_i32), _i32]),
    "cvhip_div31_consts": (_i32, [_i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "cvhip_conv2d_prep_weights": (_i32, [_dp, _p, _p, _p, _p]),
    "cvhip_prep_plan_item_bytes": (_i32, []),
    "cvhip_prep_plan_build": (_i32, [_p, _i32, _p, C.POINTER(_i32)]),
    "cvhip_prep_plan_run": (_i32, [_p, _i32, _i32, _p]),
    "cvhip_conv2d_fprop": (_i32, [_dp, _p, _p, _p, _p, _p, _p]),
    "cvhip_conv2d_fprop_fused": (_i32, [_dp, _p, _p, _p, C.POINTER(ConvFuse), _p]),
    "cvhip_conv2d_wgrad_image": (_i32, [_dp, _p, _i32, _p, _p, _p]),
    "cvhip_conv2d_fprop_prologue_ok": (_i32, [_dp, _i32]),
    "cvhip_conv2d_patch_plan": (_i32, [_dp, _i32, C.POINTER(_i32), _i32]),
    "cvhip_conv2d_band_plan": (_i32, [_dp, _i32, C.POINTER(_i32)]),
    "cvhip_conv2d_wgrad_band_plan": (_i32, [_dp, C.POINTER(_i32)]),
    "cvhip_conv1x1_stream_prologue_ok": (_i32, [_dp, _i32]),
    "cvhip_conv2d_wgrad_stem_bn": (_i32, [_dp, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _f32, _p, _p]),
    "cvhip_bn_finalize_acc": (_i32, [_p, _i32, _i32, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p]),
    "cvhip_bn_act_fwd_acc_lazyres": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _i32, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _i32, _f32,
                                     _p, _i32, _p, _p, _p]),
    "cvhip_conv1x1_bwd_fused_split": (_i32, [_dp, _p, _i32, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _f32, _p,
                                      _i32, _p, _i32, _p, C.POINTER(LazyIn), _p]),
    "cvhip_conv1x1_bwd_fused_lazy": (_i32, [_dp, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _f32, _p, _i32,
                                     _p, _i32, _p, C.POINTER(LazyIn), _p]),
    "cvhip_conv2d_dgrad": (_i32, [_dp, _p, _p, _p, _p]),
    "cvhip_conv2d_dgrad_add": (_i32, [_dp, _p, _p, _p, _i32, _p, _p]),
    "cvhip_conv2d_wgrad": (_i32, [_dp, _p, _p, _p, _i32, _p]),
    "cvhip_conv2d_wgrad_det_workspace_bytes": (_i64, [_dp]),
    "cvhip_conv2d_wgrad_det": (_i32, [_dp, _p, _p, _p, _i32, _p, _i64, _p]),
    "cvhip_conv1x1_bwd_fused_ok": (_i32, [_dp]),
    "cvhip_conv1x1_bwd_fused": (_i32, [_dp, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _f32, _p, _i32, _p, _i32,
                                _p, _p]),
    "cvhip_dwconv2d_fprop": (_i32, [_dp, _p, _p, _p, _p, _p]),
    "cvhip_dwconv2d_fprop_act": (_i32, [_dp, _p, _p, _p, _i32, _f32, _p, _p]),
    "cvhip_dwconv2d_fprop_stats_rows": (_i64, [_dp, _p, _p]),
    "cvhip_dwconv2d_fprop_stats": (_i32, [_dp, _p, _p, _p, _p, _p, _p]),
    "cvhip_dwconv2d_dgrad": (_i32, [_dp, _p, _p, _p, _p]),
    "cvhip_dwconv2d_wgrad": (_i32, [_dp, _p, _p, _p, _i32, _p]),
    "cvhip_colreduce_rows": (_i32, [_i64, _i32]),
    "cvhip_bn_stats_partial": (_i32, [_p, _i64, _i32, _i32, _p, _p]),
    "cvhip_bn_finalize": (_i32, [_p, _i32, _i32, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p]),
    "cvhip_bn_eval_scale_shift": (_i32, [_i32, _p, _p, _p, _p, _f32, _p, _p, _p]),
    "cvhip_bn_act_fwd": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _p, _i32, _f32, _p, _i32, _p]),
    "cvhip_bn_add_act_fwd": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _p, _i32, _f32, _p, _i32, _p]),
    "cvhip_bn_act_bwd_partial": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _p, _p, _p, _i32, _f32, _p, _p]),
    "cvhip_bn_bwd_finalize": (_i32, [_p, _i32, _i32, _p, _p, _p, _p, _p]),
    "cvhip_bn_act_bwd_apply": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i64, _i32, _p, _p, _p, _p, _p, _p, _i32, _f32, _p]),
    "cvhip_bn_acc_shards": (_i32, []),
    "cvhip_conv2d_fprop_acc": (_i32, [_dp, _p, _p, _p, _p, _p]),
    "cvhip_bn_act_fwd_acc": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _i32, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _i32, _f32, _p, _i32,
                             _i32, _p]),
    "cvhip_bn_act_bwd_sums_acc": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p, _p, _p, _p, _i32, _f32, _p, _i32, _p]),
    "cvhip_bn_tail_bwd_sums_acc": (_i32, [_p, _i32, _p, _i32, _p, _i32, _p, _i32, _i64, _i32, _p, _p, _i32, _f32, _p, _i32, _p]),
    "cvhip_bn_act_bwd_apply_acc": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _f32, _p]),
    "cvhip_conv1x1_bwd_fused_acc": (_i32, [_dp, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _f32, _p, _i32,
                                    _p, _i32, _p, _p, _p]),
    "cvhip_conv2d_dgrad_tail": (_i32, [_dp, _p, _p, _p, _i32, _p, _p, _p]),
    "cvhip_colsum_partial": (_i32, [_p, _i64, _i32, _i32, _p, _p]),
    "cvhip_colsum_finalize": (_i32, [_p, _i32, _i32, _p, _i32, _p]),
    "cvhip_maxpool2d_fwd": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_maxpool2d_bwd": (_i32, [_p, _i32, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_upsample2x_cat_fwd": (_i32, [_p, _i32, _i32, _p, _i32, _i32, _p, _i32, _i32, _i32, _i32, _p]),
    "cvhip_upsample2x_bwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_zero_fill": (_i32, [_p, _i64, _p]),
    "cvhip_f32_unpad_add": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "cvhip_copy2d": (_i32, [_p, _i32, _p, _i32, _i64, _i32, _p]),
    "cvhip_add2d": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i64, _i32, _p]),
    "cvhip_add_act_fwd": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i64, _i32, _i32, _f32, _p]),
    "cvhip_channel_scale_bwd_ds": (_i32, [_p, _i32, _p, _i32, _p, _i32, _i32, _i32, _p]),
    "cvhip_scale_nc": (_i32, [_p, _i32, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "cvhip_seg_ce_rows": (_i32, [_i64]),
    "cvhip_seg_ce_fwd": (_i32, [_p, _i32, _p, _i64, _i32, _i32, _p, _p, _p]),
    "cvhip_seg_ce_bwd": (_i32, [_p, _i32, _p, _i64, _i32, _i32, _p, _p, _p, _i32, _p]),
    "cvhip_seg_ce_bilinear_ok": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "cvhip_seg_ce_bilinear_fwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "cvhip_seg_ce_bilinear_bwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    "cvhip_seg_ce_bilinear_fwd_px": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "cvhip_seg_ce_bilinear_bwd_px": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    "cvhip_detail_boundary_targets": (_i32, [_p, _i32, _i32, _i32, _f32, _p, _p]),
    "cvhip_ohem_select_workspace_bytes": (_i64, []),
    "cvhip_ohem_select": (_i32, [_p, _i64, _i32, _f32, _f32, _p, _p, _p]),
    "cvhip_seg_ce_bilinear_bwd_ohem": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _f32, _p, _p, _p, _i32, _p]),
    "cvhip_resize_nearest_fwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_resize_nearest_bwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_resize_bilinear_fwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_resize_bilinear_bwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_resize_bilinear_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "cvhip_resize_bilinear_bwd_ws": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "cvhip_global_avgpool_fwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _p]),
    "cvhip_global_avgpool_bwd": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    "cvhip_nchw_f32_to_nhwc_bf16": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_focus_nchw_f32_to_nhwc_bf16": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_nhwc_bf16_to_nchw_f32": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _p]),
    "cvhip_nchw_f32_to_nhwc_bf16_ld": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_head_permute_fwd": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_head_permute_bwd": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cvhip_yolov5_decode": (_i32, [_p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _f32, _p, _i64, _i64, _p]),
    "cvhip_nms_workspace_bytes": (_i64, [_i32]),
    "cvhip_nms_sorted": (_i32, [_p, _i32, _f32, _p, _p, _p, _p]),
    "cvhip_detect_postprocess_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cvhip_detect_postprocess": (_i32, [_p, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "cvhip_sort_workspace_bytes": (_i64, [_i64]),
    "cvhip_argsort_desc_f32": (_i32, [_p, _i64, _p, _p, _p]),
    "cvhip_box_iou": (_i32, [_p, _i32, _p, _i32, _p, _p]),
    "cvhip_adamw_ema": (_i32, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _i32, _f32, _f32, _f32, _p, _f32, _f32, _p, _p, _p]),
    "cvhip_sgd_nesterov_ema": (_i32, [_p, _p, _p, _p, _i64, _p, _p, _p, _i32, _f32, _i32, _i32, _f32, _f32, _p, _p]),
    "cvhip_sgd_nesterov_ema_scaled": (_i32, [_p, _p, _p, _p, _i64, _p, _p, _p, _i32, _f32, _i32, _i32, _f32, _f32, _p, _p, _p]),
    "cvhip_loss_scale_check": (_i32, [_p, _i64, _p, _p]),
    "cvhip_loss_scale_update": (_i32, [_p, _p, _f32, _f32, _i32, _p]),
    "cvhip_i64_add": (_i32, [_p, _i64, _i64, _p]),
    "cvhip_ema_update": (_i32, [_p, _p, _i64, _f32, _p, _p]),
    "cvhip_u8_nhwc_to_bf16_norm": (_i32, [_p, _i64, _i32, _p, _i32, _p, _p, _p]),
    "cvhip_yolov5_loss_workspace_bytes": (_i64, [_ylp]),
    "cvhip_yolov5_loss_level_fwd": (_i32, [_ylp, _p, _p, _p, _p, _p]),
    "cvhip_yolov5_loss_finalize": (_i32, [_p, _i32, _p, _p, _f32, _f32, _f32, _i32, _f32, _p, _p, _p]),
    "cvhip_yolov5_loss_level_bwd": (_i32, [_ylp, _p, _p, _p, _p, _p, _f32, _f32, _f32, _p, _p]),
    "cvhip_yolov5_loss_level_bwd_bias": (_i32, [_ylp, _p, _p, _p, _p, _p, _f32, _f32, _f32, _p, _p, _p]),
    "cvhip_ota_workspace_bytes": (_i64, [_otp]),
    "cvhip_ota_assign": (_i32, [_otp, _pp, _p, _p, _p, _p]),
    "cvhip_ota_read_overflow": (_i32, [_otp, _p, _p, _p]),
    "cvhip_yolov5_loss_level_fwd_assigned": (_i32, [_ylp, _p, _p, _p, _p, _p, _p]),
    "cvhip_simota_workspace_bytes": (_i64, [_smp]),
    "cvhip_simota_loss_fwd": (_i32, [_smp, _pp, _p, _p, _p, _p]),
    "cvhip_simota_loss_bwd": (_i32, [_smp, _pp, _p, _p, _p, _pp, _p]),
    "cvhip_simota_read_assignment": (_i32, [_smp, _p, _p, _p, _p]),
    "cvhip_comm_available": (_i32, []),
    "cvhip_comm_rccl_version": (_i32, []),
    "cvhip_comm_unique_id_bytes": (_i32, []),
    "cvhip_comm_get_unique_id": (_i32, [_p]),
    "cvhip_comm_init_rank": (_i32, [_pp, _i32, _i32, _p]),
    "cvhip_comm_destroy": (_i32, [_p]),
    "cvhip_comm_world": (_i32, [_p]),
    "cvhip_comm_rank": (_i32, [_p]),
    "cvhip_allreduce_bucket": (_i32, [_p, _p, _i64, _p]),
    "cvhip_comm_allreduce": (_i32, [_p, _p, _i64, _i32, _i32, _p]),
    "cvhip_comm_broadcast": (_i32, [_p, _p, _i64, _i32, _p]),
    "cvhip_comm_reduce_scatter_f32": (_i32, [_p, _p, _i64, _p]),
    "cvhip_comm_all_gather_f32": (_i32, [_p, _p, _i64, _p]),
}

# libcvhip_probes.so (include/cvhip_probes.h): known-answer and machine-ceiling kernels for tests/ and tools/ — not part of the
# product library, loaded only when one of them is called
PROBE_SIGNATURES = {
    "cvhip_probe_grid_barrier": (_i32, [_i32, _i32, _i32, _p, _p, _p, _p]),
    "cvhip_probe_mfma_16x16x32": (_i32, [_p, _p, _p, _p]),
    "cvhip_probe_ds_read_tr16": (_i32, [_p, _p, _p]),
    "cvhip_probe_lds_read_bw": (_i32, [_i32, _i32, _i32, _p, _p]),
    "cvhip_probe_mfma_peak": (_i32, [_i32, _i32, _p, _p]),
    "cvhip_probe_lds_read2": (_i32, [_i32, _i32, _i32, _i32, _p, _p]),
    "cvhip_probe_mfma_peak2": (_i32, [_i32, _i32, _i32, _i32, _i32, _p, _p]),
    "cvhip_probe_load_path": (_i32, [_i32, _i32, _p, _i64, _i64, _i32, _i32, _i32, _p, _p]),
    "cvhip_probe_atomic_add": (_i32, [_i32, _p, _i32, _i32, _i32, _p]),
    "cvhip_probe_stage": (_i32, [_i32, _p, _i64, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "cvhip_probe_gather": (_i32, [_i32, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
}
PROBES_LIB_PATH = os.path.join(_HERE, "libcvhip_probes.so")

_lib = None
_probes = None
_F16 = set()        # entry points that exist a second time with the suffix _f16 (fp16 storage: csrc/common.h)
PRECISION = "bf16"  # storage precision the engine currently runs in: `call` routes to the _f16 symbols when "fp16"


def set_precision(p):
    """Select the storage precision of activations / operand images / activation gradients: "bf16" (default) or "fp16"
    (reference: autocast fp16, trainer.py:179). Process-wide, like the reference's autocast context."""
    global PRECISION
    if p not in ("bf16", "fp16"):
        raise CvhipError("precision must be 'bf16' or 'fp16'")
    load()
    PRECISION = p


def load():
    """Load libcvhip.so (once). Raises CvhipError if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CvhipError(
            "libcvhip.so not found at %s — build it with `python -m cvpytorch_amd.build` "
            "(hipcc --offload-arch=gfx950). The HIP engine has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("CVHIP_LIB") and not hasattr(lib, name):
            continue  # an older build under A/B (CVHIP_LIB): entry points added since are simply absent from it
        fn = getattr(lib, name)  # AttributeError here == ABI drift; let it propagate loudly
        fn.restype = res
        fn.argtypes = args
        try:
            f16 = getattr(lib, name + "_f16")
        except AttributeError:
            continue
        f16.restype = res
        f16.argtypes = args
        _F16.add(name)
    if lib.cvhip_version() < 100:
        raise CvhipError("libcvhip.so too old")
    _lib = lib
    return lib


def load_probes():
    """Load libcvhip_probes.so (once). Raises CvhipError if it has not been built."""
    global _probes
    if _probes is not None:
        return _probes
    if not os.path.exists(PROBES_LIB_PATH):
        raise CvhipError("libcvhip_probes.so not found at %s — build it with `python -m cvpytorch_amd.build`" % PROBES_LIB_PATH)
    lib = C.CDLL(PROBES_LIB_PATH)
    for name, (res, args) in PROBE_SIGNATURES.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    lib.cvhip_probes_last_error.restype = C.c_char_p
    lib.cvhip_probes_last_error.argtypes = []
    _probes = lib
    return lib


def check(status, what):
    if status != OK and what in PROBE_SIGNATURES:
        last = (load_probes().cvhip_probes_last_error() or b"").decode()
        msg = {ERR_INVALID: "invalid argument", ERR_UNSUPPORTED: "unsupported shape/feature",
               ERR_LAUNCH: "HIP launch error: " + last}.get(status, "status %d" % status)
        raise CvhipError("%s failed: %s" % (what, msg))
    if status != OK:
        lib = load()
        # the library keeps one error string per build of the sources: the 16-bit-typed entry points of the active precision and the
        # single-precision sources (communicator, post-processing) report through different copies — show whichever is set
        errs = [(getattr(lib, n)() or b"").decode() for n in ("cvhip_last_error", "cvhip_last_error_f16") if hasattr(lib, n)]
        if PRECISION == "fp16":
            errs.reverse()
        last = "; ".join(dict.fromkeys(e for e in errs if e))
        msg = {ERR_INVALID: "invalid argument", ERR_UNSUPPORTED: "unsupported shape/feature",
               ERR_LAUNCH: "HIP launch error: " + last}.get(status, "status %d" % status)
        raise CvhipError("%s failed: %s" % (what, msg))


def fn(name):
    """The entry point `name` of the active precision (cvhip_probe_*: of libcvhip_probes.so, bf16 forms only)."""
    if name in PROBE_SIGNATURES:
        load()  # torch's HIP runtime first, as for the product library
        return getattr(load_probes(), name)
    lib = load()
    if PRECISION == "fp16" and name in _F16:
        name += "_f16"
    return getattr(lib, name)


def call(name, *args):
    """Invoke an int-status entry point (of the active precision) and raise on failure."""
    check(fn(name)(*args), name)
