"""ctypes binding of libmtbt_hip.so (C ABI: include/mtbt_hip.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmtbt_hip.so")

ABI_VERSION = 5          # include/mtbt_hip.h MTBT_ABI_VERSION
F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_ELU, ACT_GELU, ACT_GELU_POLY, ACT_DSILU, ACT_DELU, ACT_DGELU, ACT_DGELU_POLY = 0, 1, 2, 3, 4, 5, 6, 7, 8
OUT_NHWC, OUT_CONVT2X2 = 0, 1
RES_ID, RES_UP_BILINEAR, RES_DOWN_MEAN, RES_UP_NEAREST, RES_MAXPOOL = 0, 1, 2, 3, 4

ERRORS = {-1: "MTBT_EINVAL (bad argument / unsupported shape)", -2: "MTBT_EALIGN (misaligned pointer or stride)",
          -3: "MTBT_ELAUNCH (kernel launch failed)", -4: "MTBT_EWORKSPACE (workspace too small)"}


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("res", C.c_void_p),
        ("x_batch_stride", C.c_int64), ("y_batch_stride", C.c_int64), ("res_batch_stride", C.c_int64),
        ("x_pixel_stride", C.c_int32), ("y_pixel_stride", C.c_int32), ("res_pixel_stride", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("K", C.c_int32),
        ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("act", C.c_int32), ("out_mode", C.c_int32), ("tile_hint", C.c_int32), ("y2", C.c_void_p),
        ("policy", C.c_int32), ("debug", C.c_int32),
        ("colsum", C.c_void_p), ("colsum_shift", C.c_void_p), ("colsum_sq", C.c_int32), ("colsum_accumulate", C.c_int32),
        ("colsum_ws", C.c_void_p), ("colsum_ws_bytes", C.c_int64),
    ]


class FuseArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p * 3), ("wgt", C.c_float * 3), ("resample", C.c_int32 * 3), ("n_in", C.c_int32),
        ("y", C.c_void_p), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("dtype", C.c_int32), ("add_weight_bug", C.c_int32), ("wgt_dev", C.c_void_p),
    ]


class NodeArgs(C.Structure):  # mtbt_node_args
    _fields_ = [("fuse", FuseArgs), ("w", C.c_void_p), ("shift", C.c_void_p), ("y", C.c_void_p), ("y_pixel_stride", C.c_int32), ("K", C.c_int32),
                ("act", C.c_int32)]


class UpconvArgs(C.Structure):  # mtbt_upconv_args
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("shift", C.c_void_p),
                ("x_batch_stride", C.c_int64), ("y_batch_stride", C.c_int64), ("x_pixel_stride", C.c_int32), ("y_pixel_stride", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("dtype", C.c_int32), ("act", C.c_int32)]


class DecodeArgs(C.Structure):
    _fields_ = [
        ("map", C.c_void_p * 3), ("h", C.c_int32 * 3), ("w", C.c_int32 * 3), ("map_pixel_stride", C.c_int32 * 3),
        ("stride", C.c_float * 3), ("n_levels", C.c_int32), ("N", C.c_int32), ("nc", C.c_int32),
        ("reg_max", C.c_int32), ("xywh", C.c_int32),
        ("boxes", C.c_void_p), ("scores", C.c_void_p), ("best_score", C.c_void_p), ("best_label", C.c_void_p),
        ("preds_cat", C.c_void_p), ("cat_stride", C.c_int32),
    ]


class MaskArgs(C.Structure):
    _fields_ = [
        ("protos", C.c_void_p), ("coeff", C.c_void_p),
        ("coeff_batch_stride", C.c_int64), ("coeff_k_stride", C.c_int64), ("coeff_c_stride", C.c_int64),
        ("gather_idx", C.c_void_p), ("counts", C.c_void_p), ("bias", C.c_float),
        ("N", C.c_int32), ("K", C.c_int32), ("nm", C.c_int32), ("hp", C.c_int32), ("wp", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("logits", C.c_void_p), ("masks", C.c_void_p),
    ]


class LossArgs(C.Structure):
    _fields_ = [
        ("map", C.c_void_p * 3), ("h", C.c_int32 * 3), ("w", C.c_int32 * 3), ("map_pixel_stride", C.c_int32 * 3),
        ("n_levels", C.c_int32), ("N", C.c_int32), ("nc", C.c_int32), ("reg_max", C.c_int32), ("img_size", C.c_float),
        ("gt_xyxy", C.c_void_p), ("gt_cls", C.c_void_p), ("gt_off", C.c_void_p),
        ("iou_thresh", C.c_float), ("label_smoothing", C.c_float), ("training", C.c_int32),
        ("seg_logits", C.c_void_p), ("seg_targets", C.c_void_p), ("seg_bias", C.c_void_p), ("seg_n", C.c_int64),
        ("img_logits", C.c_void_p), ("img_gt", C.c_void_p), ("n_img_classes", C.c_int32),
        ("w_seg", C.c_float), ("w_box", C.c_float), ("w_dfl", C.c_float), ("w_cls", C.c_float), ("w_img", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("out", C.c_void_p),
    ]


# every symbol include/mtbt_hip.h declares: name -> (restype, argtypes)
class PrepDesc(C.Structure):  # mtbt_prep_desc
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("scale0", C.c_void_p), ("scale1", C.c_void_p),
                ("sstride", C.c_int64 * 4), ("dim", C.c_int32 * 4), ("flip", C.c_int32 * 4),
                ("scale0_dim", C.c_int32), ("scale1_dim", C.c_int32), ("dst_dtype", C.c_int32), ("src_dim3", C.c_int32)]


class RawImage(C.Structure):  # mtbt_raw_image
    _fields_ = [("bgr", C.c_void_p), ("mask", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32),
                ("row_stride", C.c_int64), ("mask_row_stride", C.c_int64)]


SYMBOLS = {
    "mtbt_abi_version": (C.c_int, []),
    "mtbt_sizeof_args": (C.c_int, [C.c_int]),
    "mtbt_target_arch": (C.c_char_p, []),
    "mtbt_conv2d_nhwc": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "mtbt_convt2x2_conv3x3_nhwc": (C.c_int, [C.POINTER(UpconvArgs), C.c_void_p]),
    "mtbt_stem_conv4x4_ln": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "mtbt_dwconv_nhwc": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
                         + [C.c_int] * 6 + [C.c_void_p]),
    "mtbt_layernorm_nhwc": (C.c_int, [C.c_void_p] * 3 + [C.c_float, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_bifpn_fuse": (C.c_int, [C.POINTER(FuseArgs), C.c_void_p]),
    "mtbt_bifpn_node_nhwc": (C.c_int, [C.POINTER(NodeArgs), C.c_void_p]),
    "mtbt_bn_train_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "mtbt_bn_train_nhwc": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int64, C.c_void_p]),
    "mtbt_gap_fc": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "mtbt_decode_boxes": (C.c_int, [C.POINTER(DecodeArgs), C.c_void_p]),
    "mtbt_nms_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "mtbt_nms_batched": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
                         + [C.c_void_p] * 8 + [C.c_int64, C.c_void_p]),
    "mtbt_mask_assemble": (C.c_int, [C.POINTER(MaskArgs), C.c_void_p]),
    "mtbt_loss_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int64]),
    "mtbt_multitask_loss": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "mtbt_multitask_loss_grad": (C.c_int, [C.POINTER(LossArgs), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtbt_convnext_mlp_fused": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_void_p]),
    "mtbt_convnext_mlp_fused_dt": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_convnext_mlp_fused_train": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_void_p]),
    "mtbt_bbox_iou_pairwise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "mtbt_letterbox_batch": (C.c_int, [C.POINTER(RawImage), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "mtbt_seg_confusion_workspace_bytes": (C.c_int64, [C.c_int]),
    "mtbt_seg_confusion": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_conv_wgrad_workspace_bytes": (C.c_int64, [C.c_int] * 7),
    "mtbt_conv_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_void_p,
                                  C.c_int64, C.c_void_p]),
    "mtbt_conv_wgrad_xact": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 9 + [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "mtbt_conv_wgrad_bias": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "mtbt_act_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_channel_sum_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "mtbt_channel_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_int64, C.c_void_p]),
    "mtbt_channel_affine2": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_layernorm_backward_nhwc": (C.c_int, [C.c_void_p] * 3 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_layernorm_backward_params_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "mtbt_layernorm_backward_params_nhwc": (C.c_int, [C.c_void_p] * 3 + [C.c_float, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                      C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_dwconv_wgrad_workspace_bytes": (C.c_int64, [C.c_int] * 5),
    "mtbt_dwconv_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_dwconv_wgrad_bias": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_adamw_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 5 + [C.c_int64, C.c_void_p, C.c_void_p]),
    "mtbt_sgd_step": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_float] * 4 + [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "mtbt_sumsq_workspace_bytes": (C.c_int64, []),
    "mtbt_sumsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_clip_coef": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtbt_stem_conv4x4_ln_train": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "mtbt_dwconv_nhwc_train": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
                               + [C.c_int] * 6 + [C.c_void_p]),
    "mtbt_bn_forward_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int64, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_bn_forward_sums_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int64, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtbt_conv_colsum_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "mtbt_conv_colsum_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtbt_conv_kernel_choice": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtbt_bn_forward_partials_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtbt_bn_backward_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "mtbt_bn_backward_nhwc": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 3
                              + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_weight_prep_blocks": (C.c_int, [C.c_int64]),
    "mtbt_weight_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_bifpn_norm_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "mtbt_bifpn_norm_weights_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mtbt_wadd_norm_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "mtbt_wadd_norm_weights_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mtbt_resample_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p]),
    "mtbt_bifpn_fuse_backward_workspace_bytes": (C.c_int64, []),
    "mtbt_bifpn_fuse_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 5
                                 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_projector_backward_workspace_bytes": (C.c_int64, [C.c_int] * 4),
    "mtbt_projector_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 6
                                + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_gap_fc_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "mtbt_copy_strided": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_int, C.c_int64,
                                    C.c_int, C.c_int, C.c_void_p]),
    "mtbt_scale_grad": (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_add_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mtbt_stem_wgrad_workspace_bytes": (C.c_int64, [C.c_int]),
    "mtbt_stem_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p, C.c_int64, C.c_void_p]),
    "mtbt_cast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
}

ARG_STRUCTS = (ConvArgs, FuseArgs, DecodeArgs, MaskArgs, LossArgs, PrepDesc, RawImage, UpconvArgs, NodeArgs)   # order of mtbt_sizeof_args(which)
_lib = None


def load():
    """Load the HIP library or raise: the product has no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m multitask_bonetumor_yolo_amd.build` "
                               "(there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mtbt_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libmtbt_hip.so reports ABI version {lib.mtbt_abi_version()}, this binding is version {ABI_VERSION}: rebuild "
                               "(`python -m multitask_bonetumor_yolo_amd.build`)")
        for which, st in enumerate(ARG_STRUCTS):
            if lib.mtbt_sizeof_args(which) != C.sizeof(st):
                raise RuntimeError(f"libmtbt_hip.so was built with sizeof({st.__name__}) = {lib.mtbt_sizeof_args(which)}, this binding lays it out in "
                                   f"{C.sizeof(st)} bytes: stale library, rebuild")
        _lib = lib
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what}: {ERRORS.get(code, code)}")
