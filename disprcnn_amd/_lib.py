"""ctypes binding of libdisprcnn_hip.so (the C ABI declared in include/disprcnn_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, ``lib()`` raises.  Status codes are turned into ``RuntimeError`` the way the
reference turns ``AT_ASSERTM``/``THCudaCheck`` failures into Python errors
(reference: csrc/ROIAlign.h:21,44; csrc/cuda/ROIAlign_cuda.cu:297).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported BEFORE the .so: both must share torch's libamdhip64 runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdisprcnn_hip.so")

DRC_MAX_CLASSES = 8


class DrcTapClass(C.Structure):
    _fields_ = [("nd", C.c_int32), ("nh", C.c_int32), ("nw", C.c_int32),
                ("dd0", C.c_int32), ("dh0", C.c_int32), ("dw0", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("wbase", C.c_int32), ("wsd", C.c_int32), ("wsh", C.c_int32), ("wsw", C.c_int32),
                ("out_off_d", C.c_int32), ("out_off_h", C.c_int32), ("out_off_w", C.c_int32)]


class DrcTapconvParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("res", C.c_void_p), ("y", C.c_void_p),
                ("x_n_stride", C.c_int64), ("x_cb_stride", C.c_int64), ("x_d_stride", C.c_int64), ("x_h_stride", C.c_int64),
                ("y_n_stride", C.c_int64), ("y_cb_stride", C.c_int64), ("y_d_stride", C.c_int64), ("y_h_stride", C.c_int64),
                ("y_off0", C.c_int64),
                ("r_n_stride", C.c_int64), ("r_cb_stride", C.c_int64), ("r_d_stride", C.c_int64), ("r_h_stride", C.c_int64),
                ("r_off0", C.c_int64),
                ("N", C.c_int32), ("OD", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
                ("in_mul", C.c_int32), ("out_mul", C.c_int32), ("cb_in", C.c_int32), ("cout_pad", C.c_int32),
                ("R", C.c_int32), ("WT", C.c_int32), ("relu", C.c_int32), ("n_classes", C.c_int32),
                ("lds_bytes_per_wave", C.c_int32), ("reserved", C.c_int32),
                ("cls", DrcTapClass * DRC_MAX_CLASSES)]


class DrcCostvolSrc(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p),
                ("n_stride", C.c_int64), ("cb_stride", C.c_int64), ("h_stride", C.c_int64),
                ("cbi", C.c_int32), ("pad", C.c_int32), ("lo4", C.c_int32), ("Wp", C.c_int32)]


class DrcS16ConvParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("res", C.c_void_p),
                ("y16", C.c_void_p), ("y32", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32), ("lo4", C.c_int32), ("dil", C.c_int32),
                ("head", C.c_void_p), ("w1", C.c_void_p), ("ovf", C.c_void_p)]


class DrcFpnPyramid(C.Structure):
    _fields_ = [("feat", C.c_void_p * 8), ("H", C.c_int32 * 8), ("W", C.c_int32 * 8), ("scale", C.c_float * 8), ("n_levels", C.c_int32)]


class DrcWgradParams(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("gw", C.c_void_p),
                ("a_n_stride", C.c_int64), ("a_cb_stride", C.c_int64), ("a_d_stride", C.c_int64), ("a_h_stride", C.c_int64),
                ("b_n_stride", C.c_int64), ("b_cb_stride", C.c_int64), ("b_d_stride", C.c_int64), ("b_h_stride", C.c_int64),
                ("b_off0", C.c_int64),
                ("N", C.c_int32), ("OD", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
                ("in_mul", C.c_int32), ("cb_a", C.c_int32), ("cb_b", C.c_int32),
                ("nd", C.c_int32), ("nh", C.c_int32), ("nw", C.c_int32), ("dd0", C.c_int32), ("dh0", C.c_int32), ("dw0", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("R", C.c_int32), ("WT", C.c_int32), ("lds_bytes_per_wave", C.c_int32), ("overwrite", C.c_int32),
                ("scratch", C.c_void_p), ("scratch_floats", C.c_int64)]


WGRAD_SCRATCH_FLOATS = (1024 + 3 * 64) * 49 * 256      # DRC_WGRAD_SCRATCH_FLOATS
LOSS_SCRATCH_FLOATS = 1024 * 8                          # DRC_LOSS_SCRATCH_FLOATS


def cout1_wgrad_scratch_floats(cb_in):                  # DRC_COUT1_WGRAD_SCRATCH_FLOATS
    return 1024 * cb_in * 27 * 16


_P = C.c_void_p
_I = C.c_int
_SIGS = {
    "drc_version": (C.c_char_p, []),
    "drc_cost_volume_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_cost_volume_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_cost_volume_blocked_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_dense_to_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_blocked_to_dense": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_tapconv_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_tapconv3d_direct_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_deconv3d_k3s2_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_deconv3d_k3s2_direct_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_conv3d_k3s2_direct_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_conv3d_k3_wino_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_linear_scratch_floats": (C.c_int64, [_I, _I, _I]),
    "drc_linear_packed_floats": (C.c_int64, [_I, _I]),
    "drc_linear_pack_rows": (_I, [_P, _I, _I, _P, _P]),
    "drc_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, C.c_int64, _P]),
    "drc_conv3d_k3_wino_rb_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_conv3d_k3_wino_rb_supported": (_I, [_I, _I, _I, _I]),
    "drc_conv2d_k3_wino_rb_supported": (_I, [_I, _I, _I]),
    "drc_conv2d_k3_wino_rb_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_pack_weights_wino2d_rb": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "drc_conv3d_k3_wino_rb_costvol_fwd": (_I, [C.POINTER(DrcTapconvParams), C.POINTER(DrcCostvolSrc), _P]),
    "drc_pack_weights_wino_rb": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "drc_box_decode_fwd": (_I, [_P, _P, _P, C.c_int64, _I, _I, _P, C.c_float, C.c_float, C.c_float, _P]),
    "drc_srpn_proposals_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, C.c_int64, C.c_int64, C.c_float, _P, _P, _P, _P]),
    "drc_conv3d_k3_wino_costvol_fwd": (_I, [C.POINTER(DrcTapconvParams), C.POINTER(DrcCostvolSrc), _I, _P]),
    "drc_disparity_paste_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "drc_disparity_resize_fwd": (_I, [_P, _I, _I, _P, _I, _I, _I, _P]),
    "drc_roi_depth_maps_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "drc_conv2d_k3_wino_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_pack_weights_wino2d": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "drc_pack_weights_wino": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "drc_conv2d_k1_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_conv2d_k3_direct_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_conv3d_cout1_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "drc_upsample_softargmin_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_avgpool2d_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_pack_weights": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "drc_avgpool2d_blocked_slice": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_bilinear_up_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_bilinear_resize_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_maxpool2d_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_copy_blocks": (_I, [_P, _P, _I, _I, C.c_int64, _I, _I, _P]),
    "drc_roi_align_fpn_fwd": (_I, [C.POINTER(DrcFpnPyramid), _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "drc_roi_align_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P, _P, _P]),
    "drc_roi_align_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _I, _P]),
    "drc_align_roi_pairs": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "drc_conv16_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_dense_to_blocked16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "drc_avgpool2d_blocked16_slice": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_bilinear_up_blocked16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_cost_volume16_from16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_conv16_k3_tile_supported": (_I, [C.POINTER(DrcTapconvParams)]),
    "drc_conv16_k3_tile_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_conv2d_k3s2_stem_fwd": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _I, _I, _I, _P]),
    "drc_conv16_k3_costvol_fwd": (_I, [C.POINTER(DrcTapconvParams), _I, _P]),
    "drc_conv16_k3s2_tile_supported": (_I, [C.POINTER(DrcTapconvParams)]),
    "drc_conv16_k3s2_tile_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_deconv16_k3s2_tile_supported": (_I, [C.POINTER(DrcTapconvParams)]),
    "drc_deconv16_k3s2_tile_fwd": (_I, [C.POINTER(DrcTapconvParams), _P]),
    "drc_cost_volume16_blocked_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_conv3d_k3_s16_supported": (_I, [_I, _I, _I, _I, _I]),
    "drc_conv3d_k3_s16_fwd": (_I, [C.POINTER(DrcS16ConvParams), _P]),
    "drc_conv3d_k3_s16_wide": (_I, [C.POINTER(DrcS16ConvParams)]),
    "drc_conv3d_k3_s16_wide_fwd": (_I, [C.POINTER(DrcS16ConvParams), _P]),
    "drc_conv3d_k3s2_s16_supported": (_I, [_I, _I, _I, _I, _I]),
    "drc_conv3d_k3s2_s16_fwd": (_I, [C.POINTER(DrcS16ConvParams), _P]),
    "drc_deconv3d_k3s2_s16_supported": (_I, [_I, _I, _I, _I, _I]),
    "drc_deconv3d_k3s2_s16_fwd": (_I, [C.POINTER(DrcS16ConvParams), _P]),
    "drc_rs16_from_dense": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "drc_rs16_from_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "drc_rs16_to_dense": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "drc_rs16_to_blocked": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "drc_head_gather_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, C.c_float, _P]),
    "drc_conv2d_k3_s16_supported": (_I, [_I, _I, _I, _I, _I]),
    "drc_conv2d_k3_s16_fwd": (_I, [C.POINTER(DrcS16ConvParams), _P]),
    "drc_deconv3d_k3s2_direct_s16_fwd": (_I, [C.POINTER(DrcTapconvParams), _P, _P, _P]),
    "drc_nms_sorted_fwd": (_I, [_P, _I, C.c_float, _I, _P, _P, _P]),
    "drc_nms_sorted_batch_fwd": (_I, [_P, _I, _I, C.c_float, _I, _P, _P, _P]),
    "drc_nms_sorted_pair_joint_fwd": (_I, [_P, _I, C.c_float, _I, _I, _P, _P, _P]),
    "drc_roi_train_targets_fwd": (_I, [_P, _P, _P, _I, _I, C.c_float, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "drc_bn_stats_blocked": (_I, [_P, _P, _P, _P, _P]),
    "drc_bn_finalize": (_I, [_P, _I, _I, C.c_longlong, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "drc_bn_apply_blocked": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "drc_tapconv_wgrad": (_I, [C.POINTER(DrcWgradParams), _P]),
    "drc_bilinear_up_blocked_bwd": (_I, [_P, _P, _P, _P, _P]),
    "drc_avgpool2d_blocked_bwd": (_I, [_P, _P, _P, _P, _I, _P]),
    "drc_upsample_softargmin_bwd_scratch_floats": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "drc_upsample_softargmin_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, C.c_int64, _P]),
    "drc_conv3d_cout1_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "drc_conv3d_cout1_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "drc_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "drc_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _I, _P, _P, _P, _P, _I, _P]),
    "drc_psm_loss_sums": (_I, [_P, _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "drc_psm_loss_grad": (_I, [_P, _P, _P, C.c_int64, _P, C.c_float, _P, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or python -m disprcnn_amd.csrc.build). "
                "There is no CPU/torch fallback for this path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        kind = "bad argument / unsupported shape" if status < 0 else "hipError_t"
        raise RuntimeError(f"{what} failed: status {status} ({kind})")
