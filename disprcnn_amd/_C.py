"""`disprcnn._C` -- the names of the reference's compiled extension (csrc/vision.cpp:7-15), served by libdisprcnn_hip.so.

The reference's Python layers bind their kernels with `from disprcnn import _C` (layers/roi_align.py:3, layers/nms.py:3) and call
`_C.nms`, `_C.roi_align_forward`, `_C.roi_align_backward` with the pybind signatures of csrc/nms.h:12-28 and csrc/ROIAlign.h:12-46.  This
module keeps exactly those call signatures (positional, same order and meaning) over the C ABI (`drc_nms_sorted_fwd`, `drc_roi_align_fwd`,
`drc_roi_align_bwd`, include/disprcnn_hip.h), so a maintainer who keeps the reference's own `layers/*.py` gets the HIP kernels through the
alias package (`disprcnn/__init__.py`).  GPU tensors only: the reference's CPU kernels are not part of this build (its CPU backward raises
"Not implemented on the CPU" itself, csrc/ROIAlign.h:44); a CPU tensor raises RuntimeError here as well.

`roi_pool_*` and `sigmoid_focalloss_*` (vision.cpp:11-14) belong to heads the shipped KITTI configs never build (ROIPool: C4 models;
focal loss: RetinaNet) -- outside the hot path: present as names, they raise NotImplementedError.
"""
from .layers.nms import nms as _nms
from .layers.roi_align import roi_align_backward as _roi_align_backward
from .layers.roi_align import roi_align_forward as _roi_align_forward


def nms(dets, scores, threshold):
    """csrc/nms.h:12-28: dets [N,4] xyxy, scores [N], threshold -> int64 indices of the kept boxes in ascending order (the CUDA op's
    convention: suppress when IoU > threshold, legacy +1 pixel widths, csrc/cuda/nms.cu:23-131)."""
    return _nms(dets, scores, float(threshold))


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """csrc/ROIAlign.h:12-26 -> csrc/cuda/ROIAlign_cuda.cu:256-301: input [B,C,H,W], rois [K,5] (batch_idx, x1, y1, x2, y2) -> [K,C,ph,pw]."""
    return _roi_align_forward(input, rois, float(spatial_scale), int(pooled_height), int(pooled_width), int(sampling_ratio))


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio):
    """csrc/ROIAlign.h:28-46 -> csrc/cuda/ROIAlign_cuda.cu:304-346: grad [K,C,ph,pw] -> the input's gradient [B,C,H,W]."""
    return _roi_align_backward(grad, rois, float(spatial_scale), int(pooled_height), int(pooled_width), int(batch_size), int(channels),
                               int(height), int(width), int(sampling_ratio))


def _out_of_scope(name, where):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"disprcnn._C.{name}: {where} is outside the instance-disparity hot path and not built on MI355X (SURVEY 2)")
    fn.__name__ = name
    return fn


roi_pool_forward = _out_of_scope("roi_pool_forward", "ROIPool (csrc/ROIPool.h; C4 heads)")
roi_pool_backward = _out_of_scope("roi_pool_backward", "ROIPool (csrc/ROIPool.h; C4 heads)")
sigmoid_focalloss_forward = _out_of_scope("sigmoid_focalloss_forward", "SigmoidFocalLoss (csrc/SigmoidFocalLoss.h; RetinaNet)")
sigmoid_focalloss_backward = _out_of_scope("sigmoid_focalloss_backward", "SigmoidFocalLoss (csrc/SigmoidFocalLoss.h; RetinaNet)")
