"""ROIAlign operator -- drop-in for ``disprcnn.layers.ROIAlign`` / ``roi_align`` (reference layers/roi_align.py:13-73).

Forward and backward run in libdisprcnn_hip.so (drc_roi_align_fwd / drc_roi_align_bwd); the reference's CPU
forward + "Not implemented on the CPU" backward (csrc/ROIAlign.h:21,44) become: GPU only, RuntimeError otherwise.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _lib
from .. import engine as E


def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, mean=None, std=None):
    """Functional mirror of ``_C.roi_align_forward`` (csrc/vision.cpp:9); ``mean``/``std`` fuse the crop normalisation."""
    E.require_gpu(input, "roi_align_forward")
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise RuntimeError("rois must be [K,5] (batch_idx, x1, y1, x2, y2)")
    input = input.contiguous()
    rois = rois.to(device=input.device, dtype=torch.float32).contiguous()
    K, (B, Cc, H, W) = rois.shape[0], input.shape
    out = torch.empty(K, Cc, pooled_height, pooled_width, dtype=torch.float32, device=input.device)
    st = _lib.lib().drc_roi_align_fwd(E._ptr(input), E._ptr(rois), E._ptr(out), K, Cc, H, W, pooled_height, pooled_width,
                                      float(spatial_scale), int(sampling_ratio), E._ptr(mean), E._ptr(std), E._stream_ptr(input.device))
    _lib.check(st, "drc_roi_align_fwd")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio):
    """Functional mirror of ``_C.roi_align_backward`` (csrc/vision.cpp:10)."""
    E.require_gpu(grad, "roi_align_backward")
    grad = grad.contiguous()
    rois = rois.to(device=grad.device, dtype=torch.float32).contiguous()
    gin = torch.zeros(batch_size, channels, height, width, dtype=torch.float32, device=grad.device)
    st = _lib.lib().drc_roi_align_bwd(E._ptr(grad), E._ptr(rois), E._ptr(gin), rois.shape[0], channels, height, width, pooled_height,
                                      pooled_width, float(spatial_scale), int(sampling_ratio), E._stream_ptr(grad.device))
    _lib.check(st, "drc_roi_align_bwd")
    return gin


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale, ctx.sampling_ratio, ctx.input_shape = spatial_scale, sampling_ratio, input.size()
        return roi_align_forward(input, roi, spatial_scale, ctx.output_size[0], ctx.output_size[1], sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        g = roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0], ctx.output_size[1], bs, ch, h, w,
                               ctx.sampling_ratio)
        return g, None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois, spatial_scale=None):
        return roi_align(input, rois, self.output_size, self.spatial_scale if spatial_scale is None else spatial_scale,
                         self.sampling_ratio)

    def __repr__(self):
        return (f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio})")
