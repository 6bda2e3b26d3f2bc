"""PSMLoss / EndPointErrorLoss -- drop-ins for ``disprcnn.utils.loss_utils.PSMLoss`` (loss_utils.py:4-32) and
``disprcnn.utils.stereo_utils.EndPointErrorLoss`` (stereo_utils.py:184-208).  Reductions and gradients run in
libdisprcnn_hip.so (drc_psm_loss_sums / drc_psm_loss_grad); note the two classes take their arguments in
different orders, as in the reference."""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib
from .. import engine as E

_WEIGHTS = (0.5, 0.7, 1.0)


def _sums(preds, target, mask):
    E.require_gpu(target, "PSMLoss target")
    dev = target.device
    sums = torch.empty(5, dtype=torch.float32, device=dev)           # overwritten: per-block partials added in block order (no atomics)
    ps = [p.contiguous() for p in preds] + [None] * (3 - len(preds))
    for p in preds:
        E.require_gpu(p, "PSMLoss prediction")
        if p.shape != target.shape:
            raise ValueError(f"prediction {tuple(p.shape)} vs target {tuple(target.shape)}")
    st = _lib.lib().drc_psm_loss_sums(E._ptr(ps[0]), E._ptr(ps[1]), E._ptr(ps[2]), E._ptr(target), E._ptr(mask), target.numel(), E._ptr(sums),
                                      E._ptr(E.scratch(dev, "loss", _lib.LOSS_SCRATCH_FLOATS)), E._stream_ptr(dev))
    _lib.check(st, "drc_psm_loss_sums")
    return sums


class _PSMTrainLoss(Function):
    @staticmethod
    def forward(ctx, target, mask, p1, p2, p3):
        target = target.contiguous().float()
        mask = (mask != 0).to(torch.uint8).contiguous()
        sums = _sums((p1, p2, p3), target, mask)
        ctx.save_for_backward(target, mask, p1, p2, p3, sums)
        denom = torch.where(sums[3] != 0, sums[3], torch.ones_like(sums[3]))      # "if mask.sum() != 0: divide"
        return (0.5 * sums[0] + 0.7 * sums[1] + sums[2]) / denom

    @staticmethod
    def backward(ctx, g):
        target, mask, p1, p2, p3, sums = ctx.saved_tensors
        g = g.contiguous().float().reshape(1)
        grads = []
        for p, w in zip((p1, p2, p3), _WEIGHTS):
            gp = torch.empty_like(p, memory_format=torch.contiguous_format)
            pc = p.contiguous()
            st = _lib.lib().drc_psm_loss_grad(E._ptr(pc), E._ptr(target), E._ptr(mask), p.numel(), E._ptr(sums), w, E._ptr(g),
                                              E._ptr(gp), E._stream_ptr(p.device))
            _lib.check(st, "drc_psm_loss_grad")
            grads.append(gp)
        return (None, None) + tuple(grads)


def _loss(output, target, mask):
    if isinstance(output, (list, tuple)) and len(output) == 3:
        return _PSMTrainLoss.apply(target, mask, *output)
    target = target.contiguous().float()
    mask8 = (mask != 0).to(torch.uint8).contiguous()
    sums = _sums((output,), target, mask8)
    return torch.where(sums[3] != 0, sums[4] / torch.where(sums[3] != 0, sums[3], torch.ones_like(sums[3])), torch.zeros_like(sums[4]))


class PSMLoss(nn.Module):
    def forward(self, output, y):
        return _loss(output, y["disparity"], y["mask"])


class EndPointErrorLoss(nn.Module):
    def forward(self, disp_target, disp_pred, mask=None):
        if mask is None:          # reference stereo_utils.py:185-187: no mask = every pixel counts
            mask = torch.ones_like(disp_target, dtype=torch.uint8)
        return _loss(disp_pred, disp_target, mask)
