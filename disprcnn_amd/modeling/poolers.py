"""Pooler -- drop-in for ``disprcnn.modeling.poolers.Pooler`` (poolers.py:10-149): multi-level ROIAlign over the FPN.

Fork behaviour kept: the level of a box is round(4 + ln(sqrt(area) / 224)) (natural log, no epsilon), clamped to the configured
levels; the extra top level (P6) is never pooled from; the ROIAlign scale of a level is feature height / image height, not the
configured constant.  ROIAlign itself is the bit-exact HIP kernel behind layers.ROIAlign (drc_roi_align_fwd)."""
import ctypes as C
import math

import torch
from torch import nn

from .. import _lib
from .. import engine as E
from ..layers.roi_align import ROIAlign


class LevelMapper:
    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([b.area() for b in boxlists]))
        lv = torch.round(self.lvl0 + torch.log(s / self.s0))
        return torch.clamp(lv, min=self.k_min, max=self.k_max).to(torch.int64) - int(self.k_min)


class Pooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio):
        super().__init__()
        self.poolers = nn.ModuleList([ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio) for s in scales])
        self.output_size = tuple(output_size)
        self.map_levels = LevelMapper(-math.log2(scales[0]), -math.log2(scales[-1]))
        self.single_launch = True        # all levels in one drc_roi_align_fpn_fwd launch (False: the reference's per-level loop)

    @staticmethod
    def convert_to_roi_format(boxes):
        dev = boxes[0].bbox.device
        ids = torch.cat([torch.full((len(b), 1), float(i), dtype=torch.float32, device=dev) for i, b in enumerate(boxes)], 0)
        return torch.cat([ids, torch.cat([b.bbox for b in boxes], 0)], 1)

    def forward(self, x, boxes):
        """x: per-level [N,C,H,W] maps (a trailing extra level is ignored); boxes: list[BoxList] -> [R,C,oh,ow] in box order."""
        rois = self.convert_to_roi_format(boxes)
        c = x[0].shape[1]
        if len(self.poolers) == 1:
            return self.poolers[0](x[0], rois)
        if len(rois) == 0:
            return torch.zeros(0, c, *self.output_size, dtype=torch.float32, device=x[0].device)
        levels = self.map_levels(boxes)
        feats = [f.contiguous() for f in x[: len(self.poolers)]]
        needs_grad = torch.is_grad_enabled() and any(f.requires_grad for f in feats)
        if not needs_grad and self.single_launch and all(f.is_cuda and f.dtype == torch.float32 and f.shape[1] == c for f in feats) and len(feats) <= 8:
            # one launch over the pyramid (drc_roi_align_fpn_fwd): every roi reads its level's map and scale; no per-level host sync
            pyr = _lib.DrcFpnPyramid()
            pyr.n_levels = len(feats)
            for i, f in enumerate(feats):
                pyr.feat[i], pyr.H[i], pyr.W[i] = f.data_ptr(), f.shape[2], f.shape[3]
                pyr.scale[i] = f.shape[2] / boxes[0].height                     # (fork quirk: feature height / image height)
            lv32 = levels.to(torch.int32).contiguous()
            rois = rois.contiguous()
            out = torch.empty(len(rois), c, *self.output_size, dtype=torch.float32, device=rois.device)
            st = _lib.lib().drc_roi_align_fpn_fwd(C.byref(pyr), E._ptr(rois), E._ptr(lv32), E._ptr(out), len(rois), c, self.output_size[0],
                                                  self.output_size[1], int(self.poolers[0].sampling_ratio), E._stream_ptr(rois.device))
            _lib.check(st, "drc_roi_align_fpn_fwd")
            return out
        out = torch.zeros(len(rois), c, *self.output_size, dtype=torch.float32, device=x[0].device)       # (the per-level path only)
        for level, (feat, pooler) in enumerate(zip(x[: len(self.poolers)], self.poolers)):
            idx = torch.nonzero(levels == level).reshape(-1)
            if idx.numel() == 0:
                continue
            out.index_copy_(0, idx, pooler(feat, rois.index_select(0, idx), feat.shape[2] / boxes[0].height))
        return out


def make_pooler(cfg, head_name):
    h = getattr(cfg.MODEL, head_name)
    return Pooler((h.POOLER_RESOLUTION, h.POOLER_RESOLUTION), h.POOLER_SCALES, h.POOLER_SAMPLING_RATIO)
