"""AnchorGenerator -- drop-in for ``disprcnn.modeling.rpn.anchor_generator.AnchorGenerator`` (anchor_generator.py:35-150).

Two anchor tables live in the reference and both are kept: ``cell_anchors`` (classic Faster R-CNN cell anchors, :222-298) are only
state_dict buffers (``anchor_generator.cell_anchors.N``); the anchors the forward pass actually uses come from the pyramid
enumeration (:300-357): per level one size, every ratio, boxes of exact size (size/sqrt(r), size*sqrt(r)) centred on the cell
ORIGIN, position-major / ratio-minor.  They depend on the feature shapes only, so they are built once per shape on the host (numpy
float64 like the reference) and kept on the device."""
import numpy as np
import torch
from torch import nn

from ...structures.bounding_box import BoxList


def _cell_anchors(stride, sizes, ratios):
    w = h = float(stride)
    cx = cy = 0.5 * (stride - 1)
    rows = []
    for r in ratios:
        ws = np.round(np.sqrt(w * h / r))
        hs = np.round(ws * r)
        for s in np.asarray(sizes, dtype=np.float64) / stride:
            rows.append([cx - 0.5 * (ws * s - 1), cy - 0.5 * (hs * s - 1), cx + 0.5 * (ws * s - 1), cy + 0.5 * (hs * s - 1)])
    return torch.from_numpy(np.asarray(rows, dtype=np.float64)).float()


def pyramid_level_anchors(size, ratios, shape, stride):
    """float64 [H*W*A, 4] of one level (reference generate_anchors_single_pyramid)."""
    r = np.asarray(ratios, dtype=np.float64)
    hs, ws = size / np.sqrt(r), size * np.sqrt(r)
    h, w = shape
    cx = np.repeat(np.tile(np.arange(w) * stride, h), len(r)).astype(np.float64)
    cy = np.repeat(np.repeat(np.arange(h) * stride, w), len(r)).astype(np.float64)
    bw, bh = np.tile(ws, h * w), np.tile(hs, h * w)
    return np.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], axis=1)


class _BufferList(nn.Module):
    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(4, 8, 16, 32, 64), straddle_thresh=0):
        super().__init__()
        if len(anchor_strides) != len(sizes):
            raise RuntimeError("FPN should have #anchor_strides == #sizes")
        self.sizes, self.ratios, self.feature_strides = tuple(sizes), tuple(aspect_ratios), tuple(anchor_strides)
        self.straddle_thresh = straddle_thresh
        self.cell_anchors = _BufferList([_cell_anchors(s, (z,), self.ratios) for s, z in zip(self.feature_strides, self.sizes)])
        self._cache = {}

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def level_anchors(self, feature_shapes, device):
        """Per level float32 [H*W*A, 4] on `device` (cached per pyramid shape)."""
        key = (tuple(tuple(s) for s in feature_shapes), device)
        a = self._cache.get(key)
        if a is None:
            if len(self._cache) > 8:
                self._cache.clear()
            a = self._cache[key] = [torch.from_numpy(pyramid_level_anchors(z, self.ratios, shp, st)).float().to(device)
                                    for z, shp, st in zip(self.sizes, feature_shapes, self.feature_strides)]
        return a

    def forward(self, image_list, feature_maps):
        """-> per image a list (one BoxList of anchors per level, field 'visibility'), as the reference returns them."""
        dev = feature_maps[0].device
        per_level = self.level_anchors([tuple(f.shape[-2:]) for f in feature_maps], dev)
        out = []
        for (ih, iw) in image_list.image_sizes:
            lv = []
            for a in per_level:
                b = BoxList(a, (iw, ih), mode="xyxy")
                if self.straddle_thresh >= 0:
                    t = self.straddle_thresh
                    vis = (a[:, 0] >= -t) & (a[:, 1] >= -t) & (a[:, 2] < iw + t) & (a[:, 3] < ih + t)
                else:
                    vis = torch.ones(a.shape[0], dtype=torch.bool, device=dev)
                b.add_field("visibility", vis)
                lv.append(b)
            out.append(lv)
        return out


def make_anchor_generator(cfg):
    r = cfg.MODEL.RPN
    if not r.USE_FPN:
        raise NotImplementedError("only the FPN anchor generator of the shipped configs is built")
    return AnchorGenerator(r.ANCHOR_SIZES, r.ASPECT_RATIOS, r.ANCHOR_STRIDE, r.STRADDLE_THRESH)
