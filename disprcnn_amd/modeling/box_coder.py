"""BoxCoder -- drop-in for ``disprcnn.modeling.box_coder.BoxCoder`` (box_coder.py:6-279).

``decode`` runs in libdisprcnn_hip.so (drc_box_decode_fwd): 4 codes per class (x1y1x2y2) or 6 (x1y1x2y2 + x1', x2' of the right view,
decoded against 4-coordinate reference boxes -- the form the Stereo RPN and the stereo box head use).  ``encode`` (training targets of
the 2D stage, which is not trained here) is plain tensor arithmetic."""
import math

import torch

from .. import _lib
from .. import engine as E


class BoxCoder:
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = tuple(float(w) for w in weights)
        self.bbox_xform_clip = float(bbox_xform_clip)
        self._w = {}

    def decode(self, rel_codes, boxes, clip_to=None, per=None):
        """rel_codes [R, K*4] or [R, K*6], boxes [R,4] xyxy -> same shape; clip_to = (width, height) fuses BoxList.clip_to_image.
        per = codes per box (4 or 6); by default the reference's rule: 6 when the row length is a multiple of 6, else 4."""
        E.require_gpu(rel_codes, "BoxCoder.decode")
        if boxes.shape[1] != 4:
            raise ValueError("BoxCoder.decode: 6-coordinate reference boxes (decode_..._fromboxes6) are not used on this path")
        per = per or (6 if rel_codes.shape[1] % 6 == 0 else 4)
        if per not in (4, 6) or rel_codes.shape[1] % per:
            raise ValueError("wrong shape.")
        dev = rel_codes.device
        codes = rel_codes.contiguous().float()
        boxes = boxes.to(dev).contiguous().float()
        if boxes.shape[0] != codes.shape[0]:
            raise ValueError("BoxCoder.decode: one reference box per row of codes")
        w = self._w.get(dev)
        if w is None:
            w = self._w[dev] = torch.tensor(self.weights, dtype=torch.float32, device=dev)
        out = torch.empty_like(codes)
        cw, ch = (float(clip_to[0]), float(clip_to[1])) if clip_to is not None else (0.0, 0.0)
        st = _lib.lib().drc_box_decode_fwd(E._ptr(codes), E._ptr(boxes), E._ptr(out), codes.shape[0], codes.shape[1] // per, per, E._ptr(w),
                                           self.bbox_xform_clip, cw, ch, E._stream_ptr(dev))
        _lib.check(st, "drc_box_decode_fwd")
        return out

    def encode(self, reference_boxes, proposals):
        """Targets (dx, dy, dw, dh[, dx', dw']) of reference boxes [R,4|6] w.r.t. proposals [R,4] (box_coder.py:21-50,83-116)."""
        wx, wy, ww, wh = self.weights
        ew, eh = proposals[:, 2] - proposals[:, 0] + 1, proposals[:, 3] - proposals[:, 1] + 1
        ex, ey = proposals[:, 0] + 0.5 * ew, proposals[:, 1] + 0.5 * eh
        gw, gh = reference_boxes[:, 2] - reference_boxes[:, 0] + 1, reference_boxes[:, 3] - reference_boxes[:, 1] + 1
        gx, gy = reference_boxes[:, 0] + 0.5 * gw, reference_boxes[:, 1] + 0.5 * gh
        cols = [wx * (gx - ex) / ew, wy * (gy - ey) / eh, ww * torch.log(gw / ew), wh * torch.log(gh / eh)]
        if reference_boxes.shape[1] == 6:
            if proposals.shape[1] != 4:
                raise ValueError("encode: 6-coordinate proposals are not used on this path")
            gwp = reference_boxes[:, 5] - reference_boxes[:, 4] + 1
            gxp = reference_boxes[:, 4] + 0.5 * gwp
            cols += [wx * (gxp - ex) / ew, ww * torch.log(gwp / ew)]
        elif reference_boxes.shape[1] != 4:
            raise ValueError("size(1) is not 4 or 6.")
        return torch.stack(cols, dim=1)
