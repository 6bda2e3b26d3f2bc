"""StereoCombinedROIHeads -- drop-in for ``disprcnn.modeling.roi_heads.roi_heads`` (roi_heads.py:66-153), inference: the stereo box
head on both pyramids, then the mask head on the LEFT pyramid at the left detections."""
from torch import nn

from .box_head import build_roi_box_head
from .mask_head import build_roi_mask_head


class StereoCombinedROIHeads(nn.ModuleDict):
    def __init__(self, cfg, heads):
        super().__init__(heads)
        self.cfg = cfg

    def forward(self, left_features, right_features, left_proposals, right_proposals, left_targets=None, right_targets=None):
        x, left_det, right_det, losses = self.box({"left": left_features, "right": right_features},
                                                  {"left": left_proposals, "right": right_proposals},
                                                  {"left": left_targets, "right": right_targets})
        if self.cfg.MODEL.MASK_ON:
            x, left_det, loss_mask = self.mask(left_features, left_det, left_targets)
            losses.update(loss_mask)
        return x, left_det, right_det, losses


def build_roi_heads(cfg, in_channels):
    if not cfg.MODEL.STEREO_ON:
        raise NotImplementedError("only the stereo heads (MODEL.STEREO_ON) of the shipped configs are built")
    heads = [("box", build_roi_box_head(cfg, in_channels))]
    if cfg.MODEL.MASK_ON:
        heads.append(("mask", build_roi_mask_head(cfg, in_channels)))
    return StereoCombinedROIHeads(cfg, heads)
