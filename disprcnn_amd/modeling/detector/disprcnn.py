"""DispRCNN -- the 2D stage of the reference: drop-in for ``disprcnn.modeling.detector.disprcnn.DispRCNN`` (disprcnn.py:15-82 on top
of generalized_rcnn.py:15-64) in its shipped stereo configuration (configs/kitti/*/mask.yaml), inference.

  left + right image batch -> ONE backbone pass over the 2N images (ResNet-FPN on the HIP engine) -> Stereo RPN -> stereo box head
  -> mask head on the left view -> {'left': [BoxList], 'right': [BoxList]} with fields scores / labels / mask.

state_dict layout as in the reference: ``backbone.*``, ``rpn.*``, ``roi_heads.box.*``, ``roi_heads.mask.*`` (tests/test_detector2d.py
checks the key sets of the heads against the reference's).  DISPNET_ON (the FPN-feature disparity head of the early fork, dispmodule.py)
and DET3D_ON are separate models downstream and raise; training the 2D stage is not built (the shipped 3D configs freeze it:
FIX_BACKBONE / FIX_RPN / FIX_BOX_HEAD)."""
from types import SimpleNamespace

import torch
from torch import nn

from ...structures.image_list import to_image_list
from ..backbone import build_backbone
from ..roi_heads import build_roi_heads
from ..rpn import build_stereorpn


def default_cfg_2d(conv_body="R-101-FPN", num_classes=2, post_nms_top_n_test=300):
    """The yacs keys the 2D stage reads: config/defaults.py values overlaid with configs/kitti/car/vob/mask.yaml."""
    scales = (0.25, 0.125, 0.0625, 0.03125)
    return SimpleNamespace(MODEL=SimpleNamespace(
        META_ARCHITECTURE="DispRCNN", DEVICE="cuda", STEREO_ON=True, MASK_ON=True, DISPNET_ON=False, DET3D_ON=False, RPN_ONLY=False,
        CLS_AGNOSTIC_BBOX_REG=False,
        BACKBONE=SimpleNamespace(CONV_BODY=conv_body),
        RESNETS=SimpleNamespace(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256),
        RPN=SimpleNamespace(USE_FPN=True, ANCHOR_SIZES=(32, 64, 128, 256, 512), ANCHOR_STRIDE=(4, 8, 16, 32, 64), ASPECT_RATIOS=(0.5, 1.0, 2.0),
                            STRADDLE_THRESH=0, PRE_NMS_TOP_N_TEST=6000, POST_NMS_TOP_N_TEST=post_nms_top_n_test, NMS_THRESH=0.7, MIN_SIZE=0,
                            FPN_POST_NMS_TOP_N_TEST=2000),
        ROI_HEADS=SimpleNamespace(USE_FPN=True, BBOX_REG_WEIGHTS=(10.0, 10.0, 5.0, 5.0), SCORE_THRESH=0.05, NMS=0.5, DETECTIONS_PER_IMG=100),
        ROI_BOX_HEAD=SimpleNamespace(POOLER_RESOLUTION=7, POOLER_SCALES=scales, POOLER_SAMPLING_RATIO=0,
                                     FEATURE_EXTRACTOR="StereoFPN2MLPFeatureExtractor", PREDICTOR="StereoFPNPredictor", NUM_CLASSES=num_classes,
                                     MLP_HEAD_DIM=2048),
        ROI_MASK_HEAD=SimpleNamespace(POOLER_SCALES=scales, FEATURE_EXTRACTOR="MaskRCNNFPNFeatureExtractor", PREDICTOR="MaskRCNNC4Predictor",
                                      POOLER_RESOLUTION=14, POOLER_SAMPLING_RATIO=2, RESOLUTION=28, SHARE_BOX_FEATURE_EXTRACTOR=False,
                                      CONV_LAYERS=(256, 256, 256, 256), USE_GN=False, DILATION=1)))


class DispRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        m = cfg.MODEL
        if getattr(m, "DISPNET_ON", False) or getattr(m, "DET3D_ON", False):
            raise NotImplementedError("DISPNET_ON / DET3D_ON attach other models to the 2D stage; the instance-disparity stage is DispRCNN3D")
        self.backbone = build_backbone(cfg)
        self.rpn = build_stereorpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)

    def forward(self, lrimages, lrtargets=None):
        if self.training:
            raise NotImplementedError("training the 2D stage is not built (the shipped disparity / 3D configs freeze it)")
        left_images, right_images = to_image_list(lrimages["left"]), to_image_list(lrimages["right"])
        # the split-f16 layers of the trunk and of the RPN head report to ONE range guard, read once at the end of the stage; an overflow
        # (|v| > 65504: the fp32 reference has no such limit) repeats the stage on the fp32 kernels (engine.guarded)
        from ... import engine as E
        if left_images.tensors.is_cuda:
            if getattr(self, "_guard", None) is None:
                self._guard = E.OverflowGuard(left_images.tensors.device)
            return E.guarded(self._guard, lambda: self._forward_once(left_images, right_images), what="DispRCNN 2D stage (split-f16 3x3 layers)",
                             enabled=bool(getattr(self, "overflow_check", True)))
        return self._forward_once(left_images, right_images)

    def _forward_once(self, left_images, right_images):
        n = left_images.tensors.shape[0]
        feats = self.backbone(torch.cat((left_images.tensors, right_images.tensors), dim=0))
        left_features, right_features = [f[:n] for f in feats], [f[n:] for f in feats]
        rt = getattr(self.backbone, "_rt", None)
        # the blocked pyramid behind THESE dense maps (None if another forward of the backbone has overwritten the workspace since)
        levels = rt.blocked_levels(feats) if rt is not None and hasattr(rt, "blocked_levels") else None
        left_prop, right_prop, _ = self.rpn(left_images, right_images, left_features, right_features, blocked_levels=levels)
        _, left_result, right_result, _ = self.roi_heads(left_features, right_features, left_prop, right_prop)
        return {"left": left_result, "right": right_result}
