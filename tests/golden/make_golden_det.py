#!/usr/bin/env python3
"""Golden fixtures for the 2D detection stage's heads, recorded from the IMPORTED REFERENCE (authoring container only).

    python oracle/build_ref.py && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_det.py

Reference code exercised (its own Python on torch-CPU, eval mode):
  StereoRPN / SRPNHead / SRPNPostProcessor / AnchorGenerator      disprcnn/modeling/rpn/stereo_rpn/{srpn,inference}.py, rpn/anchor_generator.py
  BoxCoder.decode, double_view_boxlist_nms                        modeling/box_coder.py, structures/boxlist_ops.py
  StereoCombinedROIHeads: ROIBoxHead (StereoFPN2MLPFeatureExtractor, StereoFPNPredictor, PostProcessor),
  ROIMaskHead (MaskRCNNFPNFeatureExtractor, MaskRCNNC4Predictor, MaskPostProcessor), Pooler / LevelMapper
                                                                  modeling/roi_heads/**, modeling/poolers.py
  ROIAlign / nms = disprcnn._C.{roi_align_forward,nms}: the reference's own CPU kernels, built by oracle/build_ref.py
Harness-only stand-ins (never part of the numbers): cv2 / pycocotools / PointRCNN CUDA ops / yacs (imports of neighbouring modules),
numpy's removed `np.float` alias, `Tensor.cuda` and `BoxList.to` = identity (anchor_generator.py hard-codes .to(device='cuda')).
Inputs (pyramids, weights) come from disprcnn_amd.utils.synth and are rebuilt by the tests.
"""
import copy
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

for name in ("cv2", "pycocotools", "pycocotools.mask", "pointnet2_cuda", "iou3d_cuda", "roipool3d_cuda", "tensorboardX", "termcolor",
             "numba", "zarr", "fastai", "matplotlib", "matplotlib.pyplot", "dl_ext", "dl_ext.primitive", "dl_ext.vision_ext",
             "dl_ext.vision_ext.datasets", "dl_ext.vision_ext.datasets.kitti", "dl_ext.vision_ext.datasets.kitti.structures"):
    sys.modules.setdefault(name, MagicMock())


class CfgNode(dict):
    def __init__(self, init=None, *a, **k):
        super().__init__(init or {})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)


yacs, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
yc.CfgNode = CfgNode
yacs.config = yc
sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yc
torch._six = types.SimpleNamespace(PY3=True, PY37=True, string_classes=(str,), int_classes=(int,),
                                   container_abcs=__import__("collections").abc)
sys.modules["torch._six"] = torch._six
np.float, np.int, np.bool = float, int, bool

from oracle import build_ref  # noqa: E402

_ref_c = build_ref.load() or (build_ref.build() and build_ref.load())
assert _ref_c is not None, "python oracle/build_ref.py first"
sys.modules["disprcnn._C"] = _ref_c
torch.Tensor.cuda = lambda self, *a, **k: self

from disprcnn.config import cfg as _cfg  # noqa: E402  (the reference)
from disprcnn.modeling.roi_heads.roi_heads import build_roi_heads  # noqa: E402
from disprcnn.modeling.rpn.stereo_rpn.srpn import StereoRPN  # noqa: E402
from disprcnn.structures.bounding_box import BoxList  # noqa: E402
from disprcnn.structures.image_list import ImageList  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402

BoxList.to = lambda self, *a, **k: self
torch.set_num_threads(16)

# the shipped 2D config (configs/kitti/car/vob/mask.yaml) on top of config/defaults.py; POST_NMS_TOP_N_TEST lowered from 300 so that
# the CPU run and the fixture stay small
POST_NMS = 80
CASES = [("a", 2, 160, 320), ("b", 1, 200, 264)]


def make_cfg():
    cfg = _cfg.clone()
    m = cfg.MODEL
    m.STEREO_ON, m.MASK_ON = True, True
    m.RPN.USE_FPN, m.RPN.ANCHOR_STRIDE = True, (4, 8, 16, 32, 64)
    m.RPN.PRE_NMS_TOP_N_TEST, m.RPN.POST_NMS_TOP_N_TEST = 6000, POST_NMS
    m.ROI_HEADS.USE_FPN = True
    b = m.ROI_BOX_HEAD
    b.POOLER_RESOLUTION, b.POOLER_SCALES, b.POOLER_SAMPLING_RATIO = 7, (0.25, 0.125, 0.0625, 0.03125), 0
    b.FEATURE_EXTRACTOR, b.PREDICTOR, b.NUM_CLASSES, b.MLP_HEAD_DIM = "StereoFPN2MLPFeatureExtractor", "StereoFPNPredictor", 2, 2048
    k = m.ROI_MASK_HEAD
    k.POOLER_SCALES, k.FEATURE_EXTRACTOR, k.PREDICTOR = (0.25, 0.125, 0.0625, 0.03125), "MaskRCNNFPNFeatureExtractor", "MaskRCNNC4Predictor"
    k.POOLER_RESOLUTION, k.POOLER_SAMPLING_RATIO, k.RESOLUTION, k.SHARE_BOX_FEATURE_EXTRACTOR = 14, 2, 28, False
    return cfg


def samples(out, key, t, n=4096):
    flat = t.detach().reshape(-1).double()
    idx = (synth.hash_uniform("det:idx:" + key, (min(n, flat.numel()),), 0.0, 1.0).double() * flat.numel()).long().clamp(max=flat.numel() - 1)
    out[key + "_idx"], out[key + "_val"] = idx.numpy(), flat[idx].numpy().astype(np.float32)
    out[key + "_abssum"] = np.array(flat.abs().sum().item())


def main():
    cfg = make_cfg()
    rpn = StereoRPN(cfg, 256).eval()
    heads = build_roi_heads(cfg, 256).eval()
    rpn.load_state_dict(synth.synth_det_state(rpn.state_dict(), gain=synth.DET_GAIN), strict=True)
    heads.load_state_dict(synth.synth_det_state(heads.state_dict(), gain=synth.DET_GAIN), strict=True)
    out = {}
    for lvl, ca in enumerate(rpn.anchor_generator.cell_anchors):
        out[f"cell_anchors_{lvl}"] = ca.numpy()
    with torch.no_grad():
        for tag, n, h, w in CASES:
            fl, fr = synth.synth_pyramid(n, h, w, tag="det" + tag)
            images = ImageList(torch.zeros(n, 3, h, w), [(h, w)] * n)
            objectness, regression = rpn.head(fl, fr)
            for lvl in range(5):
                samples(out, f"{tag}_obj{lvl}", objectness[lvl])
                samples(out, f"{tag}_reg{lvl}", regression[lvl])
            anchors = rpn.anchor_generator(images, fl)
            for lvl in range(5):
                samples(out, f"{tag}_anchors{lvl}", anchors[0][lvl].bbox)
            lp, rp, _ = rpn(images, images, fl, fr)
            for i in range(n):
                out[f"{tag}_prop_left{i}"], out[f"{tag}_prop_right{i}"] = lp[i].bbox.numpy(), rp[i].bbox.numpy()
                out[f"{tag}_prop_score{i}"] = lp[i].get_field("objectness").numpy()
                print(tag, i, "proposals", len(lp[i]), "score range", float(lp[i].get_field("objectness").min()), float(lp[i].get_field("objectness").max()))
            box = heads.box
            x = box.feature_extractor({"left": fl, "right": fr}, {"left": lp, "right": rp})
            logits, deltas = box.predictor(x)
            samples(out, f"{tag}_box_x", x)
            out[f"{tag}_box_logits"], out[f"{tag}_box_deltas"] = logits.numpy(), deltas.numpy()
            _, ld, rd, _ = heads(fl, fr, lp, rp)
            for i in range(n):
                out[f"{tag}_det_left{i}"], out[f"{tag}_det_right{i}"] = ld[i].bbox.numpy(), rd[i].bbox.numpy()
                out[f"{tag}_det_score{i}"], out[f"{tag}_det_label{i}"] = ld[i].get_field("scores").numpy(), ld[i].get_field("labels").numpy()
                m = ld[i].get_field("mask")
                out[f"{tag}_det_mask_shape{i}"] = np.array(m.shape)
                samples(out, f"{tag}_det_mask{i}", m, 8192)
                print(tag, i, "detections", len(ld[i]), "mask", tuple(m.shape), "labels", ld[i].get_field("labels").unique().tolist())
    path = os.path.join(HERE, "det_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
