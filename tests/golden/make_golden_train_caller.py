#!/usr/bin/env python3
"""Golden fixtures for DispRCNN3D's TRAINING entry (BASELINE configs[2]), recorded from the IMPORTED REFERENCE
(authoring container only).

    python oracle/build_ref.py && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train_caller.py

Reference code exercised (its own Python on torch-CPU):
  DispRCNN3D.forward / _forward_train / remove_illegal_detections / remove_low_score_rois /
  prepare_psmnet_input_and_target / crop_and_transform_roi_img     disprcnn/modeling/detector/disprcnn3d.py:44-112,192-264,286-308
  Masker / paste_mask_in_image                                     disprcnn/modeling/roi_heads/mask_head/inference.py:90-190
  DisparityMap.crop / .resize                                      disprcnn/structures/disparity.py:38-77
  ROIAlign -> disprcnn._C.roi_align_forward = the reference's own CPU kernel, built by oracle/build_ref.py
  PSMNet (train mode) + EndPointErrorLoss                          modeling/psmnet/stackhourglass.py, utils/stereo_utils.py:184-208
Harness-only stand-ins (never part of the numbers): cv2 / pycocotools / PointRCNN CUDA ops / yacs (imports of neighbouring
modules), `Tensor.cuda` = identity (the reference hard-codes .cuda(), disprcnn3d.py:106-111), and a 3-line holder object for
the ground-truth masks (the reference's SegmentationMask rasterises polygons with pycocotools; only its
get_full_image_mask_tensor() result enters this path).
Inputs come from disprcnn_amd.utils.synth and are rebuilt by the tests from the stored boxes.
"""
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


yacs, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
yc.CfgNode = CfgNode
yacs.config = yc
sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yc
torch._six = types.SimpleNamespace(PY3=True, PY37=True, string_classes=(str,), int_classes=(int,),
                                   container_abcs=__import__("collections").abc)
sys.modules["torch._six"] = torch._six

from oracle import build_ref  # noqa: E402

_ref_c = build_ref.load() or (build_ref.build() and build_ref.load())
assert _ref_c is not None, "python oracle/build_ref.py first"
sys.modules["disprcnn._C"] = _ref_c
torch.Tensor.cuda = lambda self, *a, **k: self

from disprcnn.modeling.detector.disprcnn3d import DispRCNN3D  # noqa: E402  (the reference)
from disprcnn.structures.bounding_box import BoxList  # noqa: E402
from disprcnn.structures.disparity import DisparityMap  # noqa: E402
from disprcnn.structures.image_list import ImageList  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402
from tests.helpers import state_for  # noqa: E402

torch.set_num_threads(8)
W, H, RES = 640, 300, 224


class GTMasks:
    """Holder for the rasterised ground-truth instance masks of one image ([G,H,W] uint8)."""

    def __init__(self, m):
        self.m = m

    def get_full_image_mask_tensor(self):          # the ONE method of SegmentationMask this path calls (segmentation_mask.py:537-542)
        return self.m.sum(dim=0).clamp(max=1)


def cfg_for(min_score, max_roi):
    NS = types.SimpleNamespace
    return NS(MODEL=NS(DISPNET_ON=True, DET3D_ON=False,
                       DISPNET=NS(MAX_DISP=48, MIN_DISP=-48, SINGLE_MODAL_WEIGHTED_AVERAGE=False, RESOLUTIONS=(RES,), TRAINED_MODEL="",
                                  ROI_MIN_SCORE=min_score, MAX_ROI_FOR_TRAINING=max_roi),
                       POINTRCNN=NS(TRAINED_MODEL="")),
              SOLVER=NS(TRAIN_PSM=True, TRAIN_PC=False))


def scene():
    """Three images: 3, 0 and 4 detections (incl. one illegal box and two low-score ones); instance masks and a disparity map per
    image.  Everything closed-form (synth) or literal."""
    base = synth.hash_uniform("tc:L", (3, 3, H // 6, W // 8), 0.0, 1.0)
    limg = torch.nn.functional.interpolate(base, (H, W), mode="bilinear", align_corners=True)
    rimg = torch.roll(limg, -7, 3)
    lboxes = [torch.tensor([[100.3, 20.6, 135.8, 110.2], [300.0, 10.0, 420.0, 260.0], [500.7, 130.2, 515.9, 171.4]]),
              torch.zeros(0, 4),
              torch.tensor([[10.0, 40.0, 70.0, 180.0], [600.4, 50.0, 639.9, 299.9], [200.0, 100.0, 290.0, 200.0], [330.5, 60.5, 352.0, 118.0],
                            [50.0, 50.0, 50.5, 50.9]])]                  # the last one is illegal (w,h <= 1)
    shifts = [[6.2, 14.7, 2.1], [], [30.0, 9.5, 21.3, 4.0, 1.0]]
    scores = [torch.tensor([0.9, 0.8, 0.3]), torch.zeros(0), torch.tensor([0.95, 0.6, 0.2, 0.7, 0.99])]
    rboxes = []
    for lb, sh in zip(lboxes, shifts):
        rb = lb.clone()
        if len(sh):
            rb[:, [0, 2]] -= torch.tensor(sh)[:, None]
            rb[:, 2] += torch.tensor([1.5, -3.0, 0.0, 2.0, 0.0][: len(sh)])
        rboxes.append(rb)
    masks28 = [synth.hash_uniform(f"tc:m{i}", (len(lb), 1, 28, 28), 0.0, 1.0) ** 0.5 for i, lb in enumerate(lboxes)]     # mask head probabilities
    gt_masks, disp_maps = [], []
    for i, lb in enumerate(lboxes):
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        g = []
        for b in lb.tolist():                           # an ellipse inside every detection box
            cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, max((b[2] - b[0]) * 0.45, 0.6), max((b[3] - b[1]) * 0.45, 0.6)
            g.append((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).to(torch.uint8))
        gt_masks.append(torch.stack(g) if g else torch.zeros(0, H, W, dtype=torch.uint8))
        disp_maps.append(synth.hash_uniform(f"tc:d{i}", (H // 4, W // 4), 2.0, 40.0).repeat_interleave(4, 0).repeat_interleave(4, 1))
    return limg, rimg, lboxes, rboxes, scores, masks28, gt_masks, disp_maps


def build_inputs(sc):
    limg, rimg, lboxes, rboxes, scores, masks28, gt_masks, disp_maps = sc
    lres, rres, ltg = [], [], []
    for i in range(3):
        l, r = BoxList(lboxes[i], (W, H)), BoxList(rboxes[i], (W, H))
        l.add_field("scores", scores[i]); l.add_field("mask", masks28[i])
        r.add_field("scores", scores[i])
        t = BoxList(lboxes[i], (W, H))
        t.add_field("masks", GTMasks(gt_masks[i]))
        t.add_map("disparity", DisparityMap(disp_maps[i]))
        lres.append(l); rres.append(r); ltg.append(t)
    sizes = [(H, W)] * 3
    return {"left": ImageList(limg, sizes), "right": ImageList(rimg, sizes)}, {"left": lres, "right": rres}, {"left": ltg, "right": ltg}


def sample(out, tag, t, k=2048):
    flat = t.reshape(-1).float()
    u = synth.hash_uniform(f"sample:{tag}:{flat.numel()}", (k,), 0.0, 1.0).double()
    idx = (u * flat.numel()).long().clamp(max=flat.numel() - 1).numpy()
    out[tag + "_idx"], out[tag + "_val"] = idx, flat[idx].numpy()
    out[tag + "_abssum"] = np.array(flat.double().abs().sum().item())


def main():
    out = {}
    sc = scene()
    for tag, min_score, max_roi in (("all", 0.05, 12), ("trunc", 0.5, 3)):
        model = DispRCNN3D(cfg_for(min_score, max_roi))
        model.dispnet.load_state_dict(state_for("B"), strict=True)
        model.train()
        images, results, targets = build_inputs(sc)
        # stage 1: the target preparation alone (what the engine's device kernels replace)
        lr, rr = model.remove_illegal_detections(results["left"], results["right"])
        lr, rr = model.remove_low_score_rois(lr, rr)
        out[f"{tag}_kept"] = np.array([len(a) for a in lr])
        li, ri, tg, mk = model.prepare_psmnet_input_and_target(images["left"], images["right"], lr, rr, targets["left"])
        out[f"{tag}_targets"] = tg.numpy().astype(np.float32)
        out[f"{tag}_masks"] = np.packbits(mk.numpy().astype(np.uint8), axis=None)
        out[f"{tag}_masks_shape"] = np.array(mk.shape)
        sample(out, f"{tag}_left", li); sample(out, f"{tag}_right", ri)
        # stage 2: the whole training forward -> loss (train-mode PSMNet: batch-statistic BatchNorm over the ROI batch)
        images, results, targets = build_inputs(sc)
        torch.manual_seed(0)
        with torch.no_grad():
            losses = model(images, results, targets)
        out[f"{tag}_loss"] = np.array(float(losses["disp_loss"]))
        print(tag, "kept", out[f"{tag}_kept"], "rois", tg.shape[0], "mask px", int(mk.sum()), "loss", out[f"{tag}_loss"])
    sc_np = {"lboxes": sc[2], "rboxes": sc[3], "scores": sc[4]}
    for k, v in sc_np.items():
        for i, t in enumerate(v):
            out[f"scene_{k}{i}"] = t.numpy()
    path = os.path.join(HERE, "train_caller_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
