"""Pins oracle/train_caller_oracle.py to the reference's recorded training targets (tests/golden/train_caller_golden.npz)."""
import os

import numpy as np
import torch

from oracle import train_caller_oracle as T
from disprcnn_amd.utils import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_caller_golden.npz"), allow_pickle=False)
W, H, RES = 640, 300, 224


def scene():
    """Rebuilds the inputs of tests/golden/make_golden_train_caller.py (boxes / scores are stored, the rest is closed form)."""
    lboxes = [torch.from_numpy(G[f"scene_lboxes{i}"]).reshape(-1, 4) for i in range(3)]
    rboxes = [torch.from_numpy(G[f"scene_rboxes{i}"]).reshape(-1, 4) for i in range(3)]
    scores = [torch.from_numpy(G[f"scene_scores{i}"]).reshape(-1) for i in range(3)]
    base = synth.hash_uniform("tc:L", (3, 3, H // 6, W // 8), 0.0, 1.0)
    limg = torch.nn.functional.interpolate(base, (H, W), mode="bilinear", align_corners=True)
    rimg = torch.roll(limg, -7, 3)
    masks28 = [synth.hash_uniform(f"tc:m{i}", (len(lb), 1, 28, 28), 0.0, 1.0) ** 0.5 for i, lb in enumerate(lboxes)]
    gt_masks, disp_maps = [], []
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    for i, lb in enumerate(lboxes):
        g = []
        for b in lb.tolist():
            cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, max((b[2] - b[0]) * 0.45, 0.6), max((b[3] - b[1]) * 0.45, 0.6)
            g.append((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).to(torch.uint8))
        gt_masks.append(torch.stack(g) if g else torch.zeros(0, H, W, dtype=torch.uint8))
        disp_maps.append(synth.hash_uniform(f"tc:d{i}", (H // 4, W // 4), 2.0, 40.0).repeat_interleave(4, 0).repeat_interleave(4, 1))
    return limg, rimg, lboxes, rboxes, scores, masks28, gt_masks, disp_maps


def golden_masks(tag):
    shape = tuple(G[f"{tag}_masks_shape"])
    return np.unpackbits(G[f"{tag}_masks"])[: int(np.prod(shape))].reshape(shape)


def selected(sc, min_score):
    """(image, roi index) pairs after remove_illegal_detections + remove_low_score_rois, per image."""
    _, _, lboxes, rboxes, scores, *_ = sc
    legal = [T.legal_keep(lb, rb) for lb, rb in zip(lboxes, rboxes)]
    keep = T.low_score_keep([s[k] for s, k in zip(scores, legal)], min_score)
    return [torch.nonzero(k)[kk].reshape(-1).tolist() for k, kk in zip(legal, keep)]


def test_roi_selection_and_targets_match_reference():
    sc = scene()
    _, _, lboxes, rboxes, scores, masks28, gt_masks, disp_maps = sc
    for tag, min_score in (("all", 0.05), ("trunc", 0.5)):
        sel = selected(sc, min_score)
        assert [len(s) for s in sel] == G[f"{tag}_kept"].tolist()
        tg, mk = [], []
        for i, idx in enumerate(sel):
            gfull = gt_masks[i].sum(dim=0).clamp(max=1).to(torch.uint8) if len(gt_masks[i]) else torch.zeros(H, W, dtype=torch.uint8)
            for r in idx:
                _, _, t, m = T.roi_targets(lboxes[i][r].tolist(), rboxes[i][r].tolist(), masks28[i][r, 0], gfull, disp_maps[i], RES)
                tg.append(t); mk.append(m)
        tg, mk = torch.stack(tg).numpy(), torch.stack(mk).numpy()
        assert np.array_equal(tg, G[f"{tag}_targets"]), np.abs(tg - G[f"{tag}_targets"]).max()
        assert np.array_equal(mk, golden_masks(tag))


def test_truncation_counts():
    assert T.truncate_counts([2, 0, 3], 3) == [2, 0, 1] and T.truncate_counts([3, 0, 4], 12) == [3, 0, 4] and T.truncate_counts([5, 2], 4) == [4, 0]
