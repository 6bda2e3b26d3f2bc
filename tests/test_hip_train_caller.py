"""GPU parity of DispRCNN3D's TRAINING entry (BASELINE configs[2]) against outputs recorded from the imported
reference (tests/golden/train_caller_golden.npz: its own DispRCNN3D._forward_train with Masker, DisparityMap and the compiled
reference ROIAlign): ROI selection, crops, disparity targets, masks, the loss of a train-mode forward, ROI truncation."""
import os
import tempfile

import numpy as np
import pytest
import torch

from disprcnn_amd.utils import synth
from tests.helpers import state_for
from tests.test_oracle_train_caller import G, H, RES, W, golden_masks, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _inputs(sc, dev):
    from disprcnn_amd.structures import BoxList, DisparityMap, ImageList
    limg, rimg, lboxes, rboxes, scores, masks28, gt_masks, disp_maps = sc
    lres, rres, ltg = [], [], []
    for i in range(3):
        l, r = BoxList(lboxes[i], (W, H)), BoxList(rboxes[i], (W, H))
        l.add_field("scores", scores[i]); l.add_field("mask", masks28[i]); r.add_field("scores", scores[i])
        t = BoxList(lboxes[i], (W, H))
        t.add_field("masks", gt_masks[i])                       # instance stack [G,H,W]: the union is taken by the caller
        t.add_map("disparity", DisparityMap(disp_maps[i]))
        lres.append(l); rres.append(r); ltg.append(t)
    sizes = [(H, W)] * 3
    return ({"left": ImageList(limg.to(dev), sizes), "right": ImageList(rimg.to(dev), sizes)}, {"left": lres, "right": rres},
            {"left": ltg, "right": ltg})


def _model(dev, min_score, max_roi):
    from disprcnn_amd.modeling.detector import build_detection_model
    from disprcnn_amd.modeling.detector.disprcnn3d import default_cfg
    cfg = default_cfg(48, -48, RES)
    cfg.MODEL.DISPNET.ROI_MIN_SCORE, cfg.MODEL.DISPNET.MAX_ROI_FOR_TRAINING = min_score, max_roi
    m = build_detection_model(cfg)
    m.dispnet.load_state_dict(state_for("B"), strict=True)
    return m.to(dev).train()


@pytest.mark.parametrize("tag,min_score,max_roi", [("all", 0.05, 12), ("trunc", 0.5, 3)])
def test_training_targets_and_loss_vs_reference(dev, tag, min_score, max_roi):
    sc = scene()
    model = _model(dev, min_score, max_roi)
    images, results, targets = _inputs(sc, dev)
    lr, rr = model.remove_illegal_detections(results["left"], results["right"])
    lr, rr = model.remove_low_score_rois(lr, rr)
    assert [len(a) for a in lr] == G[f"{tag}_kept"].tolist()
    left, right, tg, mk = model.prepare_psmnet_input_and_target(images["left"], images["right"], lr, rr, targets["left"])
    ref_t = G[f"{tag}_targets"]
    err = np.abs(tg.cpu().numpy() - ref_t)
    assert (err <= 1e-5 * np.abs(ref_t) + 1e-4).all(), err.max()                  # fp32 bilinear, values up to ~600 (x res / width)
    ref_m = golden_masks(tag)
    mism = (mk.cpu().numpy() != ref_m).mean()
    print(f"{tag}: target max err {err.max():.2e}, mask pixels {int(ref_m.sum())}, mismatching fraction {mism:.2e}")
    # .byte() truncates the interpolated mask: a pixel survives only if the fp32 blend of four 1s is exactly 1.0, which depends on the
    # last-ulp behaviour of the host's vectorised F.interpolate -- a handful of pixels may differ
    assert mism <= 2e-3, mism
    for side, crop in (("left", left), ("right", right)):
        flat = crop.cpu().reshape(-1).numpy()
        assert np.abs(flat[G[f"{tag}_{side}_idx"]] - G[f"{tag}_{side}_val"]).max() < 2e-5
    # the whole training forward: loss of the three heads (train-mode BatchNorm over the ROI batch), then a backward pass
    images, results, targets = _inputs(sc, dev)
    losses = model(images, results, targets)
    loss = losses["disp_loss"]
    ref_loss = float(G[f"{tag}_loss"])
    print(f"{tag}: loss {loss.item():.4f} vs reference {ref_loss:.4f}")
    assert abs(loss.item() - ref_loss) <= 3e-3 * abs(ref_loss), (loss.item(), ref_loss)
    kept_after = [len(a) for a in results["left"]]                                   # the caller's lists are not mutated
    assert kept_after == [3, 0, 5]
    loss.backward()
    grads = [p.grad for p in model.dispnet.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0


def test_training_without_rois_gives_zero_loss_reaching_every_parameter(dev):
    """An image batch whose detections are all removed (or absent) must still yield a loss every rank can call backward() on
    (data-parallel ranks never skip a step)."""
    from disprcnn_amd.structures import BoxList, DisparityMap, ImageList
    model = _model(dev, 0.05, 12)
    img = synth.hash_uniform("nr", (1, 3, H, W), 0.0, 1.0).to(dev)
    l = BoxList(torch.zeros(0, 4), (W, H)); l.add_field("scores", torch.zeros(0)); l.add_field("mask", torch.zeros(0, 1, 28, 28))
    r = BoxList(torch.zeros(0, 4), (W, H)); r.add_field("scores", torch.zeros(0))
    t = BoxList(torch.zeros(0, 4), (W, H)); t.add_field("masks", torch.zeros(0, H, W, dtype=torch.uint8)); t.add_map("disparity", DisparityMap(torch.zeros(H, W)))
    losses = model({"left": ImageList(img, [(H, W)]), "right": ImageList(img, [(H, W)])}, {"left": [l], "right": [r]}, {"left": [t], "right": [t]})
    assert losses["disp_loss"].item() == 0.0
    losses["disp_loss"].backward()
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in model.dispnet.parameters() if p.requires_grad)
    assert tuple(l.get_field("disparity").shape) == (0, RES, RES) if l.has_field("disparity") else True


def test_load_state_dict_reloads_trained_dispnet(dev):
    """Reference disprcnn3d.py:310-316: loading a detector checkpoint re-loads the disparity net from DISPNET.TRAINED_MODEL."""
    from disprcnn_amd.modeling.detector.disprcnn3d import DispRCNN3D, default_cfg
    sd_b = state_for("B")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "bestmodel.pth")
        torch.save({"model": sd_b}, path)
        cfg = default_cfg(48, -48, RES)
        m = DispRCNN3D(cfg)
        other = {k: (v + 1 if v.is_floating_point() else v) for k, v in m.state_dict().items()}
        cfg.MODEL.DISPNET.TRAINED_MODEL = path
        m.load_state_dict(other)
        got = m.dispnet.state_dict()
        assert all(torch.equal(got[k], sd_b[k]) for k in sd_b)
