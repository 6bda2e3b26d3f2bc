"""CPU tests that PIN oracle/det_oracle.py (Stereo RPN, stereo box head, mask head -- SURVEY f3/f4) to the fixtures
tests/golden/make_golden_det.py recorded from the imported reference modules.  Tolerances: conv / FC outputs 2e-4 abs
(fp32 summation order of F.conv2d on another machine), boxes 2e-3 px, scores 1e-5; kept sets must be identical."""
import numpy as np
import pytest
import torch

from oracle import det_oracle as D
from disprcnn_amd.utils import synth
from tests.helpers import check_samples, golden_npz

SIZES, RATIOS, STRIDES = (32, 64, 128, 256, 512), (0.5, 1.0, 2.0), (4, 8, 16, 32, 64)
CASES = [("a", 2, 160, 320), ("b", 1, 200, 264)]
POST_NMS = 80


def det_templates():
    """state-dict templates (key -> zero tensor of the right shape) of the reference's StereoRPN and StereoCombinedROIHeads."""
    t = {f"anchor_generator.cell_anchors.{i}": torch.from_numpy(D.cell_anchors(s, (z,), RATIOS)).float() for i, (s, z) in enumerate(zip(STRIDES, SIZES))}
    rpn = {"head.conv.weight": (512, 256, 3, 3), "head.conv.bias": (512,), "head.cls_logits.weight": (6, 1024, 1, 1), "head.cls_logits.bias": (6,),
           "head.bbox_pred.weight": (18, 1024, 1, 1), "head.bbox_pred.bias": (18,)}
    heads = {"box.feature_extractor.RCNN_top.0.weight": (2048, 512, 7, 7), "box.feature_extractor.RCNN_top.0.bias": (2048,),
             "box.feature_extractor.RCNN_top.3.weight": (2048, 2048, 1, 1), "box.feature_extractor.RCNN_top.3.bias": (2048,),
             "box.predictor.cls_score.weight": (2, 2048), "box.predictor.cls_score.bias": (2,),
             "box.predictor.bbox_pred.weight": (12, 2048), "box.predictor.bbox_pred.bias": (12,)}
    for i in (1, 2, 3, 4):
        heads[f"mask.feature_extractor.mask_fcn{i}.weight"], heads[f"mask.feature_extractor.mask_fcn{i}.bias"] = (256, 256, 3, 3), (256,)
    heads.update({"mask.predictor.conv5_mask.weight": (256, 256, 2, 2), "mask.predictor.conv5_mask.bias": (256,),
                  "mask.predictor.mask_fcn_logits.weight": (2, 256, 1, 1), "mask.predictor.mask_fcn_logits.bias": (2,)})
    t.update({k: torch.zeros(v) for k, v in rpn.items()})
    return t, {k: torch.zeros(v) for k, v in heads.items()}


def det_states():
    t_rpn, t_heads = det_templates()
    return synth.synth_det_state(t_rpn, gain=synth.DET_GAIN), synth.synth_det_state(t_heads, gain=synth.DET_GAIN)


@pytest.fixture(scope="module")
def z():
    return golden_npz("det_golden.npz")


def test_cell_anchors_match_reference_buffers(z):
    for lvl, (s, size) in enumerate(zip(STRIDES, SIZES)):
        np.testing.assert_allclose(D.cell_anchors(s, (size,), RATIOS), z[f"cell_anchors_{lvl}"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("tag,n,h,w", CASES)
def test_srpn_and_heads_vs_reference(z, tag, n, h, w):
    torch.set_num_threads(8)
    w_rpn, w_heads = det_states()
    fl, fr = synth.synth_pyramid(n, h, w, tag="det" + tag)
    obj, reg = D.srpn_head(fl, fr, w_rpn)
    for lvl in range(5):
        check_samples(z, tag, f"obj{lvl}", obj[lvl], 2e-5)
        check_samples(z, tag, f"reg{lvl}", reg[lvl], 2e-4)
    anchors = D.pyramid_anchors(SIZES, RATIOS, [tuple(f.shape[-2:]) for f in fl], STRIDES)
    for lvl in range(5):
        check_samples(z, tag, f"anchors{lvl}", torch.from_numpy(anchors[lvl]).float(), 1e-4)
    props = D.srpn_select(anchors, obj, reg, [(w, h)] * n, 6000, POST_NMS, 0.7, 0, strict=False)
    for i, (pl, pr, ps) in enumerate(props):
        assert pl.shape == z[f"{tag}_prop_left{i}"].shape
        np.testing.assert_allclose(ps.numpy(), z[f"{tag}_prop_score{i}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(pl.numpy(), z[f"{tag}_prop_left{i}"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(pr.numpy(), z[f"{tag}_prop_right{i}"], rtol=0, atol=2e-3)
    # heads on the REFERENCE's proposals (so that one flipped NMS tie upstream cannot cascade)
    pl = [torch.from_numpy(z[f"{tag}_prop_left{i}"]) for i in range(n)]
    pr = [torch.from_numpy(z[f"{tag}_prop_right{i}"]) for i in range(n)]
    x, logits, deltas = D.box_head(fl, fr, pl, pr, h, w_heads)
    check_samples(z, tag, "box_x", x, 3e-4)
    np.testing.assert_allclose(logits.numpy(), z[f"{tag}_box_logits"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(deltas.numpy(), z[f"{tag}_box_deltas"], rtol=0, atol=1e-3)
    dets = D.box_post(torch.from_numpy(z[f"{tag}_box_logits"]), torch.from_numpy(z[f"{tag}_box_deltas"]), pl, pr, [(w, h)] * n, strict=False)
    for i, d in enumerate(dets):
        assert d["left"].shape == z[f"{tag}_det_left{i}"].shape
        np.testing.assert_allclose(d["scores"].numpy(), z[f"{tag}_det_score{i}"], rtol=0, atol=1e-5)
        np.testing.assert_array_equal(d["labels"].numpy(), z[f"{tag}_det_label{i}"])
        np.testing.assert_allclose(d["left"].numpy(), z[f"{tag}_det_left{i}"], rtol=0, atol=2e-3)
        np.testing.assert_allclose(d["right"].numpy(), z[f"{tag}_det_right{i}"], rtol=0, atol=2e-3)
    boxes = [torch.from_numpy(z[f"{tag}_det_left{i}"]) for i in range(n)]
    labels = [torch.from_numpy(z[f"{tag}_det_label{i}"]) for i in range(n)]
    masks = D.mask_head(fl, boxes, labels, h, w_heads)
    off = 0
    for i in range(n):
        k = len(boxes[i])
        assert tuple(z[f"{tag}_det_mask_shape{i}"]) == (k, 1, 28, 28)
        check_samples(z, tag, f"det_mask{i}", masks[off:off + k], 2e-4)
        off += k
