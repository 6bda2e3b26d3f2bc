"""CPU tests of the 2D stage's host logic: state_dict layout of DispRCNN's heads vs the reference's, anchor tables,
level mapping, BoxCoder.encode, BoxList helpers.  No HIP compute here (the GPU parity tests are tests/test_hip_detector2d.py)."""
import numpy as np
import pytest
import torch

from oracle import det_oracle as D
from disprcnn_amd.modeling.box_coder import BoxCoder
from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
from disprcnn_amd.modeling.poolers import LevelMapper
from disprcnn_amd.modeling.rpn.anchor_generator import AnchorGenerator
from disprcnn_amd.structures.bounding_box import BoxList
from disprcnn_amd.structures.boxlist_ops import cat_boxlist, intersect_sorted
from disprcnn_amd.utils import synth
from tests.helpers import golden_npz
from tests.test_oracle_det import RATIOS, SIZES, STRIDES, det_templates


def test_state_dict_layout_matches_reference_heads():
    m = DispRCNN(default_cfg_2d("R-50-FPN"))
    sd = m.state_dict()
    t_rpn, t_heads = det_templates()                      # key -> shape as printed by the reference's StereoRPN / StereoCombinedROIHeads
    want = {"rpn." + k: tuple(v.shape) for k, v in t_rpn.items()}
    want.update({"roi_heads." + k: tuple(v.shape) for k, v in t_heads.items()})
    got = {k: tuple(v.shape) for k, v in sd.items() if not k.startswith("backbone.")}
    assert got == want
    m.load_state_dict({**{k: v for k, v in sd.items() if k.startswith("backbone.")},
                       **{"rpn." + k: v for k, v in synth.synth_det_state(t_rpn).items()},
                       **{"roi_heads." + k: v for k, v in synth.synth_det_state(t_heads).items()}}, strict=True)


def test_anchor_tables_vs_reference():
    z = golden_npz("det_golden.npz")
    g = AnchorGenerator(SIZES, RATIOS, STRIDES)
    assert g.num_anchors_per_location() == [3] * 5
    for lvl, ca in enumerate(g.cell_anchors):
        np.testing.assert_allclose(ca.numpy(), z[f"cell_anchors_{lvl}"], rtol=0, atol=1e-4)
    shapes = [(40, 80), (20, 40), (10, 20), (5, 10), (3, 5)]
    got = g.level_anchors(shapes, torch.device("cpu"))
    ref = D.pyramid_anchors(SIZES, RATIOS, shapes, STRIDES)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a.numpy(), b.astype(np.float32), rtol=0, atol=0)
    lists = g(type("IL", (), {"image_sizes": [(160, 320), (150, 300)]})(), [torch.zeros(2, 1, *s) for s in shapes])
    assert len(lists) == 2 and len(lists[0]) == 5 and lists[1][0].size == (300, 150)
    vis = lists[0][0].get_field("visibility")
    a = got[0]
    assert torch.equal(vis, (a[:, 0] >= 0) & (a[:, 1] >= 0) & (a[:, 2] < 320) & (a[:, 3] < 160))


def test_level_mapper_and_boxcoder_encode_vs_oracle():
    boxes = synth.hash_uniform("lm", (200, 4), 0.0, 1.0)
    x1, y1 = boxes[:, 0] * 900, boxes[:, 1] * 300
    b = torch.stack([x1, y1, x1 + 4 + boxes[:, 2] ** 3 * 900, y1 + 4 + boxes[:, 3] ** 3 * 500], 1)
    lv = LevelMapper(2.0, 5.0)([BoxList(b[:120], (1242, 375)), BoxList(b[120:], (1242, 375))])
    assert torch.equal(lv, D.map_levels(b, 2.0, 5.0))
    assert set(lv.tolist()) == {0, 1, 2, 3}
    # decode(encode(gt)) gives gt with x2 / y2 one pixel larger: the reference's legacy +1 widths are not undone by its decode
    coder = BoxCoder((10.0, 10.0, 5.0, 5.0))
    prop = b[:50]
    j = synth.hash_uniform("enc", (50, 4), 0.0, 6.0)
    gt = torch.stack([prop[:, 0] + j[:, 0] - 3, prop[:, 1] + j[:, 1] - 3, prop[:, 2] + j[:, 0] + j[:, 2], prop[:, 3] + j[:, 1] + j[:, 3]], 1)
    one4, one6 = torch.tensor([0.0, 0.0, 1.0, 1.0]), torch.tensor([0.0, 0.0, 1.0, 1.0, 0.0, 1.0])
    np.testing.assert_allclose(D.decode(coder.encode(gt, prop), prop, coder.weights).numpy(), (gt + one4).numpy(), rtol=0, atol=2e-3)
    gt6 = torch.cat([gt, gt[:, [0, 2]] - 7.0], 1)
    np.testing.assert_allclose(D.decode(coder.encode(gt6, prop), prop, coder.weights).numpy(), (gt6 + one6).numpy(), rtol=0, atol=2e-3)
    with pytest.raises(ValueError):
        coder.encode(torch.zeros(3, 5), torch.zeros(3, 4))


def test_boxlist_helpers():
    b = BoxList(torch.tensor([[-5.0, 2.0, 30.0, 400.0], [10.0, 10.0, 10.0, 10.0], [1.0, 2.0, 5.0, 9.0]]), (100, 50))
    b.add_field("scores", torch.tensor([0.1, 0.2, 0.3]))
    assert torch.equal(b.area(), torch.tensor([36.0 * 399.0, 1.0, 40.0]))
    c = b.clip_to_image(remove_empty=True)
    assert len(c) == 2 and torch.equal(c.bbox[0], torch.tensor([0.0, 2.0, 30.0, 49.0])) and torch.equal(c.get_field("scores"), torch.tensor([0.1, 0.3]))
    assert torch.equal(c.xywh()[1], torch.tensor([1.0, 2.0, 5.0, 8.0]))
    both = cat_boxlist([c, c])
    assert len(both) == 4 and both.copy_with_fields("scores").get_field("scores").shape == (4,)
    with pytest.raises(ValueError):
        cat_boxlist([c, BoxList(torch.zeros(1, 4), (9, 9))])
    assert intersect_sorted(torch.tensor([0, 2, 5, 9]), torch.tensor([1, 2, 9, 11])).tolist() == [2, 9]


def test_training_raises_loudly():
    m = DispRCNN(default_cfg_2d("R-50-FPN")).train()
    with pytest.raises(NotImplementedError):
        m({"left": torch.zeros(1, 3, 64, 64), "right": torch.zeros(1, 3, 64, 64)})
