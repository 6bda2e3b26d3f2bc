"""The value holders' geometry against fixtures recorded from the reference's own classes (tests/golden/make_golden_structures.py:
disprcnn/structures/disparity.py:39-83, bounding_box.py:119-277).  BoxList and DisparityMap.crop / __sub__ are host logic (CPU);
DisparityMap.resize is a HIP kernel (gpu-marked, through the C ABI)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from disprcnn_amd.structures import BoxList, DisparityMap
from disprcnn_amd.structures.bounding_box import FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM
from disprcnn_amd.utils import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "structures_golden.npz"))
_spec = importlib.util.spec_from_file_location("make_golden_structures_params", os.path.join(os.path.dirname(__file__), "golden", "_structures_cases.py"))
_cases = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_cases)
RESIZE, CROPS, BOXES, SIZE = _cases.RESIZE, _cases.CROPS, _cases.BOXES, _cases.SIZE


def _eq(a, key):
    ref = torch.from_numpy(G[key])
    assert tuple(a.shape) == tuple(ref.shape), (key, tuple(a.shape), tuple(ref.shape))
    assert torch.equal(a.cpu(), ref), (key, (a.cpu() - ref).abs().max().item())


@pytest.mark.parametrize("name,hw,box", CROPS)
def test_disparity_map_crop_and_sub_vs_reference(name, hw, box):
    d = synth.hash_uniform(f"structures:crop:{name}", hw, -48.0, 48.0)
    _eq(DisparityMap(d).crop(box).data, f"crop:{name}")
    _eq((DisparityMap(d) - 3.25).data[::4, ::4], f"sub:{name}")
    assert (DisparityMap(d) - 3.25).data is not d and torch.equal(DisparityMap(d).data, d)      # the operand is left alone


def test_disparity_map_crop_refuses_negative_windows():
    with pytest.raises(ValueError):
        DisparityMap(torch.zeros(4, 4)).crop((-1, 0, 2, 2))


def test_disparity_map_resize_negative_size_is_a_copy():
    d = DisparityMap(torch.arange(12.0).reshape(3, 4))
    with pytest.warns(UserWarning):
        r = d.resize((-1, 5))
    assert torch.equal(r.data, d.data) and r.data is not d.data


@pytest.mark.parametrize("name,hw,dst", RESIZE)
def test_disparity_map_resize_cpu_vs_reference(name, hw, dst):
    """A CPU map (the dataloader side: the reference resamples its targets' CPU DisparityMaps in the transforms) against the
    reference-recorded outputs: bilinear to 1e-5 px, the signed max pooling exactly."""
    d = synth.hash_uniform(f"structures:resize:{name}", hw, -48.0, 48.0)
    bil = DisparityMap(d).resize(dst).data
    ref = torch.from_numpy(G[f"resize:{name}:bilinear"])
    assert tuple(bil.shape) == tuple(ref.shape)
    assert (bil - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    _eq(DisparityMap(d).resize(dst, use_max_pooling=True).data, f"resize:{name}:maxpool")


def test_boxlist_map_follows_resize_on_cpu():
    """BoxList.resize forwards to its maps (reference bounding_box.py:156-199) -- on the CPU tensors a dataloader worker holds."""
    d = synth.hash_uniform("structures:map", (SIZE[1], SIZE[0]), 0.0, 64.0)
    b = BoxList(torch.tensor(BOXES), SIZE)
    b.add_map("disparity", DisparityMap(d))
    got = b.resize((160, 48)).get_map("disparity").data
    assert (got - torch.from_numpy(G["map:resize"])).abs().max().item() <= 1e-5 * 64.0


def test_boxlist_convert_resize_transpose_crop_vs_reference():
    b = BoxList(torch.tensor(BOXES), SIZE)
    bw = b.convert("xywh")
    _eq(bw.bbox, "box:xywh")
    _eq(bw.convert("xyxy").bbox, "box:xywh_back")
    assert b.convert("xyxy") is b and bw.mode == "xywh"
    for tag, src in (("xyxy", b), ("xywh", bw)):
        _eq(src.resize((640, 192)).bbox, f"box:{tag}:resize_equal")
        _eq(src.resize((400, 300)).bbox, f"box:{tag}:resize_unequal")
        _eq(src.transpose(FLIP_LEFT_RIGHT).bbox, f"box:{tag}:flip_lr")
        _eq(src.transpose(FLIP_TOP_BOTTOM).bbox, f"box:{tag}:flip_tb")
        c = src.crop((40, 10, 250, 80))
        _eq(c.bbox, f"box:{tag}:crop")
        assert tuple(int(v) for v in c.size) == tuple(int(v) for v in G[f"box:{tag}:crop_size"]) and c.mode == tag
        assert src.resize((400, 300)).size == (400, 300) and src.resize((400, 300)).mode == tag
    with pytest.raises(NotImplementedError):
        b.transpose(2)
    with pytest.raises(ValueError):
        b.convert("cxcywh")


def test_boxlist_fields_and_maps_follow_the_boxes():
    """Tensor fields pass through untouched (why 'disparity' fields stay ROI-normalised); a non-tensor field / map that knows the
    operation follows; crop hands image-level maps over unless crop_map."""
    class Probe:
        def __init__(self, log): self.log = log
        def resize(self, size, *a, **k): return Probe(self.log + [("resize", tuple(size))])
        def transpose(self, m): return Probe(self.log + [("transpose", m)])
        def crop(self, box): return Probe(self.log + [("crop", tuple(box))])
    b = BoxList(torch.tensor(BOXES), SIZE)
    t = torch.arange(5.0)
    b.add_field("scores", t)
    b.add_field("probe", Probe([]))
    d = synth.hash_uniform("structures:map", (SIZE[1], SIZE[0]), 0.0, 64.0)
    b.add_map("disparity", DisparityMap(d))
    c = b.crop((40, 10, 250, 80), crop_map=True)
    assert c.get_field("scores") is t and c.get_field("probe").log == [("crop", (40, 10, 250, 80))]
    _eq(c.get_map("disparity").data, "map:crop")
    keep = b.crop((40, 10, 250, 80))
    assert tuple(keep.get_map("disparity").data.shape) == tuple(int(v) for v in G["map:crop_nomap_shape"])
    f = b.transpose(FLIP_LEFT_RIGHT)
    assert f.get_field("probe").log == [("transpose", 0)] and f.get_map("disparity") is b.get_map("disparity")


@pytest.mark.gpu
@pytest.mark.parametrize("name,hw,dst", RESIZE)
def test_disparity_map_resize_vs_reference(name, hw, dst):
    """drc_disparity_resize_fwd: bilinear(align_corners) to 1e-5 px of the reference's F.interpolate (fp32 blend order), the signed max
    pooling exactly."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    d = synth.hash_uniform(f"structures:resize:{name}", hw, -48.0, 48.0).to(dev)
    bil = DisparityMap(d).resize(dst).data.cpu()
    ref = torch.from_numpy(G[f"resize:{name}:bilinear"])
    assert tuple(bil.shape) == tuple(ref.shape)
    assert (bil - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), (bil - ref).abs().max().item()
    mp = DisparityMap(d).resize(dst, use_max_pooling=True).data.cpu()
    _eq(mp, f"resize:{name}:maxpool")


@pytest.mark.gpu
def test_boxlist_map_follows_resize_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    d = synth.hash_uniform("structures:map", (SIZE[1], SIZE[0]), 0.0, 64.0).to(dev)
    b = BoxList(torch.tensor(BOXES).to(dev), SIZE)
    b.add_map("disparity", DisparityMap(d))
    got = b.resize((160, 48)).get_map("disparity").data.cpu()
    ref = torch.from_numpy(G["map:resize"])
    assert (got - ref).abs().max().item() <= 1e-5 * 64.0
