"""f2 post-processing on the GPU (through the C ABI) against the reference's recorded maps and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from disprcnn_amd.utils import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "post_golden.npz"))
CASES = sorted({k.split(":")[0] for k in G.files})
H, W, S = 96, 320, 224


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _boxlists(name, dev):
    from disprcnn_amd.structures import BoxList
    lb, rb = torch.from_numpy(G[f"{name}:left"]).reshape(-1, 4), torch.from_numpy(G[f"{name}:right"]).reshape(-1, 4)
    disp = synth.hash_uniform(f"post:{name}", (len(lb), S, S), -48.0, 48.0)
    left, right = BoxList(lb.to(dev), (W, H)), BoxList(rb.to(dev), (W, H))
    left.add_field("disparity", disp.to(dev))
    return left, right


def test_disparity_map_processor_vs_reference_goldens(dev):
    """All five recorded images in ONE call (one launch): overlapping ROIs, a right box wider than the left one (crop), a
    single ROI whose negative values survive, 1-pixel ROIs, an image without ROIs.  fp32 resampling: 1e-4 px."""
    from disprcnn_amd.modeling.psmnet.inference import DisparityMapProcessor
    pairs = [_boxlists(n, dev) for n in CASES]
    maps = DisparityMapProcessor()([p[0] for p in pairs], [p[1] for p in pairs])
    assert len(maps) == len(CASES)
    for name, m in zip(CASES, maps):
        ref = torch.from_numpy(G[f"{name}:map"])
        assert tuple(m.data.shape) == (H, W)
        err = (m.data.cpu() - ref).abs().max().item()
        assert err <= 1e-4, f"{name}: max err {err:.3e}"
    one = DisparityMapProcessor()(*_boxlists("overlap3", dev))            # single BoxList in, single DisparityMap out
    assert (one.data.cpu() - torch.from_numpy(G["overlap3:map"])).abs().max().item() <= 1e-4


def _random_case(seed, n_img, max_rois, h, w):
    g = torch.Generator().manual_seed(seed)
    lbs, rbs, counts = [], [], []
    for _ in range(n_img):
        r = int(torch.randint(0, max_rois + 1, (1,), generator=g))
        x1 = torch.rand(r, generator=g) * (w - 40); y1 = torch.rand(r, generator=g) * (h - 30)
        bw = 2 + torch.rand(r, generator=g) * 200; bh = 2 + torch.rand(r, generator=g) * 150
        x2 = torch.minimum(x1 + bw, torch.tensor(float(w))); y2 = torch.minimum(y1 + bh, torch.tensor(float(h)))
        shift = torch.rand(r, generator=g) * 60 - 10
        x1p = (x1 - shift).clamp(min=0); x2p = (x2 - shift * (0.5 + torch.rand(r, generator=g))).clamp(min=1, max=float(w))
        x2p = torch.maximum(x2p, x1p + 1)
        lbs.append(torch.stack([x1, y1, x2, y2], 1)); rbs.append(torch.stack([x1p, y1, x2p, y2], 1)); counts.append(r)
    R = sum(counts)
    disp = (torch.rand(R, 56, 56, generator=g) * 96 - 48)
    return lbs, rbs, counts, disp


@pytest.mark.parametrize("seed,clamp0,with_masks", [(1, False, False), (2, True, True), (3, False, False)])
def test_disparity_paste_vs_oracle(dev, seed, clamp0, with_masks):
    from disprcnn_amd import ops
    from oracle import post_oracle as O
    h, w = 187, 413
    lbs, rbs, counts, disp = _random_case(seed, 4, 9, h, w)
    R = sum(counts)
    masks = (torch.rand(R, h, w, generator=torch.Generator().manual_seed(seed)) > 0.3).float() if with_masks else None
    boxes6 = ops.integer_roi_boxes(torch.cat(lbs).to(dev), torch.cat(rbs).to(dev))
    got = ops.disparity_paste(disp.to(dev), boxes6, counts, h, w, clamp0, masks.to(dev) if with_masks else None).cpu()
    o = 0
    for b, c in enumerate(counts):
        ref = O.disparity_map(lbs[b], rbs[b], disp[o:o + c], h, w, clamp0, masks[o:o + c] if with_masks else None)
        err = (got[b] - ref).abs().max().item()
        assert err <= 1e-4, f"image {b} ({c} rois): max err {err:.3e}"
        o += c


def test_roi_depth_maps_vs_oracle(dev):
    from disprcnn_amd import ops
    from oracle import post_oracle as O
    h, w = 120, 300
    lbs, rbs, counts, disp = _random_case(5, 1, 7, h, w)
    lb, rb = lbs[0], rbs[0]
    if len(lb) == 0:
        pytest.skip("empty draw")
    got = ops.roi_depth_maps(disp.to(dev), ops.integer_roi_boxes(lb.to(dev), rb.to(dev)), 389.34, h, w).cpu()
    ref = O.roi_depth_maps(lb, rb, disp, h, w, 389.34)
    # fuxb / (d + 1e-6) amplifies the resampling's rounding near d = 0: compare where the disparity is away from zero
    far = ref.abs() < 389.34 / 0.05
    assert torch.allclose(got[far], ref[far], rtol=2e-4, atol=1e-4)
    assert (got == 0).eq(ref == 0).all()


def test_paste_without_rois_is_zero(dev):
    from disprcnn_amd import ops
    out = ops.disparity_paste(torch.zeros(0, 8, 8, device=dev), torch.zeros(0, 6, dtype=torch.int32, device=dev), [0, 0], 17, 33)
    assert tuple(out.shape) == (2, 17, 33) and out.abs().max().item() == 0
