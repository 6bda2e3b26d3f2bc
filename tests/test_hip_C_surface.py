"""`disprcnn._C` on the GPU (VERDICT r5 missing #3): the reference's layers call `_C.nms`, `_C.roi_align_forward`, `_C.roi_align_backward`
(layers/nms.py:3-8, layers/roi_align.py:3-46; pybind surface csrc/vision.cpp:7-15).  Driven here the way the reference's `_ROIAlign`
autograd function drives them, against the goldens of the reference's own compiled CPU kernels (tests/golden/roi_golden.npz,
nms_golden.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

from disprcnn_amd.utils import synth
from oracle import nms_oracle as N
from oracle import roi_oracle as R

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_C_roi_align_forward_bit_exact_vs_reference_kernel(dev):
    from disprcnn import _C
    z = np.load(os.path.join(GOLDEN, "roi_golden.npz"), allow_pickle=False)
    img = synth.hash_uniform("roi:img", (2, 3, 37, 53), 0.0, 1.0).to(dev)
    rois = torch.from_numpy(z["small_rois"]).to(dev)
    for k, (ph, pw, sr, scale) in enumerate(z["small_settings"]):
        got = _C.roi_align_forward(img, rois, float(scale), int(ph), int(pw), int(sr)).cpu().numpy()
        assert np.array_equal(got, z[f"small_out{k}"]), k


def test_C_roi_align_backward_as_the_reference_autograd_function_calls_it(dev):
    """The reference's `_ROIAlign.backward` (layers/roi_align.py:29-46): grad_input = _C.roi_align_backward(grad, rois, scale, ph, pw, bs, ch,
    h, w, sampling_ratio).  Checked as the adjoint of the forward: <forward(x), g> == <x, backward(g)> (the op is linear in x), and against
    this package's own autograd path."""
    from disprcnn import _C
    from disprcnn.layers import roi_align
    x = synth.hash_uniform("C:x", (2, 4, 23, 31), -1.0, 1.0).to(dev)
    rois = torch.tensor([[0, 2.5, 3.0, 20.0, 18.5], [1, 0.0, 0.0, 30.0, 22.0], [1, 7.0, 5.0, 12.0, 9.0]], device=dev)
    g = synth.hash_uniform("C:g", (3, 4, 5, 6), -1.0, 1.0).to(dev)
    y = _C.roi_align_forward(x, rois, 0.5, 5, 6, 2)
    gx = _C.roi_align_backward(g, rois, 0.5, 5, 6, 2, 4, 23, 31, 2)
    assert tuple(gx.shape) == (2, 4, 23, 31)
    lhs, rhs = (y.double() * g.double()).sum().item(), (x.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (lhs, rhs)
    xr = x.clone().requires_grad_(True)
    roi_align(xr, rois, (5, 6), 0.5, 2).backward(g)
    assert torch.allclose(xr.grad, gx, rtol=1e-5, atol=1e-6)          # (the scatter is float atomics: the order of the adds is not fixed)


def test_C_nms_vs_reference_kernel_golden_and_oracle(dev):
    from disprcnn import _C
    g = torch.Generator().manual_seed(3)
    n = 700
    xy = torch.rand(n, 2, generator=g) * 300
    wh = torch.rand(n, 2, generator=g) * 60 + 2
    boxes = torch.cat((xy, xy + wh), 1)
    scores = torch.rand(n, generator=g)
    keep = _C.nms(boxes.to(dev), scores.to(dev), 0.5).cpu()
    want = N.nms(boxes.numpy(), scores.numpy(), 0.5, strict=True)              # the CUDA op's test (IoU > thresh), csrc/cuda/nms.cu:23-131
    assert keep.dtype == torch.int64 and np.array_equal(keep.numpy(), want)
    from tests.golden.make_golden_nms import CASES, proposals
    G = np.load(os.path.join(GOLDEN, "nms_golden.npz"), allow_pickle=False)
    tag, n_, thr = CASES[0]
    dets, sc = proposals(tag, n_)
    k = _C.nms(dets.to(dev), sc.to(dev), thr).cpu().numpy()
    assert np.array_equal(k, N.nms(dets.numpy(), sc.numpy(), thr, strict=True))
    # (the fixture was recorded from the reference's CPU op, which suppresses at IoU >= thresh: equal wherever no pair sits exactly AT it)
    if np.array_equal(N.nms(dets.numpy(), sc.numpy(), thr, strict=False), N.nms(dets.numpy(), sc.numpy(), thr, strict=True)):
        assert np.array_equal(k, G[f"{tag}_keep"])
    assert _C.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5).numel() == 0
