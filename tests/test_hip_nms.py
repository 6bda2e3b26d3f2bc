"""GPU parity of drc_nms_sorted_fwd / layers.nms against the indices recorded from the reference's own nms_cpu and
against the oracle; integer output: exact."""
import os

import numpy as np
import pytest
import torch

from oracle import nms_oracle as N
from tests.golden.make_golden_nms import CASES, proposals

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_golden.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag,n,thr", CASES)
def test_nms_vs_reference_kernel_golden(dev, tag, n, thr):
    from disprcnn_amd.layers import nms
    dets, scores = proposals(tag, n)
    keep = nms(dets.to(dev), scores.to(dev), thr, strict=False).cpu().numpy()          # the CPU op's >= (what the fixture was recorded with)
    assert keep.dtype == np.int64 and np.array_equal(keep, G[f"{tag}_keep"])
    keep_s = nms(dets.to(dev), scores.to(dev), thr).cpu().numpy()                       # the CUDA op's > (default on the GPU)
    assert np.array_equal(keep_s, N.nms(dets.numpy(), scores.numpy(), thr, strict=True))


def test_nms_edge_cases(dev):
    from disprcnn_amd.layers import nms
    assert tuple(nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5).shape) == (0,)
    d = torch.tensor([[10., 10., 50., 50.]] * 3, device=dev)
    s = torch.tensor([0.3, 0.9, 0.5], device=dev)
    assert nms(d, s, 0.5).tolist() == [1] and nms(d, s, 1.0, strict=False).tolist() == [1] and nms(d, s, 1.0).tolist() == [0, 1, 2]
    # suppression chains across 64-box blocks: a ladder of shifted boxes, each overlapping only its neighbours
    n = 300
    x = torch.arange(n, dtype=torch.float32) * 6
    dets = torch.stack([x, torch.zeros(n), x + 20, torch.full((n,), 20.)], 1)
    scores = torch.linspace(1.0, 0.1, n)
    ref = N.nms(dets.numpy(), scores.numpy(), 0.4, strict=True)
    assert np.array_equal(nms(dets.to(dev), scores.to(dev), 0.4).cpu().numpy(), ref) and 0 < len(ref) < n
    with pytest.raises(RuntimeError):
        nms(torch.zeros(3, 4), torch.zeros(3), 0.5)                                   # CPU tensors: no fallback


def test_nms_more_than_32768_boxes(dev):
    """n > 32,768 (a KITTI pyramid's ~120 k anchors when PRE_NMS_TOP_N_TEST is off): the walk with its removed-words in LDS."""
    from disprcnn_amd.layers import nms
    n = 33_500
    g = torch.Generator().manual_seed(7)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([300.0, 90.0])
    wh = 8.0 + torch.rand(n, 2, generator=g) * torch.tensor([60.0, 40.0])
    dets = torch.cat((xy, xy + wh), 1)
    scores = torch.rand(n, generator=g)
    ref = N.nms(dets.numpy(), scores.numpy(), 0.5, strict=True)
    got = nms(dets.to(dev), scores.to(dev), 0.5).cpu().numpy()
    assert 10 < len(ref) < n // 4 and np.array_equal(got, ref)


def test_nms_pair_equals_two_calls():
    """nms_pair (one sort + one launch pair for the two views of a stereo list) == two nms calls on the same scores."""
    from disprcnn_amd.layers import nms, nms_pair
    from tests.golden.make_golden_nms import proposals
    dev = torch.device("cuda:0")
    for tag, n in (("pa", 700), ("pb", 65), ("pc", 1), ("pd", 0)):
        a, s = proposals(tag, n) if n else (torch.zeros(0, 4), torch.zeros(0))
        b = a.clone()
        if n:
            b[:, [0, 2]] -= 17.0 + 40.0 * s[:, None]
        ka, kb = nms_pair(a.to(dev), b.to(dev), s.to(dev), 0.7)
        assert torch.equal(ka, nms(a.to(dev), s.to(dev), 0.7)) and torch.equal(kb, nms(b.to(dev), s.to(dev), 0.7))
        # joint=True: the intersection double_view_boxlist_nms keeps (reference intersect_pytorch, boxlist_ops.py:36-46)
        from disprcnn_amd.structures.boxlist_ops import intersect_sorted
        kj = nms_pair(a.to(dev), b.to(dev), s.to(dev), 0.7, joint=True)
        assert torch.equal(kj, intersect_sorted(ka, kb))
        # score-sorted input: the joint walk with early exit == the first max_keep of the intersection, for any max_keep
        from disprcnn_amd.layers import nms_pair_sorted_joint
        o = torch.sort(s, descending=True, stable=True)[1]
        a_s, b_s, s_s = a[o].to(dev), b[o].to(dev), s[o].to(dev)
        full = nms_pair(a_s, b_s, s_s, 0.7, joint=True)
        for mk in (-1, 1, 7, 64, 65, 200, 100000):
            got = nms_pair_sorted_joint(a_s, b_s, 0.7, mk)
            assert torch.equal(got, full[:mk] if mk > 0 else full), (tag, mk)
