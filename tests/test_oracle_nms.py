"""Pins oracle/nms_oracle.py to the reference's own nms_cpu (tests/golden/nms_golden.npz; live binary when oracle/_ref is built)."""
import os

import numpy as np
import pytest
import torch

from oracle import nms_oracle as N
from tests.golden.make_golden_nms import CASES, proposals

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_golden.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag,n,thr", CASES)
def test_oracle_matches_reference_kernel(tag, n, thr):
    dets, scores = proposals(tag, n)
    assert np.array_equal(N.nms(dets.numpy(), scores.numpy(), thr), G[f"{tag}_keep"])


def test_oracle_edge_cases_and_live_binary():
    assert N.nms(np.zeros((0, 4)), np.zeros(0), 0.5).shape == (0,)
    # identical boxes: IoU = 1 >= t suppresses all but the best; threshold 1.0: >= suppresses duplicates, the CUDA op's > does not
    d = np.array([[10, 10, 50, 50]] * 3, dtype=np.float32)
    s = np.array([0.3, 0.9, 0.5], dtype=np.float32)
    assert N.nms(d, s, 0.5).tolist() == [1] and N.nms(d, s, 1.0).tolist() == [1] and N.nms(d, s, 1.0, strict=True).tolist() == [0, 1, 2]
    from oracle import build_ref
    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    g = torch.Generator().manual_seed(3)
    for n, thr in ((300, 0.6), (77, 0.2)):
        xy = torch.rand(n, 2, generator=g) * 200
        wh = torch.rand(n, 2, generator=g) * 80 + 1
        dets, scores = torch.cat([xy, xy + wh], 1), torch.rand(n, generator=g)
        assert np.array_equal(N.nms(dets.numpy(), scores.numpy(), thr), ref.nms(dets, scores, thr).numpy())
