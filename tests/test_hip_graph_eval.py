"""Eval batches replayed from a captured HIP graph (PSMNet.graph_eval, runtime._replay): one image's ROIs are ~45-150 launches of a few
microseconds each, so the eager step is bound by the host (reference: engine/inference.py:24-50 runs one image per step).  The graph must give
bit-identical results to the eager launches, and must be re-captured when what it points at is replaced (weights, BN folds, a workspace pool)."""
import pytest
import torch

from disprcnn_amd.utils import synth
from tests.helpers import state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _model(dev, case, mode, mx=48, mn=0):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(mx, mn)
    m.load_state_dict(state_for(case), strict=True)
    m.graph_eval = mode
    return m.to(dev).eval()


def test_graph_replay_equals_eager_from_features(dev):
    """One graph per unit-count BUCKET (the ROI count changes from image to image): a count is padded to its bucket and sliced, a
    bucket is captured the second time it is seen; ROI pairs are independent units, so the padded replay equals the eager pass bit for bit."""
    me, mg = _model(dev, "A", False), _model(dev, "A", "auto")
    seq = ((16, "g0", 0), (16, "g1", 1), (16, "g2", 1), (5, "g3", 1), (6, "g4", 2), (5, "g5", 2), (4, "g6", 2), (1, "g7", 2), (1, "g8", 3), (6, "g9", 3))
    with torch.no_grad():
        for n, tag, graphs in seq:          # eager first sighting, capture, replay; 5 and 6 share a bucket (5 replays the graph captured for 6)
            fl, fr = synth.synth_features(n, 32, 28, 28, tag=tag)
            fl, fr = fl.to(dev), fr.to(dev)
            want = me.forward_from_features(fl, fr, (112, 112))
            got = mg.forward_from_features(fl, fr, (112, 112))
            assert got.shape == want.shape and torch.equal(got, want), (n, tag)
            assert len(mg._rt._graphs) == graphs, (n, tag, len(mg._rt._graphs))
    assert len(mg._rt._graphs) == 3 and not me._rt._graphs
    # above GRAPH_MAX_UNITS "auto" stays eager; True always replays
    from disprcnn_amd.modeling.psmnet import runtime as R
    fl, fr = synth.synth_features(R.GRAPH_MAX_UNITS + 8, 32, 28, 28, tag="big")
    with torch.no_grad():
        a = mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
        assert len(mg._rt._graphs) == 3
        mg.graph_eval = True
        mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))           # (first sighting of the bucket under graph_eval = True: eager)
        b = mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
        assert len(mg._rt._graphs) == 4 and torch.equal(a, b)
    mg.graph_eval = "sometimes"
    with pytest.raises(ValueError):
        mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))


def test_graph_is_recaptured_when_weights_or_pools_change(dev):
    mg = _model(dev, "A", "auto")
    fl, fr = synth.synth_features(6, 32, 28, 28, tag="rc")
    fl, fr = fl.to(dev), fr.to(dev)
    with torch.no_grad():
        a = mg.forward_from_features(fl, fr, (112, 112))
        # new weights (tempered set): the packed weights the graph points at are rebuilt
        mg.load_state_dict(state_for("At"), strict=True)
        b = mg.forward_from_features(fl, fr, (112, 112))
        want_b = _model(dev, "At", False).forward_from_features(fl, fr, (112, 112))
        assert torch.equal(b, want_b) and not torch.equal(a, b)
        # a bigger batch replaces the workspace pool the 6-ROI graph points into
        fl2, fr2 = synth.synth_features(40, 32, 28, 28, tag="rc2")
        mg.forward_from_features(fl2.to(dev), fr2.to(dev), (112, 112))
        c = mg.forward_from_features(fl, fr, (112, 112))
        assert torch.equal(c, want_b)
        # an in-place parameter update (an optimizer step between eval passes)
        mg.dres0[0][0].weight.mul_(1.01)
        d = mg.forward_from_features(fl, fr, (112, 112))
        me = _model(dev, "At", False)
        me.dres0[0][0].weight.mul_(1.01)
        assert torch.equal(d, me.forward_from_features(fl, fr, (112, 112)))


def test_graph_replay_full_psmnet_and_train_mode_is_eager(dev):
    me, mg = _model(dev, "B", False, 48, -48), _model(dev, "B", "auto", 48, -48)
    left, right = synth.synth_images(3, 224, 224, tag="gB")
    with torch.no_grad():
        want = me((left.to(dev), right.to(dev)))
        got1 = mg((left.to(dev), right.to(dev)))
        got2 = mg({"left": left.to(dev), "right": right.to(dev)})
    assert torch.equal(got1, want) and torch.equal(got2, want) and len(mg._rt._graphs) == 1
    mg.train()
    out = mg((left.to(dev), right.to(dev)))                  # train mode (and autograd on): eager, differentiable
    assert isinstance(out, tuple) and len(out) == 3 and out[2].requires_grad and len(mg._rt._graphs) == 1
