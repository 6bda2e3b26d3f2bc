"""GPU tests of the train-mode forward (per-GPU batch-statistics BatchNorm, three heads) against the CPU oracle and the
reference's recorded train-mode outputs (tests/golden: Bt_*).  Tolerances: BN statistics 1e-5 relative; disparities
mean <= 2e-3 px (batch-stat BN through ~60 layers on an untrained, saturating net)."""
import copy

import pytest
import torch

from oracle import psmnet_oracle as O
from disprcnn_amd.utils import synth
from tests.helpers import golden_npz, state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_bn_stats_and_apply_kernels(dev):
    from disprcnn_amd import engine as E
    x = synth.hash_uniform("bn:x", (3, 40, 5, 9, 11), -2.0, 3.0) + torch.arange(40).view(1, -1, 1, 1, 1) * 0.5
    b = E.Blocked(3, 40, 5, 9, 11, 1, 1, 1, dev).from_dense(x.to(dev))
    mean, var, M = E.bn_batch_stats(b)
    assert M == 3 * 5 * 9 * 11
    ref_m, ref_v = x.mean((0, 2, 3, 4)), x.var((0, 2, 3, 4), unbiased=False)
    assert (mean[:40].cpu() - ref_m).abs().max() < 1e-5 * ref_m.abs().max()
    assert (var[:40].cpu() - ref_v).abs().max() < 1e-5 * ref_v.abs().max()
    assert mean[40:].abs().max() == 0                                        # padded channels stay zero
    g = synth.hash_uniform("bn:g", (48,), 0.5, 1.5).to(dev); g[40:] = 0
    be = synth.hash_uniform("bn:b", (48,), -0.5, 0.5).to(dev); be[40:] = 0
    r = synth.hash_uniform("bn:r", (3, 40, 5, 9, 11))
    rb = E.Blocked(3, 40, 5, 9, 11, 1, 1, 1, dev).from_dense(r.to(dev))
    y = E.Blocked(3, 40, 5, 9, 11, 1, 1, 1, dev)
    invstd = torch.rsqrt(var + 1e-5)
    E.bn_apply(b, y, rb, mean, invstd, g, be, True)
    s = [1, -1, 1, 1, 1]
    ref = torch.relu((x - ref_m.view(s)) / torch.sqrt(ref_v.view(s) + 1e-5) * g[:40].cpu().view(s) + be[:40].cpu().view(s) + r)
    assert (y.to_dense().cpu() - ref).abs().max() < 2e-5
    v = y.view6()
    assert v[:, :, 0].abs().sum() == 0 and v[:, :, :, :, 0].abs().sum() == 0    # halo untouched


def test_train_forward_from_features_vs_oracle(dev):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("At")
    m = PSMNet(48, 0)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    fl, fr = synth.synth_features(3, 32, 28, 28, tag="trainA")
    with torch.no_grad():
        preds = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
        ref = O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112, training=True)
    assert isinstance(preds, tuple) and len(preds) == 3
    for p, r in zip(preds, ref):
        err = (p.cpu() - r).abs()
        assert err.mean().item() < 2e-3 and err.max().item() < 5e-2, (err.mean().item(), err.max().item())
    # running statistics were updated like nn.BatchNorm3d (momentum 0.1, unbiased variance)
    cost = O.cost_volume(fl, fr, 48, 0)
    raw = torch.nn.functional.conv3d(cost, sd["dres0.0.0.weight"], None, 1, 1)
    mu, var = raw.mean((0, 2, 3, 4)), raw.var((0, 2, 3, 4), unbiased=True)
    got_m, got_v = m.dres0[0][1].running_mean.cpu(), m.dres0[0][1].running_var.cpu()
    assert (got_m - (0.9 * sd["dres0.0.1.running_mean"] + 0.1 * mu)).abs().max() < 1e-4
    assert (got_v - (0.9 * sd["dres0.0.1.running_var"] + 0.1 * var)).abs().max() < 1e-4 * max(1.0, var.max().item())
    assert int(m.dres0[0][1].num_batches_tracked) == 1
    # eval after the train step uses the UPDATED running statistics
    m.eval()
    with torch.no_grad():
        pe = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        re = O.psmnet_from_features(sd2, fl, fr, 48, 0, 112, 112)
    err = (pe - re).abs()
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2


def test_train_forward_images_vs_reference_golden(dev):
    """Full PSMNet, train mode, against what the reference itself produced (Bt_pred*_s4, Bt_loss)."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.utils.loss_utils import PSMLoss
    z = golden_npz()
    m = PSMNet(48, -48)
    m.load_state_dict(state_for("B"), strict=True)
    m = m.to(dev).train()
    left, right = synth.synth_images(2, 224, 224, tag="caseBtrain")
    target = synth.hash_uniform("tgt", (2, 224, 224), -48.0, 48.0)
    mask = (synth.hash_uniform("mask", (2, 224, 224), 0.0, 1.0) > 0.5).to(torch.uint8)
    with torch.no_grad():
        preds = m({"left": left.to(dev), "right": right.to(dev)})
        loss = PSMLoss()(preds, {"disparity": target.to(dev), "mask": mask.to(dev)})
    for i, p in enumerate(preds):
        ref = torch.from_numpy(z[f"Bt_pred{i + 1}_s4"])
        err = (p.cpu()[:, ::4, ::4] - ref).abs()
        assert err.mean().item() < 3e-3, (i, err.mean().item(), err.max().item())
    assert abs(loss.item() - float(z["Bt_loss"])) < 2e-3 * float(z["Bt_loss"])


def test_backward_from_features_vs_oracle_autograd(dev):
    """loss.backward() through the HIP engine (BN-train backward, dgrad via the forward engine, MFMA wgrad, classifier /
    soft-argmin / cost-volume adjoints) vs torch autograd of the CPU oracle.  Tolerance 2e-3 * max|ref| per tensor."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.utils.loss_utils import PSMLoss
    sd = state_for("At")
    m = PSMNet(48, 0)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="bwdA")
    tgt = synth.hash_uniform("bwd:t", (2, 112, 112), 0.0, 47.0)
    mask = (synth.hash_uniform("bwd:m", (2, 112, 112), 0.0, 1.0) > 0.3).to(torch.uint8)
    gl, gr = fl.to(dev).requires_grad_(True), fr.to(dev).requires_grad_(True)
    preds = m.forward_from_features(gl, gr, (112, 112))
    loss = PSMLoss()(preds, {"disparity": tgt.to(dev), "mask": mask.to(dev)})
    loss.backward()
    # oracle
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
           for k, v in sd.items()}
    rl, rr = fl.clone().requires_grad_(True), fr.clone().requires_grad_(True)
    rp = O.psmnet_from_features(sdr, rl, rr, 48, 0, 112, 112, training=True)
    rloss = O.psm_loss(rp, tgt, mask)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 1e-3 * abs(rloss.item())
    named = dict(m.named_parameters())
    checked = 0
    for k, v in sdr.items():
        if not (torch.is_tensor(v) and v.requires_grad) or k.startswith("feature_extraction"):
            continue
        ref = v.grad
        got = named[k].grad
        assert got is not None, k
        scale = ref.abs().max().item() + 1e-12
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (k, err, scale)
        checked += 1
    assert checked == 514 - 361 - sum(1 for k in sd if not k.startswith("feature_extraction") and k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    for got, ref in ((gl.grad, rl.grad), (gr.grad, rr.grad)):
        assert (got.cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-7


def test_backward_through_images_is_refused_loudly(dev):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, -48).to(dev).train()
    l, r = synth.synth_images(1, 224, 224, tag="nograd")
    with pytest.raises(NotImplementedError):
        m((l.to(dev), r.to(dev)))
