"""GPU tests of the train-mode forward (per-GPU batch-statistics BatchNorm, three heads) against the CPU oracle and the
reference's recorded train-mode outputs (tests/golden: Bt_*).  Tolerances: BN statistics 1e-5 relative; disparities
mean <= 2e-3 px (batch-stat BN through ~60 layers on an untrained, saturating net)."""
import copy

import pytest
import torch
import torch.nn.functional as F

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


def _fp32_noise_floor(loss_fn, sd, extra_inputs, ref64):
    """What the REFERENCE graph itself loses in fp32: the oracle's autograd run in float32 against its float64 run (ref64: name -> grad),
    per tensor -> {name: (max-norm error, 1 - cosine)}.  The whole-net gradient bounds below are stated against this floor: a batch-stat-BN
    net at batch 2 amplifies last-ulp differences through ReLU-mask flips, differently in every tensor."""
    d = {k: (v.clone().float().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v.clone())
         for k, v in sd.items()}
    for k, v in extra_inputs.items():
        d[k] = v.clone().float().requires_grad_(True)
    loss_fn(d).backward()
    out = {}
    for k, g64 in ref64.items():
        if d[k].grad is None:
            continue
        g32, g64 = d[k].grad.double().reshape(-1), g64.reshape(-1)
        cos = torch.dot(g32, g64).item() / (g32.norm().item() * g64.norm().item() + 1e-300)
        out[k] = ((g32 - g64).abs().max().item() / (g64.abs().max().item() + 1e-300), 1.0 - cos)
    return out


def _check_grad(got, ref, name, floor, err_bar=5e-3, cos_bar=1e-4, slack=4.0):
    """Per tensor: max-norm error <= 5e-3 and 1 - cosine <= 1e-4 -- or, where the reference graph's own fp32 run is worse than a quarter of
    that (floor), `slack` times its distance from fp64.  A 1 % wiring error in any layer is 2x..10x outside either bound."""
    got, ref = got.detach().cpu().double().reshape(-1), ref.reshape(-1)
    cos = torch.dot(got, ref).item() / (got.norm().item() * ref.norm().item() + 1e-300)
    err = (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-300)
    f_err, f_cos = floor.get(name, (0.0, 0.0))
    assert err <= max(err_bar, slack * f_err), (name, "max-norm error", err, "fp32 floor of the reference graph", f_err)
    assert 1.0 - cos <= max(cos_bar, slack * f_cos), (name, "1 - cosine", 1.0 - cos, "fp32 floor", f_cos)
    return err


def test_backward_from_features_vs_oracle_autograd(dev):
    """loss.backward() through the HIP engine (BN-train backward, dgrad via the forward engine, MFMA wgrad, classifier /
    soft-argmin / cost-volume adjoints) vs torch autograd of the CPU oracle run in fp64.
    Per tensor (_check_grad): max-norm error <= 1e-4 and 1 - cosine <= 1e-8 (measured on the tempered case: max 5.5e-6, median 2.4e-6) --
    unless the reference graph's OWN fp32 run (the oracle's autograd in float32, computed here) is further than a quarter of that from
    fp64 for that tensor, then four times its distance.  Single sites are pinned to 2e-4 in the per-site test."""
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
    # oracle (fp64)
    dt = torch.float64
    sdr = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))
               else (v.to(dt) if v.is_floating_point() else v)) for k, v in sd.items()}
    rl, rr = fl.to(dt).requires_grad_(True), fr.to(dt).requires_grad_(True)
    rp = O.psmnet_from_features(sdr, rl, rr, 48, 0, 112, 112, training=True)
    rloss = O.psm_loss(rp, tgt.to(dt), mask)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 1e-5 * abs(rloss.item())
    named = dict(m.named_parameters())
    floor = _fp32_noise_floor(lambda d: O.psm_loss(O.psmnet_from_features(d, d["__l"], d["__r"], 48, 0, 112, 112, training=True), tgt.to(d["__l"].dtype), mask),
                              sd, {"__l": fl, "__r": fr}, {**{k: v.grad for k, v in sdr.items() if torch.is_tensor(v) and v.requires_grad},
                                                           "__l": rl.grad, "__r": rr.grad})
    errs = []
    for k, v in sdr.items():
        if not (torch.is_tensor(v) and v.requires_grad) or k.startswith("feature_extraction"):
            continue
        assert named[k].grad is not None, k
        errs.append(_check_grad(named[k].grad, v.grad, k, floor, err_bar=1e-4, cos_bar=1e-8))
    assert len(errs) == 514 - 361 - sum(1 for k in sd if not k.startswith("feature_extraction") and k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    _check_grad(gl.grad, rl.grad, "__l", floor, err_bar=1e-4, cos_bar=1e-8)
    _check_grad(gr.grad, rr.grad, "__r", floor, err_bar=1e-4, cos_bar=1e-8)
    errs.sort()
    print("Config A whole-net gradients: median / max max-norm error", errs[len(errs) // 2], errs[-1])
    assert errs[len(errs) // 2] <= 1e-5, errs[len(errs) // 2]


def test_full_psmnet_backward_vs_reference_gradients(dev):
    """Train step on image crops: loss.backward() through 2D CNN (both views), cost volume and regressor on the HIP engine,
    against (a) the gradient samples the REFERENCE itself recorded (tests/golden Bt_g:*), (b) the fp64 oracle's autograd
    for every parameter, per tensor max-norm error <= 6e-2 and cosine >= 0.98, median <= 1.5e-2, at most 5 % of the tensors beyond
    2e-2.  These bars are loose on purpose: this ~90-layer batch-stat-BN net at batch 2 flips ReLU masks on last-ulp changes of a batch
    statistic (the reference graph's own fp32 run sits up to 5e-3 from its fp64 run, a different rounding realisation up to 4e-2: measured
    here with a per-tensor floor, which does not bound the next realisation; the worst tensor is SPP branch1, whose BatchNorm sees 2
    samples per channel).  The TIGHT whole-net wiring check is the Config-A test above (every regressor tensor to 1e-4 on the tempered
    case); the 2D CNN's backward kernels are pinned per site below.  Loss to 1e-5 relative; sampled
    reference gradients to 2e-2 * max|ref| (the reference's own fp32 run sits 2e-3..6e-3 from fp64 there).
    The per-site tests below pin every backward kernel to 2e-4 on well conditioned single layers."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.utils.loss_utils import PSMLoss
    z = golden_npz()
    sd = state_for("B")
    m = PSMNet(48, -48)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    left, right = synth.synth_images(2, 224, 224, tag="caseBtrain")
    target = synth.hash_uniform("tgt", (2, 224, 224), -48.0, 48.0)
    mask = (synth.hash_uniform("mask", (2, 224, 224), 0.0, 1.0) > 0.5).to(torch.uint8)
    preds = m({"left": left.to(dev), "right": right.to(dev)})
    loss = PSMLoss()(preds, {"disparity": target.to(dev), "mask": mask.to(dev)})
    loss.backward()
    assert abs(loss.item() - float(z["Bt_loss"])) < 1e-5 * float(z["Bt_loss"])
    named = dict(m.named_parameters())
    for name in ("dres0.0.0.weight", "dres2.conv5.0.weight", "classif3.2.weight", "dres4.conv6.1.weight",
                 "feature_extraction.lastconv.2.weight"):
        g = named[name].grad.detach().cpu().reshape(-1)
        ref = torch.from_numpy(z[f"Bt_g:{name}_val"])
        got = g[torch.from_numpy(z[f"Bt_g:{name}_idx"])]
        scale = max(ref.abs().max().item(), float(z[f"Bt_g:{name}_abssum"]) / g.numel())
        assert (got - ref).abs().max().item() <= 2e-2 * scale + 1e-7, (name, (got - ref).abs().max().item(), scale)
        assert abs(g.double().abs().sum().item() - float(z[f"Bt_g:{name}_abssum"])) <= 2e-2 * float(z[f"Bt_g:{name}_abssum"])
    # every parameter vs the oracle's autograd in fp64
    dt = torch.float64
    sdr = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))
               else (v.to(dt) if v.is_floating_point() else v)) for k, v in sd.items()}
    rp = O.psmnet_forward(sdr, left.to(dt), right.to(dt), 48, -48, training=True)
    O.psm_loss(rp, target.to(dt), mask).backward()
    errs = []
    for k, v in sdr.items():
        if not (torch.is_tensor(v) and v.requires_grad):
            continue
        assert named[k].grad is not None, k
        errs.append(_check_grad(named[k].grad, v.grad, k, {}, err_bar=6e-2, cos_bar=2e-2))
    errs.sort()
    assert len(errs) == sum(1 for _ in m.parameters())
    beyond = [sum(e > b for e in errs) for b in (5e-3, 2e-2)]
    print("Config B whole-net gradients: median / max max-norm error", errs[len(errs) // 2], errs[-1], "| tensors beyond 5e-3 / 2e-2:", beyond, "of", len(errs))
    assert errs[len(errs) // 2] <= 1.5e-2, errs[len(errs) // 2]          # measured: median 8.5e-3, max 3.8e-2, 4 of 259 tensors beyond 2e-2
    assert beyond[1] <= len(errs) // 20, beyond


@pytest.mark.parametrize("n,cin,cout,dims,stride", [(5, 32, 32, (6, 28, 28), 1), (3, 32, 64, (12, 28, 28), 2), (2, 64, 64, (3, 7, 7), 1),
                                                    (4, 16, 48, (4, 9, 30), 1), (2, 32, 32, (1, 40, 56), 1)])
def test_wgrad_partial_sums_vs_atomic_flush_and_oracle(dev, n, cin, cout, dims, stride):
    """Weight gradient with the waves' partial sums reduced in wave order (DRC_WGRAD_SCRATCH_FLOATS workspace; wgrad_slide and
    wgrad_kernel<9>) against F.conv3d's autograd (fp64), against the atomicAdd flush, and bit-identical across two runs."""
    from disprcnn_amd import engine as E
    from disprcnn_amd.modeling.psmnet import train as T
    od = tuple(-(-d // stride) for d in dims)
    x = synth.hash_uniform(f"wg{n}{cin}{dims}:x", (n, cin) + dims)
    dy = synth.hash_uniform(f"wg{n}{cout}{dims}:dy", (n, cout) + od, -1.0, 1.0)
    w = torch.zeros(cout, cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), w, None, stride, 1).backward(dy.double())
    ref = w.grad.permute(1, 0, 2, 3, 4).reshape(cin, cout, 27)                      # [ci][co][t]
    xb = E.Blocked(n, cin, *dims, 1, 1, 1, dev).from_dense(x.to(dev))
    db = E.Blocked(n, cout, *od, 1, 1, 1, dev).from_dense(dy.to(dev))
    cls = dict(n=(3, 3, 3), first=(0, 0, 0), step=(1, 1, 1))
    outs = []
    for partial in (True, True, False):
        T.WGRAD_PARTIALS["enabled"] = partial
        try:
            outs.append(T.wgrad(xb, db, cls, stride)[:cin, :cout].cpu())
        finally:
            T.WGRAD_PARTIALS["enabled"] = True
    assert torch.equal(outs[0], outs[1]), "partial-sum reduction is not run-to-run reproducible"
    tol = 2e-5 * ref.abs().max().item() + 1e-5
    for got in (outs[0], outs[2]):
        assert (got.double() - ref).abs().max().item() <= 5 * tol


def test_spp_adjoints(dev):
    """drc_bilinear_up_blocked_bwd / drc_avgpool2d_blocked_bwd are the exact adjoints of the forward SPP kernels
    (reference: autograd of F.interpolate(bilinear, align_corners=True) and AvgPool2d, submodule.py:76-96,128-137):
    <fwd(x), g> == <x, bwd(g)> to fp32 rounding, at the real branch sizes (56x56 map, pools 8/16/32... -> 7/3/1)."""
    from disprcnn_amd import _lib, engine as E
    lib, sp = _lib.lib(), E._stream_ptr(dev)
    n, H4, W4 = 2, 56, 72
    for k in (8, 16, 32):
        oh, ow = H4 // k, W4 // k
        # bilinear upsample of a 32-channel map into channel blocks 2..3 of a 5-block tensor
        x = E.Blocked(n, 32, 1, oh, ow, 0, 0, 0, dev)
        xd = synth.hash_uniform(f"spp{k}:x", (n, 32, 1, oh, ow)).to(dev)
        x.from_dense(xd)
        cat = E.Blocked(n, 80, 1, H4, W4, 0, 1, 1, dev)
        _lib.check(lib.drc_bilinear_up_blocked(E._ptr(x.storage), E._ptr(cat.storage), n, 2, oh, ow, 0, H4, W4, cat.ph, cat.cb, 2, sp), "up")
        y = cat.to_dense()[:, 32:64]
        ref = F.interpolate(xd.squeeze(2).cpu(), (H4, W4), mode="bilinear", align_corners=True)
        assert (y.squeeze(2).cpu() - ref).abs().max().item() < 1e-5
        gcat = E.Blocked(n, 80, 1, H4, W4, 0, 1, 1, dev)
        gd = torch.zeros(n, 80, 1, H4, W4, device=dev)
        gd[:, 32:64] = synth.hash_uniform(f"spp{k}:g", (n, 32, 1, H4, W4)).to(dev)
        gcat.from_dense(gd)
        gsl = E.BlockedSlice(gcat, 2, 32)
        gx = E.Blocked(n, 32, 1, oh, ow, 0, 0, 0, dev)
        _lib.check(lib.drc_bilinear_up_blocked_bwd(E._ptr(gsl.storage), E._geom8(gsl), E._ptr(gx.storage), E._geom8(gx), sp), "up_bwd")
        lhs = (y.double() * gd[:, 32:64].double()).sum().item()
        rhs = (xd.double() * gx.to_dense().double()).sum().item()
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (k, lhs, rhs)
        xr = xd.squeeze(2).cpu().double().requires_grad_()
        F.interpolate(xr, (H4, W4), mode="bilinear", align_corners=True).backward(gd[:, 32:64].squeeze(2).cpu().double())
        assert (gx.to_dense().squeeze(2).cpu().double() - xr.grad).abs().max().item() <= 2e-5 * xr.grad.abs().max().item()
        # accumulates into grad_x: a second call doubles it
        _lib.check(lib.drc_bilinear_up_blocked_bwd(E._ptr(gsl.storage), E._geom8(gsl), E._ptr(gx.storage), E._geom8(gx), sp), "up_bwd")
        assert (gx.to_dense().squeeze(2).cpu().double() - 2 * xr.grad).abs().max().item() <= 4e-5 * xr.grad.abs().max().item()
        # average pool of a 128-channel slice (blocks 1..8 of a 10-block tensor), adjoint accumulates into the slice
        src = E.Blocked(n, 160, 1, H4, W4, 0, 2, 2, dev)
        sdn = synth.hash_uniform(f"spp{k}:s", (n, 160, 1, H4, W4)).to(dev)
        src.from_dense(sdn)
        gp = E.Blocked(n, 128, 1, oh, ow, 0, 0, 0, dev)
        gpd = synth.hash_uniform(f"spp{k}:gp", (n, 128, 1, oh, ow)).to(dev)
        gp.from_dense(gpd)
        gsrc = E.Blocked(n, 160, 1, H4, W4, 0, 2, 2, dev)
        gs = E.BlockedSlice(gsrc, 1, 128)
        _lib.check(lib.drc_avgpool2d_blocked_bwd(E._ptr(gp.storage), E._geom8(gp), E._ptr(gs.storage), E._geom8(gs), k, sp), "pool_bwd")
        pooled = F.avg_pool2d(sdn[:, 16:144].squeeze(2).cpu(), k, k)
        lhs = (pooled.double() * gpd.squeeze(2).cpu().double()).sum().item()
        gfull = gsrc.to_dense()
        rhs = (sdn[:, 16:144].double() * gfull[:, 16:144].double()).sum().item()
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), (k, lhs, rhs)
        assert gfull[:, :16].abs().max().item() == 0.0 and gfull[:, 144:].abs().max().item() == 0.0


SITES = [  # kind, cin, cout, k, stride, pad, dil, dims, relu, with_res
    ("conv3d", 64, 32, 3, 1, 1, 1, (6, 12, 20), True, False),
    ("conv3d", 32, 32, 3, 1, 1, 1, (4, 10, 12), False, True),
    ("conv3d", 32, 64, 3, 2, 1, 1, (8, 12, 16), True, False),
    ("deconv3d", 64, 64, 3, 2, 1, 1, (3, 6, 8), False, True),
    ("deconv3d", 64, 32, 3, 2, 1, 1, (4, 6, 10), False, False),
    ("conv2d", 3, 32, 3, 2, 1, 1, (24, 40), True, False),
    ("conv2d", 32, 64, 3, 2, 1, 1, (20, 28), True, False),
    ("conv2d", 32, 64, 1, 2, 0, 1, (20, 28), False, False),
    ("conv2d", 64, 128, 1, 1, 0, 1, (10, 14), False, False),
    ("conv2d", 128, 128, 3, 1, 2, 2, (10, 14), False, True),
    ("conv2d", 320, 128, 3, 1, 1, 1, (8, 12), True, False),
]


@pytest.mark.parametrize("kind,cin,cout,k,stride,pad,dil,dims,relu,with_res", SITES)
def test_site_backward_vs_torch_autograd(dev, kind, cin, cout, k, stride, pad, dil, dims, relu, with_res):
    """One conv + batch-stat BN (+res) (+ReLU) site: forward and every gradient (input, residual, conv weight, gamma, beta)
    against torch autograd in fp64 (reference layers: submodule.py:13-22, stackhourglass.py:22-30).  Single sites are
    well conditioned, so the tolerance is fp32 rounding: 2e-4 * max|ref|."""
    import torch.nn as nn
    from disprcnn_amd import engine as E
    from disprcnn_amd.modeling.psmnet.runtime import PSMNetRuntime, _Conv
    from disprcnn_amd.modeling.psmnet.train import RegressorBackward, Grads
    n = 2
    tag = f"S{kind}{cin}{cout}{k}{stride}{dil}"
    is3d = kind != "conv2d"
    if kind == "conv3d":
        conv = nn.Conv3d(cin, cout, 3, stride, 1, bias=False)
    elif kind == "deconv3d":
        conv = nn.ConvTranspose3d(cin, cout, 3, 2, 1, output_padding=1, bias=False)
    else:
        conv = nn.Conv2d(cin, cout, k, stride, pad, dil, bias=False)
    bn = nn.BatchNorm3d(cout) if is3d else nn.BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(synth.hash_uniform(tag + ":w", tuple(conv.weight.shape), -0.1, 0.1))
        bn.weight.copy_(synth.hash_uniform(tag + ":g", (cout,), 0.5, 1.5))
        bn.bias.copy_(synth.hash_uniform(tag + ":b", (cout,), -0.5, 0.5))
    x = synth.hash_uniform(tag + ":x", (n, cin) + dims)
    # fp64 reference
    c64, b64 = __import__("copy").deepcopy(conv).double(), __import__("copy").deepcopy(bn).double()
    x64 = x.double().requires_grad_(True)
    raw64 = c64(x64)
    r64 = synth.hash_uniform(tag + ":r", tuple(raw64.shape)).double().requires_grad_(True) if with_res else None
    y64 = b64.train()(raw64)
    if with_res:
        y64 = y64 + r64
    if relu:
        y64 = torch.relu(y64)
    gy = synth.hash_uniform(tag + ":gy", tuple(y64.shape))
    (y64 * gy.double()).sum().backward()
    # HIP site
    odims = tuple(raw64.shape[2:])
    if is3d:
        mk = lambda c, d: E.Blocked(n, c, d[0], d[1], d[2], 1, 1, 1, dev)
    else:
        halo = max(pad, 1)
        mk = lambda c, d: E.Blocked(n, c, 1, d[0], d[1], 0, halo, halo, dev)
    t = {"x": mk(cin, dims), "y": mk(cout, odims)}
    t["x"].from_dense(x.to(dev) if is3d else x.to(dev).unsqueeze(2))
    if with_res:
        t["r"] = mk(cout, odims)
        t["r"].from_dense(r64.detach().float().to(dev) if is3d else r64.detach().float().to(dev).unsqueeze(2))
    if kind == "conv3d":
        plan = E.plan_conv3d(t["x"], t["y"], stride, cout, relu)
    elif kind == "deconv3d":
        plan = E.plan_deconv3d(t["x"], t["y"], cout, relu)
    else:
        plan = E.plan_conv2d(t["x"], t["y"], k, stride, pad, dil, cout, relu)
    ws = {"t": t, "p": {"s": plan}, "pool": E.WorkspacePool(n, dev)}
    rt = PSMNetRuntime.__new__(PSMNetRuntime)
    rt.device, rt._training, rt._tape, rt._need_input_grad = dev, True, [], True
    W = {"s": _Conv(conv, bn, dev, kind == "deconv3d")}
    rt._site(ws, W, "s", "s", "x", "y", "r" if with_res else None)
    got_y = t["y"].to_dense().cpu()
    got_y = got_y if is3d else got_y.squeeze(2)
    assert (got_y.double() - y64.detach()).abs().max().item() <= 1e-4 * y64.abs().max().item()
    bw = RegressorBackward(rt, ws, W)
    G = Grads(ws, dev)
    G.get("y").from_dense(gy.to(dev) if is3d else gy.to(dev).unsqueeze(2))
    G.have.add("y")
    bw.site(G, "s", "s", "x", "y", "r" if with_res else None)
    torch.cuda.synchronize()

    def chk(got, ref, what):
        s = ref.abs().max().item()
        e = (got.double().cpu() - ref).abs().max().item()
        assert e <= 2e-4 * s + 1e-9, (what, e, s)

    gx = G.get("x").to_dense()
    chk(gx if is3d else gx.squeeze(2), x64.grad, "dx")
    if with_res:
        gr = G.get("r").to_dense()
        chk(gr if is3d else gr.squeeze(2), r64.grad, "dres")
    chk(bw.pg[id(conv.weight)], c64.weight.grad, "dw")
    chk(bw.pg[id(bn.weight)], b64.weight.grad, "dgamma")
    chk(bw.pg[id(bn.bias)], b64.bias.grad, "dbeta")


@pytest.mark.parametrize("shape", [(2, 12, 28, 28, 48, 0, 112, 112), (1, 6, 5, 7, 24, -24, 20, 28), (2, 3, 9, 11, 12, 0, 33, 41)])
def test_upsample_softargmin_backward_vs_oracle_autograd(dev, shape):
    """d disp / d cost of the fused trilinear(align_corners) + softmax + soft-argmin head (stackhourglass.py:169-173) vs torch
    autograd of the oracle in fp64, incl. output sizes that are not multiples of the kernel's 8 x 16 pixel tiles."""
    from disprcnn_amd import _lib, engine as E
    n, dp, hp, wp, ndisp, mn, H, W = shape
    cost = synth.hash_uniform(f"sab{shape}", (n, dp, hp, wp), -2.0, 2.0)
    gd = synth.hash_uniform(f"sabg{shape}", (n, H, W), -1.0, 1.0)
    c64 = cost.double().requires_grad_(True)
    pred = O.upsample_softargmin(c64.unsqueeze(1), mn + ndisp, mn, H, W)
    (pred * gd.double()).sum().backward()
    g = torch.full((n, dp, hp, wp), float("nan"), device=dev)   # overwritten by the gather launch
    cost_d, gd_d = cost.to(dev), gd.to(dev)                      # keep the device tensors alive across the launch
    foot = E.scratch(dev, "softargmin_bwd", _lib.lib().drc_upsample_softargmin_bwd_scratch_floats(n, dp, hp, wp, H, W))
    st = _lib.lib().drc_upsample_softargmin_bwd(E._ptr(cost_d), E._ptr(gd_d), E._ptr(g), n, dp, hp, wp, ndisp, H, W, mn,
                                                E._ptr(foot), foot.numel(), E._stream_ptr(dev))
    _lib.check(st, "drc_upsample_softargmin_bwd")
    ref = c64.grad
    assert (g.cpu().double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7
