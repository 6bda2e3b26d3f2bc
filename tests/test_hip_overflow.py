"""Range guard of the split-f16 path (round 6; csrc/s16_ovf.h, engine.guarded).

The reference computes in fp32 (config/defaults.py:22; convbn_3d submodule.py:19-22) and has no activation range limit; an RS16 value (hi + lo
fp16) must stay within +-65504.  VERDICT r5 weak #1: the round-5 kernels clamped silently.  Now every kernel that writes split-f16 values
reports a clamped value (|v| > 65504, Inf, NaN) through a device word, and the forward pass is repeated on the fp32 MFMA kernels ("auto")
or raises ("f16x2").  Held here: (a) every kernel family sets the word exactly when a value of the MAP leaves the range (idle lanes, halos
and dropped planes do not count); (b) a PSMNet whose dres0 activations exceed 65504 -- an equivalent re-parametrisation of the golden
weights, so the CPU oracle's output is unchanged -- still meets the oracle bound under "auto", and raises under "f16x2"; (c) the same
through a replayed HIP graph, through the 2D feature CNN and through the ResNet-FPN trunk; (d) an un-calibrated (default BatchNorm
statistics, reference init) state dict gets the fp32-path result.
"""
import warnings

import pytest
import torch

from disprcnn_amd import engine as E
from disprcnn_amd import s16
from disprcnn_amd.utils import synth
from oracle import psmnet_oracle as O
from tests.helpers import state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


class _Scope:
    """A guard scope around bare kernel launches (what engine.guarded opens around a forward pass)."""

    def __init__(self, dev):
        self.g = E.OverflowGuard(dev)

    def __enter__(self):
        E._GUARD["cur"] = self.g
        return self.g

    def __exit__(self, *a):
        E._GUARD["cur"] = None


def test_converters_report_values_outside_the_fp16_range(dev):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 32, 3, 5, 7, generator=g)
    for bad, expect in ((None, False), (65504.0, False), (-65504.0, False), (65536.0, True), (-7.0e4, True), (float("inf"), True), (float("nan"), True)):
        xx = x.clone()
        if bad is not None:
            xx[1, 17, 2, 3, 4] = bad
        with _Scope(dev) as gd:
            t = E.RS16(2, 32, 3, 5, 7, 1, dev).from_dense(xx.to(dev))
            assert gd.tripped() == expect, ("dense", bad)
            blk = E.Blocked(2, 32, 3, 5, 7, 1, 1, 1, dev).from_dense(xx.to(dev))
            E.RS16(2, 32, 3, 5, 7, 1, dev).from_blocked(blk)
            assert gd.tripped() == expect, ("blocked", bad)
            assert gd.tripped() is False                            # read-and-clear
        if bad is not None and bad == bad and abs(bad) != float("inf"):
            got = t.to_dense().cpu()[1, 17, 2, 3, 4].item()         # stored clamped (the documented saturation), everything else exact
            assert got == max(min(bad, 65504.0), -65504.0)
    # without a scope nothing is reported and nothing breaks
    E.RS16(2, 32, 3, 5, 7, 1, dev).from_dense((x * 1e6).to(dev))
    torch.cuda.synchronize()


def _run_layer(dev, kind, N, cin, cout, D, H, W, relu, boost, form="y16", with_res=False, dil=1):
    """One launch of a kernel family on O(1) inputs with the epilogue scale multiplied by `boost`; -> (guard tripped, max |pre-clamp value|
    of the map according to an fp32 torch convolution)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(("s1", "cv", "s2", "up", "2d").index(kind) * 101 + N + cin + 3 * cout + 7 * D + 11 * H + 13 * W + len(form))
    if kind == "2d":
        x = torch.randn(N, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        od = (1, H, W)
    elif kind == "up":
        x = torch.randn(N, cin, D, H, W, generator=g)
        w = torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin / 8)) ** 0.5
        od = (2 * D, 2 * H, 2 * W)
    else:
        x = torch.randn(N, cin, D, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
        od = (D, H, W) if kind in ("s1", "cv") else (D // 2, H // 2, W // 2)
    scale = (torch.rand(cout, generator=g) + 0.5) * boost
    shift = torch.randn(cout, generator=g) * 0.1
    wd = w.to(dev)
    wp, wexp = s16.pack_weight_s16(wd.transpose(0, 1).contiguous() if kind == "up" else wd)
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    if kind == "2d":
        y = F.conv2d(x, w, padding=dil, dilation=dil).unsqueeze(2)
    elif kind == "up":
        y = F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)
    elif kind == "cv":
        L, R = x[:, :32, 0], x[:, 32:, 0]
        from tests.test_hip_s16 import _ref_costvol
        y = F.conv3d(_ref_costvol(L, R, 0, D), w, padding=1)
    else:
        y = F.conv3d(x, w, padding=1, stride=1 if kind == "s1" else 2)
    y = y * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    res = torch.randn(N, cout, *od, generator=g) if with_res else None
    if with_res:
        y = y + res
    if relu:
        y = y.clamp_min(0)                   # (a negative pre-activation the ReLU zeroes is not a range violation)
    pd = 0 if kind == "2d" else 1
    with _Scope(dev) as gd:
        plan = E.ConvPlanS16(N, cin, cout, 1 if kind == "2d" else D, H, W, relu, cv=(kind == "cv"), device=dev, kind="s1" if kind == "cv" else kind, dil=dil)
        r16 = E.RS16(N, cout, *od, pd, dev).from_dense((res if kind != "2d" else res[:, :, 0]).to(dev)) if with_res else None
        assert not gd.tripped()
        kw = {}
        if form == "y16":
            kw["y16"] = E.RS16(N, cout, *od, pd, dev)
        elif form == "y32":
            kw["y32"] = E.Blocked(N, cout, D, H, W, 1, 1, 1, dev)
        else:
            w1 = torch.randn(1, 32, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
            hp, _ = s16.pack_head_weight_s16(w1)
            kw["head"] = (hp.to(dev), torch.zeros(N, D, H, W, 12, device=dev))
        if kind == "cv":
            lf = E.RS16(N, 32, 1, H, W, 0, dev).from_dense(x[:, :32, 0].contiguous().to(dev))
            rf = E.RS16(N, 32, 1, H, W, 0, dev).from_dense(x[:, 32:, 0].contiguous().to(dev))
            plan.run(None, wp, sc, shift.to(dev), left=lf, right=rf, lo4=0, **kw)
        else:
            x16 = E.RS16(N, cin, 1 if kind == "2d" else D, H, W, pd, dev).from_dense(x.to(dev))
            plan.run(x16, wp, sc, shift.to(dev), res=r16, **kw)
        return gd.tripped(), y.abs().max().item()


LAYERS = [
    # kind, N, cin, cout, D, H, W, relu, form, res
    ("s1", 2, 32, 32, 6, 28, 28, True, "y16", False),       # 1x28 tiles, K split over two waves
    ("s1", 2, 32, 32, 6, 28, 28, False, "y16", True),       # ... residual form, no ReLU (negative overflow counts)
    ("s1", 2, 32, 32, 6, 28, 28, True, "y32", False),       # blocked fp32 output form
    ("s1", 2, 32, 32, 6, 28, 28, True, "head", False),      # fused cout-1 head: the value it multiplies
    ("s1", 2, 64, 64, 6, 14, 14, True, "y16", True),        # 2x14 tiles, K over four waves
    ("s1", 3, 64, 64, 3, 7, 7, True, "y16", False),         # 4x7 tiles, ragged last row tile (idle rows must not report)
    ("cv", 2, 64, 32, 6, 28, 28, True, "y16", False),       # cost-volume form
    ("cv", 80, 64, 32, 3, 28, 28, True, "y16", False),      # ... at a batch the two-tiles-per-wave kernel takes (convs16w.hip)
    ("s2", 2, 32, 64, 12, 28, 28, True, "y16", False),      # stride 2, cout split
    ("s2", 2, 64, 64, 6, 14, 14, True, "y16", False),       # stride 2, 4x7 tiles (ragged)
    ("up", 2, 64, 64, 3, 7, 7, True, "y16", True),          # transposed, 4x7 input tiles
    ("up", 2, 64, 32, 6, 14, 14, False, "y16", True),       # transposed, 2x14 input tiles
    ("2d", 2, 32, 32, 1, 30, 57, True, "y16", False),       # 2D, ragged map
    ("2d", 2, 128, 128, 1, 28, 56, False, "y16", True),     # 2D, K slices per wave
]


@pytest.mark.parametrize("kind,N,cin,cout,D,H,W,relu,form,with_res", LAYERS)
def test_conv_kernels_report_exactly_when_the_map_leaves_the_range(dev, kind, N, cin, cout, D, H, W, relu, form, with_res):
    hit, m1 = _run_layer(dev, kind, N, cin, cout, D, H, W, relu, 1.0, form, with_res)
    assert not hit and m1 < 100.0, (hit, m1)                          # O(1) activations: silent (also: over-read idle lanes / planes do not report)
    # a boost that puts the largest value just INSIDE the range, then one just outside
    inside = 0.9 * 65504.0 / m1
    for _ in range(4):                          # (shift and residual do not scale with the boost: settle on a boost whose maximum is ~0.9 of the limit)
        hit, m = _run_layer(dev, kind, N, cin, cout, D, H, W, relu, inside, form, with_res)
        if 0.8 * 65504.0 < m < 0.97 * 65504.0:
            break
        inside *= 0.9 * 65504.0 / m
    assert not hit and m < 65504.0, (hit, m)
    hit, m = _run_layer(dev, kind, N, cin, cout, D, H, W, relu, inside * 1.5, form, with_res)
    assert hit and m > 65504.0, (hit, m)


def test_nan_or_inf_in_a_folded_bn_parameter_is_reported(dev):
    """NaN cannot survive a clamp (v_med3 returns a finite operand), so the kernels test the folded BN scale / shift they load: a model with
    a NaN / Inf parameter must not answer silently (the fp32 reference would propagate it)."""
    g = torch.Generator().manual_seed(5)
    for kind, (N, cin, cout, D, H, W) in (("s1", (2, 32, 32, 6, 28, 28)), ("s2", (2, 32, 64, 6, 14, 14)), ("up", (2, 64, 32, 3, 7, 7)), ("2d", (2, 64, 64, 1, 28, 28))):
        shape_w = (cout, cin, 3, 3) if kind == "2d" else (cout, cin, 3, 3, 3)
        wp, wexp = s16.pack_weight_s16((torch.randn(shape_w, generator=g) * 0.05).to(dev))
        od = {"s1": (D, H, W), "s2": (D // 2, H // 2, W // 2), "up": (2 * D, 2 * H, 2 * W), "2d": (1, H, W)}[kind]
        pd = 0 if kind == "2d" else 1
        x16 = E.RS16(N, cin, D, H, W, pd, dev).from_dense(torch.randn(N, cin, D, H, W, generator=g).to(dev) if kind != "2d" else torch.randn(N, cin, H, W, generator=g).to(dev))
        plan = E.ConvPlanS16(N, cin, cout, D, H, W, True, device=dev, kind=kind)
        for which, bad in (("scale", float("nan")), ("shift", float("inf")), (None, 0.0)):
            sc = torch.full((cout,), 2.0 ** -wexp, device=dev)
            sh = torch.zeros(cout, device=dev)
            if which == "scale":
                sc[cout - 3] = bad
            elif which == "shift":
                sh[5] = bad
            with _Scope(dev) as gd:
                plan.run(x16, wp, sc, sh, y16=E.RS16(N, cout, *od, pd, dev))
                assert gd.tripped() == (which is not None), (kind, which)


def test_dilated_2d_kernel_reports(dev):
    for boost, expect in ((1.0, False), (1e5, True)):
        hit, _ = _run_layer(dev, "2d", 2, 128, 128, 1, 56, 56, True, boost, "y16", False, dil=2)
        assert hit == expect


def _boosted_state(case, factor=1.0e5):
    """The golden state dict re-parametrised so that dres0[0]'s output is `factor` times larger and dres0[2] divides it out again:
    BN(gamma, beta) -> (factor * gamma, factor * beta) (ReLU is positively homogeneous), next conv weight / factor.  Same function in exact
    arithmetic -- the CPU oracle's output moves by fp32 rounding only -- but the activation between the two layers is O(1e5) > 65504."""
    sd = {k: v.clone() for k, v in state_for(case).items()}
    sd["dres0.0.1.weight"] *= factor
    sd["dres0.0.1.bias"] *= factor
    sd["dres0.2.0.weight"] /= factor
    return sd


def _psmnet(dev, sd, mx=48, mn=0, **attrs):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(mx, mn)
    m.load_state_dict(sd, strict=True)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m.to(dev).eval()


def test_psmnet_auto_repeats_on_fp32_when_dres0_exceeds_the_range(dev):
    sd = _boosted_state("A")
    N = 5
    fl, fr = synth.synth_features(N, 32, 28, 28, tag="ovf")
    with torch.no_grad():
        ref = O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
        ref_plain = O.psmnet_from_features(state_for("A"), fl, fr, 48, 0, 112, 112)
    assert (ref - ref_plain).abs().max().item() < 2e-2                  # the re-parametrisation is the same function (fp32 rounding apart)
    m_auto = _psmnet(dev, sd, graph_eval=False)
    m_f32 = _psmnet(dev, sd, regressor_math="f32", graph_eval=False)
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m_auto.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        want = m_f32.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
    pol = m_auto._rt._guard.policy
    assert pol.overflows == 1 and pol.checks == 1
    assert any("split-f16 range" in str(x.message) for x in w)
    assert torch.equal(got, want)                                       # the repeat IS the fp32 path
    err = (got - ref).abs()
    print(f"boosted dres0, auto: mean/max err px vs oracle {err.mean().item():.3e} {err.max().item():.3e}")
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2         # oracle parity at the bound of test_hip_parity.py
    # the silent round-5 behaviour, for the record: without the check the clamped activations give a different answer
    m_off = _psmnet(dev, sd, graph_eval=False, overflow_check=False)
    with torch.no_grad():
        bad = m_off.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
    print(f"unchecked split-f16 on the same weights: max err {(bad - ref).abs().max().item():.3e} px")
    assert (bad - ref).abs().max().item() > 2e-2
    # back-off: the next pass goes straight to fp32 (no second split-f16 attempt), the one after tries again
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        again = m_auto.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        assert torch.equal(again, want) and pol.checks == 1
        m_auto.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
        assert pol.checks == 2 and pol.overflows == 2
    # "f16x2" asked for explicitly: raise instead of answering in other arithmetic
    m_strict = _psmnet(dev, sd, regressor_math="f16x2", graph_eval=False)
    with torch.no_grad(), pytest.raises(RuntimeError, match="split-f16 range"):
        m_strict.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
    # in-range weights: no repeat, no warning, the split-f16 schedule's own result
    m_ok = _psmnet(dev, state_for("A"), graph_eval=False)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        ok = m_ok.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
    assert m_ok._rt._guard.policy.overflows == 0 and m_ok._rt._guard.policy.checks == 1
    assert (ok - ref_plain).abs().max().item() < 2e-2


def test_overflow_is_seen_through_a_replayed_graph(dev):
    sd = _boosted_state("A")
    fl, fr = synth.synth_features(4, 32, 28, 28, tag="ovfg")
    fl, fr = fl.to(dev), fr.to(dev)
    m = _psmnet(dev, sd, graph_eval=True)
    want = _psmnet(dev, sd, regressor_math="f32", graph_eval=False)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = want.forward_from_features(fl, fr, (112, 112))
        outs = []
        for i in range(6):
            if m._rt is not None and m._rt._guard is not None:
                m._rt._guard.policy.skip = 0                               # no back-off: every pass tries the split-f16 graph first
            outs.append(m.forward_from_features(fl, fr, (112, 112)))
    assert all(torch.equal(o, w) for o in outs)
    pol = m._rt._guard.policy
    assert pol.overflows == 6 and len(m._rt._graphs) == 2               # one split-f16 graph (reports through its captured word), one fp32 graph


def test_feature_cnn_overflow_repeats_on_fp32(dev):
    """Full PSMNet on crops: firstconv[0]'s output boosted beyond the range (undone by firstconv[2]) -> the RS16 converter at the seam
    reports, the pass is repeated with both the 2D CNN and the regressor on the fp32 kernels."""
    sd = {k: v.clone() for k, v in state_for("B").items()}
    sd["feature_extraction.firstconv.0.1.weight"] *= 1.0e5
    sd["feature_extraction.firstconv.0.1.bias"] *= 1.0e5
    sd["feature_extraction.firstconv.2.0.weight"] /= 1.0e5
    left, right = synth.synth_images(2, 224, 224, tag="ovf2d")
    m = _psmnet(dev, sd, 48, -48, graph_eval=False)
    f32 = _psmnet(dev, sd, 48, -48, graph_eval=False, regressor_math="f32", feature_math="f32")
    plain = _psmnet(dev, state_for("B"), 48, -48, graph_eval=False)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = m((left.to(dev), right.to(dev)))
        want = f32((left.to(dev), right.to(dev)))
        base = plain((left.to(dev), right.to(dev)))
    assert m._rt._guard.policy.overflows == 1 and plain._rt._guard.policy.overflows == 0
    assert torch.equal(got, want)
    d = (got - base).abs()
    print(f"boosted firstconv vs plain weights: mean {d.mean().item():.3e} max {d.max().item():.3e} px")
    assert d.mean().item() < 1e-3


def test_uncalibrated_state_dict_matches_the_fp32_path(dev):
    """Reference init (stackhourglass.py:90-104) with DEFAULT BatchNorm statistics (mean 0, var 1): activations are not normalised, the
    untrained net grows layer by layer.  Whatever the guard decides, "auto" must give the fp32 kernels' answer: bit-identical when it
    repeated the pass, fp32-class close when every value stayed in range -- and the CPU oracle's answer either way."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.modeling.psmnet.submodule import reference_init_
    torch.manual_seed(7)
    net = PSMNet(48, 0)
    reference_init_(net)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for scale in (1.0, 300.0):
        fl, fr = synth.synth_features(3, 32, 28, 28, tag="unc")
        fl, fr = fl * scale, fr * scale
        m = _psmnet(dev, sd, graph_eval=False)
        f32 = _psmnet(dev, sd, regressor_math="f32", graph_eval=False)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
            want = f32.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
            ref = O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
        over = m._rt._guard.policy.overflows
        d, e = (got - want).abs(), (got - ref).abs()
        print(f"un-calibrated, features x{scale:g}: overflowed {over}; vs fp32 path max {d.max().item():.3e}; vs oracle mean {e.mean().item():.3e} "
              f"max {e.max().item():.3e} px; |disp| max {ref.abs().max().item():.2f}")
        assert torch.isfinite(got).all()
        if over:
            assert torch.equal(got, want)
        # an untrained, un-normalised net has a near-one-hot softmax: a rounding-level change of a cost can move the arg-max by a bin, so the
        # bound is on the bulk (the fp32 HIP path and the oracle differ the same way)
        assert (e < 2e-2).float().mean().item() > 0.98
        assert (d < 2e-2).float().mean().item() > 0.98


def test_trunk_overflow_repeats_on_fp32(dev):
    """ResNet-50-FPN: an input scaled so that a split-f16 3x3 layer on the large maps leaves the range -> the pass is repeated on the fp32
    kernels and equals the TRUNK_S16-off result."""
    from disprcnn_amd.modeling.backbone import build_backbone
    from disprcnn_amd.modeling.detector.disprcnn import default_cfg_2d
    torch.manual_seed(0)
    bb = build_backbone(default_cfg_2d("R-50-FPN")).to(dev).eval()
    # un-calibrated BatchNorm (running stats 0 / 1): activations grow through layer1; push the input until the bridged layer overflows
    x = torch.rand(2, 3, 375, 1242, device=dev) * 1.0e6
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bb(x)
        rt = bb._rt
        assert rt._guard.policy.checks == 1, "no split-f16 layer ran: the trunk's large-map bridge is off?"
        over = rt._guard.policy.overflows
        saved = dict(E.TRUNK_S16)
        try:
            E.TRUNK_S16["enabled"] = False
            rt._ws.clear()
            want = bb(x)
        finally:
            E.TRUNK_S16.update(saved)
    assert over == 1
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_2d_stage_reads_one_guard_for_trunk_and_rpn_head(dev):
    """DispRCNN.forward: the trunk's bridged 3x3 layers and the RPN head's 256 -> 512 convolution report to ONE guard, read once at the end
    of the stage; an input that leaves the range repeats the whole stage on the fp32 kernels and equals the TRUNK_S16-off result."""
    from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
    torch.manual_seed(1)
    m = DispRCNN(default_cfg_2d("R-50-FPN", post_nms_top_n_test=40)).to(dev).eval()
    pair = torch.rand(2, 3, 375, 1242, device=dev)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m({"left": pair[:1], "right": pair[1:]})
        g = m._guard
        assert g.policy.checks == 1 and g.policy.overflows == 0              # split-f16 layers ran, one read, in range
        assert m.backbone._rt._guard.policy.checks == 0                       # the trunk's own guard stayed idle: it reported to the detector's
        big = pair * 1.0e6
        got = m({"left": big[:1], "right": big[1:]})
        assert g.policy.checks == 2 and g.policy.overflows == 1
        saved = dict(E.TRUNK_S16)
        try:
            E.TRUNK_S16["enabled"] = False
            m.backbone._rt._ws.clear()
            m.overflow_check = False
            want = m({"left": big[:1], "right": big[1:]})
        finally:
            E.TRUNK_S16.update(saved)
            m.overflow_check = True
    for side in ("left", "right"):
        for a, b in zip(got[side], want[side]):
            assert len(a) == len(b) and torch.equal(a.bbox, b.bbox) and torch.equal(a.get_field("scores"), b.get_field("scores"))
