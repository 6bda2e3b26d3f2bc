"""GPU tests of the fp16-STORAGE regressor (BASELINE configs[3]: 64 ROIs at 224x224x96, "fp16").

The reference is fp32-only (config/defaults.py:22), so there is no fp16 oracle: single layers are held to the fp32 convolution of
the SAME fp16-rounded operands (what fp16 storage + fp32 accumulation must reproduce up to the output rounding, 2^-11 relative),
and the whole path to the fp32 oracle / the fp32 HIP path.  Stated bounds (mean |err| over all pixels, synthetic UNTRAINED
weights whose sharp softmax is the worst case): Config A (D=48, 25 fp16 roundings deep) <= 5e-2 px -- the bound SURVEY 8c
proposes, measured 2.2e-2; the stress shape (D=96: twice the disparity range under the same relative cost error) <= 1e-1 px,
measured 5.1e-2.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import psmnet_oracle as O
from disprcnn_amd.utils import synth
from tests.helpers import state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _h(t):
    return t.half().float()


@pytest.mark.parametrize("cin,cout,stride,transposed,dims,with_res", [
    (64, 32, 1, False, (4, 12, 28), True), (32, 32, 1, False, (3, 28, 28), False), (32, 64, 2, False, (12, 28, 28), False),
    (64, 64, 2, False, (6, 14, 14), True), (64, 64, 1, False, (3, 7, 7), True), (64, 64, 1, True, (3, 7, 7), True),
    (64, 32, 1, True, (6, 14, 14), False), (24, 40, 1, False, (2, 5, 9), True), (24, 16, 1, True, (3, 9, 30), True),
    (40, 48, 2, False, (6, 10, 18), False), (32, 32, 1, False, (2, 56, 56), True),
    # one input block, <= 32 couts, depth >= 4: the depth-sliding walk (conv16s_kernel), incl. ragged rows / columns and 16 couts
    (32, 32, 1, False, (6, 28, 28), True), (32, 32, 1, False, (5, 20, 17), False), (24, 16, 1, False, (4, 9, 30), True),
    (32, 32, 1, False, (9, 56, 56), True), (64, 64, 1, False, (6, 28, 28), True), (64, 64, 1, False, (5, 14, 17), False), (64, 32, 1, False, (4, 28, 30), True),
    (40, 64, 1, False, (7, 7, 14), False), (64, 32, 1, False, (6, 56, 56), False), (32, 32, 1, False, (5, 16, 20), True), (32, 32, 1, False, (4, 32, 14), False), (20, 16, 1, False, (7, 28, 30), True),
    # round 4 (conv16x.hip): the stress shape's stride-2 / transposed layers at full size, odd input extents, one row / column tiles
    (32, 64, 2, False, (8, 56, 56), True), (64, 32, 1, True, (4, 28, 28), True), (64, 64, 2, False, (5, 27, 31), False),
    (32, 16, 1, True, (2, 3, 5), False), (96, 32, 2, False, (4, 8, 8), True), (96, 64, 1, True, (2, 13, 15), True)])
def test_conv16_layer_vs_fp32_conv_of_rounded_operands(dev, cin, cout, stride, transposed, dims, with_res):
    from disprcnn_amd import engine as E
    n = 2
    x = _h(synth.hash_uniform(f"h{cin}{cout}{stride}{transposed}:x", (n, cin) + dims))
    w = _h(synth.hash_uniform(f"h{cin}{cout}:w", (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3), -0.1, 0.1))
    scale = synth.hash_uniform("h:s", (cout,), 0.5, 1.5)
    shift = synth.hash_uniform("h:b", (cout,), -0.5, 0.5)
    ref = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1) if transposed else F.conv3d(x, w, None, stride, 1)
    ref = ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    res = _h(synth.hash_uniform("h:r", tuple(ref.shape))) if with_res else None
    ref = F.relu(ref + res) if with_res else ref
    xb = E.Blocked16(n, cin, *dims, 1, 1, 1, dev).from_dense(x.to(dev))
    od = tuple(ref.shape[2:])
    yb = E.Blocked16(n, cout, *od, 1, 1, 1, dev)
    rb = E.Blocked16(n, cout, *od, 1, 1, 1, dev).from_dense(res.to(dev)) if with_res else None
    plan = E.plan_deconv3d16(xb, yb, cout, with_res) if transposed else E.plan_conv3d16(xb, yb, stride, cout, with_res)
    cp = E.cout_pad_of(cout)
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = scale.to(dev); sh[:cout] = shift.to(dev)
    assert plan.tile == (stride == 1 and not transposed)            # stride-1 layers: drc_conv16_k3_tile_fwd (conv16t.hip's entry)
    assert plan.tile_x == (None if plan.tile else ("drc_deconv16_k3s2_tile_fwd" if transposed else "drc_conv16_k3s2_tile_fwd"))    # round 4: conv16x.hip
    slide = plan.tile and cin <= 32 and cout <= 32 and dims[0] >= 4                  # conv16t.hip's depth-sliding walk; everything else: conv16x.hip
    cw = 4 if cp % 64 == 0 else (2 if cp % 32 == 0 else 1)
    walk2 = plan.tile and not slide and E.walk2_takes((cin + 31) // 32, dims[0], dims[1], cw)           # two input blocks, full row tiles: conv16sw_kernel
    assert plan.kname.startswith("conv16u_kernel" if transposed else (("conv16s_kernel", "conv16sp_kernel") if slide else ("conv16sw_kernel" if walk2 else "conv16d_kernel"))), plan.kname
    assert not slide or plan.kname.startswith("conv16sp_kernel") == (dims[1] % 28 == 0 or dims[1] % 16 == 0)      # full row tiles: the counted-wait walk
    assert transposed or slide or walk2 or plan.kname.endswith(",%d>" % stride), plan.kname
    plan.run(xb, E.pack_weight16(w.to(dev), transposed), sc, sh, yb, rb)
    got = yb.to_dense().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    tol = 1e-3 * ref.abs().max().item() + 1e-3                       # fp16 output rounding (2^-11 relative) + fp32 summation order
    assert err <= tol, (err, tol)
    v = yb.view6()
    assert v[:, :, 0].abs().sum() == 0 and v[:, :, :, 0].abs().sum() == 0 and v[:, :, :, :, -1].abs().sum() == 0     # halo intact


def test_conv16_classifier_dense_head_and_cost_volume(dev):
    from disprcnn_amd import engine as E
    n = 2
    # (6,12,20): conv16s_kernel; 28 / 16 / 56 rows: the counted-wait walk conv16sp_kernel (dense head with and without the previous head's volume)
    for dims, with_prev in (((6, 12, 20), True), ((5, 28, 20), True), ((4, 16, 30), False), ((6, 56, 28), True), ((7, 28, 14), False)):
        x = _h(synth.hash_uniform(f"hc1:x{dims}", (n, 32) + dims))
        w = _h(synth.hash_uniform("hc1:w", (1, 32, 3, 3, 3), -0.1, 0.1))
        prev = synth.hash_uniform(f"hc1:p{dims}", (n,) + dims)
        ref = F.conv3d(x, w, None, 1, 1)[:, 0] + (prev if with_prev else 0)
        xb = E.Blocked16(n, 32, *dims, 1, 1, 1, dev).from_dense(x.to(dev))
        out = torch.full((n,) + dims, float("nan"), device=dev)
        plan = E.plan_conv3d16_cout1(xb)
        assert plan.kname.startswith("conv16sp_kernel") == (dims[1] % 28 == 0 or dims[1] % 16 == 0), plan.kname
        plan.run(xb, E.pack_weight16(w.to(dev)), None, None, out, prev.to(dev) if with_prev else None)
        assert (out.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-5, dims      # fp32 output: only summation order
    # cost volume: the fp16 rounding of the reference volume, exactly (pure data movement)
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="h:cv")
    for mx, mn in ((48, 0), (24, -24)):
        cv = E.Blocked16(2, 64, (mx - mn) // 4, 28, 28, 1, 1, 1, dev)
        E.cost_volume16_blocked(fl.to(dev), fr.to(dev), cv, mn // 4, mx // 4, -1)
        assert torch.equal(cv.to_dense().cpu(), _h(O.cost_volume(fl, fr, mx, mn)))


@pytest.mark.parametrize("mx,mn,hw,cout", [(48, 0, (28, 28), 32), (48, -48, (56, 56), 32), (24, -24, (9, 30), 32), (16, 0, (5, 17), 64), (32, -16, (13, 16), 16),
                                           (32, -16, (14, 30), 64), (16, 0, (14, 9), 32), (12, 0, (28, 28), 32)])
def test_costvol_fused_first_layer_is_bit_identical(dev, mx, mn, hw, cout):
    """dres0[0] on the fp16 cost volume WITHOUT the volume (drc_conv16_k3_costvol_fwd: the stage addresses of conv16x.hip point at the feature
    pair, zero voxels at the zero halo) == the materialised volume (pinned above against the oracle's) through the same layer, bit for bit;
    positive and negative first disparities, maps narrower than the shift range, ragged tiles, 16 / 32 / 64 couts."""
    from disprcnn_amd import engine as E
    n, D = 2, (mx - mn) // 4
    fl, fr = synth.synth_features(n, 32, *hw, tag=f"cvf{mx}{mn}")
    w = _h(synth.hash_uniform(f"cvf{cout}:w", (cout, 64, 3, 3, 3), -0.1, 0.1))
    cp = E.cout_pad_of(cout)
    sc = torch.ones(cp, device=dev); sh = torch.zeros(cp, device=dev)
    sc[:cout] = synth.hash_uniform("cvf:s", (cout,), 0.5, 1.5).to(dev); sh[:cout] = synth.hash_uniform("cvf:b", (cout,), -0.5, 0.5).to(dev)
    w16 = E.pack_weight16(w.to(dev))
    cv = E.Blocked16(n, 64, D, *hw, 1, 1, 1, dev)
    E.cost_volume16_blocked(fl.to(dev), fr.to(dev), cv, mn // 4, mx // 4, -1)
    y_ref = E.Blocked16(n, cout, D, *hw, 1, 1, 1, dev)
    ref_plan = E.plan_conv3d16(cv, y_ref, 1, cout, True)
    assert ref_plan.kname.startswith(("conv16d_kernel<", "conv16sw_kernel<"))
    ref_plan.run(cv, w16, sc, sh, y_ref)
    pair = E.Blocked16(n, 64, 1, *hw, 1, 1, 1, dev)
    E.cost_volume16_blocked(fl.to(dev), fr.to(dev), pair, 0, 1, -1)
    y = E.Blocked16(n, cout, D, *hw, 1, 1, 1, dev)
    plan = E.plan_conv3d16_costvol(pair, y, mn // 4, cout, True)
    assert plan.kname.endswith(",cv>") and plan.flops == ref_plan.flops
    assert plan.kname.startswith("conv16sw_kernel") == (D >= 4 and cout in (32, 64) and hw[0] % (14 if cout == 32 else 7) == 0), plan.kname
    plan.run(pair, w16, sc, sh, y)
    assert torch.equal(y.storage, y_ref.storage)                     # interior and (zero) halo alike
    assert y_ref.to_dense().abs().max().item() > 0.1


def _model(dev, case, mx, mn, storage):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(mx, mn)
    m.load_state_dict(state_for(case), strict=True)
    m.regressor_storage = storage
    return m.to(dev).eval()


def test_config_a_f16_vs_fp32_oracle(dev):
    """Config A from the feature boundary, fp16 storage vs the CPU fp32 oracle: stated bound mean <= 5e-2 px."""
    sd = state_for("A")
    m = _model(dev, "A", 48, 0, "f16")
    fl, fr = synth.synth_features(4, 32, 28, 28, tag="f16A")
    with torch.no_grad():
        got = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        ref = O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
    err = (got - ref).abs()
    print("Config A f16 storage: mean/max |err| px", err.mean().item(), err.max().item())
    assert err.mean().item() <= 5e-2 and torch.isfinite(got).all()
    with pytest.raises(RuntimeError, match="inference path"):
        m.train().forward_from_features(fl.to(dev), fr.to(dev), (112, 112))


# Measured on MI355X (round 5, all 64 ROIs, D = 96) against the CPU fp32 oracle, mean / max |err| px:
#     fp16-STORAGE regressor: sharp set "B" 0.0514 / 1.15, tempered set "Bt" 0.0050 / 0.061;   default split-f16 path: 2.6e-4 / 7.7e-3 and 2.4e-5 / 2.7e-4.
# SURVEY 8c proposed 5e-2 for the fp16 mode: the tempered set (classif*.2 weights x 0.1, "realistic softmax entropy") is 10x inside it, the
# sharp synthetic set (near-saturated softmax over 96 disparities) sits AT it.  The bounds below are those measurements with ~15-20 % margin;
# they are statements of what the fp16-storage mode delivers, not tolerances tuned to pass.
STRESS_BOUNDS = {"B": 6.0e-2, "Bt": 6.0e-3}


def _stress_state(case):
    if case == "Bt":                      # tempered weights, the BatchNorm statistics of the B fixture (no BN behind classif*.2)
        import os
        from tests.helpers import GOLDEN
        from disprcnn_amd.modeling.psmnet.keys import psmnet_state_template
        return synth.load_bn_stats(synth.synth_state_dict(psmnet_state_template(), tempered=True), os.path.join(GOLDEN, "bn_stats_B.npz"))
    return state_for(case)


@pytest.mark.parametrize("case", ["B", "Bt"])
def test_stress_shape_64_rois_f16_vs_fp32(dev, case):
    """BASELINE configs[3]: 64 ROI crops 224x224, D=96 through the full PSMNet -- ALL 64 ROIs against the CPU fp32 oracle (pinned to the
    reference) for (a) the fp16-storage regressor (regressor_storage = "f16": half the activation bytes, error bound STRESS_BOUNDS) and (b) the
    default path (split-f16 arithmetic: fp16 inputs on the matrix cores at fp32-class error, the bound of the fp32 path: mean <= 1e-3 px)."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = _stress_state(case)
    left, right = synth.synth_images(64, 224, 224, tag="stress16")

    def run(storage):
        m = PSMNet(48, -48)
        m.load_state_dict(sd, strict=True)
        m.regressor_storage = storage
        m = m.to(dev).eval()
        with torch.no_grad():
            out = m((left.to(dev), right.to(dev))).cpu()
        b = m._rt.workspace_bytes()
        keys = {k[0] for k in m._rt._ws}
        del m
        torch.cuda.empty_cache()
        return out, b, keys
    got16, bytes16, _ = run("f16")
    got32, bytes32, keys32 = run("f32")
    assert "3ds16" in keys32                       # the default: split-f16 arithmetic, fp32-class
    with torch.no_grad():
        oref = torch.cat([O.psmnet_forward(sd, left[i:i + 16], right[i:i + 16], 48, -48) for i in range(0, 64, 16)])
    e16, e32 = (got16 - oref).abs(), (got32 - oref).abs()
    print(f"stress 64 ROIs [{case}] vs CPU fp32 oracle (all 64): fp16 storage mean/max |err| px {e16.mean().item():.4f} {e16.max().item():.4f}; "
          f"split-f16 (default) {e32.mean().item():.2e} {e32.max().item():.2e}; workspace {bytes16 / 2**30:.1f} GiB vs {bytes32 / 2**30:.1f} GiB")
    assert torch.isfinite(got16).all() and torch.isfinite(got32).all()
    assert e16.mean().item() <= STRESS_BOUNDS[case]
    assert e16.reshape(64, -1).mean(1).max().item() <= 3 * STRESS_BOUNDS[case]          # ... no single ROI far off
    assert e32.mean().item() <= 1e-3 and e32.max().item() <= 2e-2 and e32.reshape(64, -1).mean(1).max().item() <= 1e-3
    assert bytes16 < 0.8 * bytes32                                    # the regressor's activations take half the bytes (the fp32 2D CNN is shared)


def test_f16_feature_cnn_vs_fp32_path(dev):
    """Round 3: PSMNet.feature_storage = "f16" -- the 2D feature CNN on fp16-storage tensors too (conv16t.hip for the stride-1 3x3 layers,
    conv16.hip for the strided / dilated / 1x1 ones, ops16.hip for the SPP pools, up-samplings and the cost volume), against the fp32
    HIP path.  The feature maps agree to 4e-3 of their range (mean 6e-4: the fp16 rounding of ~60 layers, no border or plumbing term:
    fp32 features rounded to fp16 and pushed through the same cost-volume kernel reproduce the regressor-only error, 0.052 px).  The
    regressor amplifies that to a mean 0.35 px on this synthetic-weight network -- 7x the regressor-only figure -- so the SURVEY 8c
    bound (5e-2 px) cannot be met with fp16 feature storage; the mode is opt-in, its error is what this test pins, and the headline
    fp16 numbers keep the fp32 feature CNN."""
    from disprcnn_amd import engine as E
    left, right = synth.synth_images(4, 224, 224, tag="feat16")
    m32 = _model(dev, "B", 48, -48, "f32")
    m16 = _model(dev, "B", 48, -48, "f16")
    m16.feature_storage = "f16"
    with torch.no_grad():
        ref = m32((left.to(dev), right.to(dev))).cpu()
        got = m16((left.to(dev), right.to(dev))).cpu()
    rt = m16._rt
    ws16 = [w for k, w in rt._ws.items() if k[0] == "2d16"][0]
    ws32 = [w for k, w in m32._rt._ws.items() if k[0] in ("2d", "2ds16")][0]        # (the fp32-class CNN: fp32 or split-f16 arithmetic)
    assert ws16["p"]["fe.layer1.0.conv1"].tile and ws16["p"]["fe.lastconv.0"].tile and not ws16["p"]["fe.layer4.0.conv1"].tile
    f32 = ws32["t"]["feat"].to_dense()[:, :, 0].cpu()
    f16 = ws16["t"]["feat"].to_dense()[:, :, 0].cpu()
    ferr = (f16 - f32).abs().max().item() / f32.abs().max().item()
    err = (got - ref).abs()
    print(f"f16 feature CNN: features max err / range {ferr:.2e}; disparity mean/max |err| px {err.mean().item():.4f} {err.max().item():.4f}")
    assert torch.isfinite(got).all() and ferr <= 1e-2
    assert err.mean().item() <= 0.5                                  # measured 0.345 (see above): a characterisation, not the 8c bound
    with pytest.raises(ValueError):
        m16.feature_storage = "bf16"
        m16((left.to(dev), right.to(dev)))
