"""GPU parity tests for the rows around the path: ROIAlign crop (f1), ROI pairing + DispRCNN3D caller (a11), losses (a10)."""
import numpy as np
import pytest
import torch

from oracle import psmnet_oracle as O
from oracle import roi_oracle as R
from disprcnn_amd.utils import synth
from tests.helpers import state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_roi_align_vs_oracle(dev):
    from disprcnn_amd.layers import ROIAlign
    img = synth.hash_uniform("roi:img", (2, 3, 37, 53), 0.0, 1.0)
    rois = torch.tensor([[0, 3.2, 4.7, 30.9, 28.1], [1, 10.0, 5.0, 24.0, 33.0], [0, -4.0, -3.0, 20.0, 12.0],
                         [1, 40.0, 20.0, 60.0, 45.0], [0, 8.0, 8.0, 6.0, 7.0], [1, 0.0, 0.0, 52.0, 36.0]])
    for (ph, pw, sr, scale) in [(7, 7, 2, 1.0), (6, 5, 0, 1.0), (14, 14, 0, 0.5), (3, 4, 1, 0.25)]:
        ref = R.roi_align(img.numpy(), rois.numpy(), scale, ph, pw, sr)
        got = ROIAlign((ph, pw), scale, sr)(img.to(dev), rois.to(dev)).cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-6, (ph, pw, sr, np.abs(got - ref).max())
    empty = ROIAlign((7, 7), 1.0, 2)(img.to(dev), torch.zeros(0, 5, device=dev))
    assert tuple(empty.shape) == (0, 3, 7, 7)


def test_roi_align_vs_reference_kernel_golden(dev):
    """tests/golden/roi_golden.npz holds outputs of the reference's own CPU kernel (csrc/cpu/ROIAlign_cpu.cpp, built by
    oracle/build_ref.py).  Same float32 operation order + no FMA contraction in the build => the HIP op is held to them BIT
    FOR BIT: the small cases (4 pooled-size / sampling settings) and the caller's 224x224 crops of a 375x1242 pair for 13 roi
    geometries (ped / cyclist sizes, >224-px sides, out-of-image, malformed)."""
    import hashlib
    import os
    from disprcnn_amd.layers.roi_align import roi_align_forward
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roi_golden.npz"), allow_pickle=False)
    img = synth.hash_uniform("roi:img", (2, 3, 37, 53), 0.0, 1.0).to(dev)
    rois = torch.from_numpy(z["small_rois"]).to(dev)
    for k, (ph, pw, sr, scale) in enumerate(z["small_settings"]):
        got = roi_align_forward(img, rois, float(scale), int(ph), int(pw), int(sr)).cpu().numpy()
        assert np.array_equal(got, z[f"small_out{k}"]), (k, np.abs(got - z[f"small_out{k}"]).max())
    pair = synth.hash_uniform("roi:pair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)
    got = roi_align_forward(pair, torch.from_numpy(z["crop_rois"]).to(dev), 1.0, 224, 224, 0).cpu().numpy()
    sample_err = np.abs(got.reshape(-1)[z["crop_idx"]] - z["crop_val"]).max()
    assert sample_err == 0.0, sample_err
    sha = [hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() for k in range(len(got))]
    assert sha == [str(s_) for s_ in z["crop_roi_sha"]]
    # 4-channel and 1-channel groupings (FPN-style inputs) against the oracle restatement, which is pinned to the same kernel
    for C_ in (4, 5, 8):
        x = synth.hash_uniform(f"roi:c{C_}", (2, C_, 37, 53), 0.0, 1.0)
        ref = R.roi_align(x.numpy(), z["small_rois"], 0.5, 7, 7, 2)
        assert np.array_equal(roi_align_forward(x.to(dev), rois, 0.5, 7, 7, 2).cpu().numpy(), ref)
    # fused normalisation (disprcnn3d.py:44-50): (crop - mean) / std in float32 on the bit-exact crop
    mean, std = torch.tensor([0.485, 0.456, 0.406], device=dev), torch.tensor([0.229, 0.224, 0.225], device=dev)
    gotn = roi_align_forward(pair, torch.from_numpy(z["crop_rois"][:3]).to(dev), 1.0, 224, 224, 0, mean, std).cpu().numpy()
    refn = (got[:3] - R.MEAN[None, :, None, None]) / R.STD[None, :, None, None]
    assert np.abs(gotn - refn).max() < 1e-6


def test_roi_align_backward_is_adjoint(dev):
    """<roi_align(x), g> == <x, roi_align_backward(g)> (the forward is linear in x)."""
    from disprcnn_amd.layers.roi_align import roi_align
    x = synth.hash_uniform("roi:x", (2, 2, 20, 24)).to(dev).requires_grad_(True)
    rois = torch.tensor([[0, 2.5, 3.5, 17.0, 15.2], [1, 0.0, 1.0, 23.0, 19.0]], device=dev)
    y = roi_align(x, rois, (5, 6), 1.0, 0)
    g = synth.hash_uniform("roi:g", tuple(y.shape)).to(dev)
    y.backward(g)
    lhs = (y.detach() * g).sum().item()
    rhs = (x.detach() * x.grad).sum().item()
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_align_roi_pairs_and_crops_vs_oracle(dev):
    from disprcnn_amd.modeling.detector.disprcnn3d import DispRCNN3D, default_cfg
    from disprcnn_amd.structures import BoxList, ImageList
    W, H, res = 160, 96, 32
    model = DispRCNN3D(default_cfg(resolution=res)).to(dev).eval()
    limg = synth.hash_uniform("cal:L", (2, 3, H, W), 0.0, 1.0)
    rimg = synth.hash_uniform("cal:R", (2, 3, H, W), 0.0, 1.0)
    lboxes = [torch.tensor([[10.3, 5.8, 50.2, 40.1], [-3.0, -2.0, 170.0, 120.0]]), torch.tensor([[100.5, 20.2, 158.9, 90.7]])]
    rboxes = [torch.tensor([[2.9, 6.0, 44.5, 40.0], [120.5, 0.0, 159.9, 99.0]]), torch.tensor([[90.1, 20.0, 140.0, 91.0]])]
    lres = [BoxList(b, (W, H)) for b in lboxes]
    rres = [BoxList(b, (W, H)) for b in rboxes]
    left, right, geom = model.prepare_psmnet_input(ImageList(limg.to(dev), [(H, W)] * 2), ImageList(rimg.to(dev), [(H, W)] * 2), lres, rres)
    rois_l, rois_r, gexp = [], [], []
    for i, (lb, rb) in enumerate(zip(lboxes, rboxes)):
        for l, r in zip(lb.tolist(), rb.tolist()):
            x1, y1, x1p, y2, mw = R.align_roi_pair(l, r, W, H)
            rois_l.append([i, x1, y1, x1 + mw, y2]); rois_r.append([i, x1p, y1, x1p + mw, y2]); gexp.append([x1, x1p, x1 + mw, x1p + mw])
    assert geom.cpu().tolist() == gexp
    ref_l = R.crop_and_normalise(limg.numpy(), np.array(rois_l, dtype=np.float32), res)
    ref_r = R.crop_and_normalise(rimg.numpy(), np.array(rois_r, dtype=np.float32), res)
    assert np.abs(left.cpu().numpy() - ref_l).max() < 2e-5 and np.abs(right.cpu().numpy() - ref_r).max() < 2e-5


def test_disprcnn3d_eval_end_to_end(dev):
    """images + detections -> 'disparity' field, vs the oracle pipeline (crop oracle -> PSMNet oracle)."""
    from disprcnn_amd.modeling.detector import build_detection_model
    from disprcnn_amd.modeling.detector.disprcnn3d import default_cfg
    from disprcnn_amd.structures import BoxList, ImageList
    W, H, res = 320, 256, 224
    model = build_detection_model(default_cfg(48, -48, res))
    sd = state_for("B")
    model.dispnet.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    # smooth images (coarse noise upsampled) so the crops look like image content rather than white noise
    base_l = synth.hash_uniform("e2e:L", (1, 3, H // 8, W // 8), 0.0, 1.0)
    limg = torch.nn.functional.interpolate(base_l, (H, W), mode="bilinear", align_corners=True)
    rimg = torch.roll(limg, -5, 3)
    lb = torch.tensor([[20.4, 10.2, 200.7, 180.3], [0.5, 0.5, 1.2, 1.4]])      # second box is illegal (w,h <= 1): removed
    rb = torch.tensor([[14.9, 10.0, 195.2, 181.0], [0.5, 0.5, 1.2, 1.4]])
    lres, rres = [BoxList(lb, (W, H))], [BoxList(rb, (W, H))]
    with torch.no_grad():
        out = model({"left": ImageList(limg.to(dev), [(H, W)]), "right": ImageList(rimg.to(dev), [(H, W)])}, {"left": lres, "right": rres})
    disp = out["left"][0].get_field("disparity").cpu()
    assert tuple(disp.shape) == (1, res, res) and len(out["left"][0]) == 1
    x1, y1, x1p, y2, mw = R.align_roi_pair(lb[0].tolist(), rb[0].tolist(), W, H)
    assert out["left"][0].get_field("roi_geom").cpu().tolist() == [[x1, x1p, x1 + mw, x1p + mw]]
    cl = torch.from_numpy(R.crop_and_normalise(limg.numpy(), np.array([[0, x1, y1, x1 + mw, y2]], dtype=np.float32), res))
    cr = torch.from_numpy(R.crop_and_normalise(rimg.numpy(), np.array([[0, x1p, y1, x1p + mw, y2]], dtype=np.float32), res))
    with torch.no_grad():
        ref = O.psmnet_forward(sd, cl, cr, 48, -48)
    err = (disp - ref).abs()
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (err.mean().item(), err.max().item())
    # empty detections -> empty field, no launch (reference disprcnn3d.py:272-275)
    out = model({"left": ImageList(limg.to(dev), [(H, W)]), "right": ImageList(rimg.to(dev), [(H, W)])},
                {"left": [BoxList(torch.zeros(0, 4), (W, H))], "right": [BoxList(torch.zeros(0, 4), (W, H))]})
    assert tuple(out["left"][0].get_field("disparity").shape) == (0, res, res)


def test_psm_loss_forward_backward_vs_oracle(dev):
    from disprcnn_amd.utils.loss_utils import PSMLoss, EndPointErrorLoss
    shape = (3, 40, 56)
    tgt = synth.hash_uniform("loss:t", shape, -48.0, 48.0)
    mask = (synth.hash_uniform("loss:m", shape, 0.0, 1.0) > 0.4).to(torch.uint8)
    preds = [tgt + synth.hash_uniform(f"loss:p{k}", shape, -3.0, 3.0) for k in range(3)]
    ref_in = [p.clone().requires_grad_(True) for p in preds]
    ref = O.psm_loss(ref_in, tgt, mask)
    ref.backward()
    got_in = [p.to(dev).requires_grad_(True) for p in preds]
    got = PSMLoss()(tuple(got_in), {"disparity": tgt.to(dev), "mask": mask.to(dev)})
    (got * 2.0).backward()
    assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    for a, b in zip(got_in, ref_in):
        assert (a.grad.cpu() - 2.0 * b.grad).abs().max().item() < 1e-7
    # eval form (EPE) and the empty-mask conventions (train: no division; eval: 0)
    epe = EndPointErrorLoss()(tgt.to(dev), preds[0].to(dev), mask.to(dev))
    assert abs(epe.item() - O.psm_loss(preds[0], tgt, mask).item()) < 1e-5
    zero = torch.zeros(shape, dtype=torch.uint8, device=dev)
    assert EndPointErrorLoss()(tgt.to(dev), preds[0].to(dev), zero).item() == 0.0
    assert PSMLoss()(tuple(p.to(dev) for p in preds), {"disparity": tgt.to(dev), "mask": zero}).item() == 0.0


def test_configs4_mixed_roi_sizes_multi_image_end_to_end(dev):
    """BASELINE configs[4] (pedestrian + cyclist: mixed ROI sizes w in [15,120], h in [40,250], a different ROI count on every
    image incl. none, per-ROI disparity offset x1-x1p and scale mw/224): images + paired detections -> DispRCNN3D ->
    'disparity' fields -> DisparityMapProcessor -> one full-image map per image, against the oracle pipeline (crop oracle pinned
    to the reference's ROIAlign kernel -> PSMNet oracle pinned to the reference's outputs -> post oracle pinned to the
    reference's DisparityMapProcessor).  ROI heights > 224 px exercise the 2-sample ROIAlign grid."""
    from oracle import post_oracle as P
    from disprcnn_amd.modeling.detector import build_detection_model
    from disprcnn_amd.modeling.detector.disprcnn3d import default_cfg
    from disprcnn_amd.modeling.psmnet.inference import DisparityMapProcessor
    from disprcnn_amd.structures import BoxList, ImageList
    W, H, res = 640, 300, 224
    model = build_detection_model(default_cfg(48, -48, res))
    sd = state_for("B")
    model.dispnet.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    base = synth.hash_uniform("c4:L", (3, 3, H // 6, W // 8), 0.0, 1.0)
    limg = torch.nn.functional.interpolate(base, (H, W), mode="bilinear", align_corners=True)
    rimg = torch.roll(limg, -7, 3)
    lboxes = [torch.tensor([[100.3, 20.6, 135.8, 110.2],        # pedestrian 36 x 90
                            [300.0, 10.0, 420.0, 260.0],        # cyclist 120 x 250 (> 224 rows: 2 samples along y)
                            [500.7, 130.2, 515.9, 171.4]]),     # far pedestrian 15 x 41
              torch.zeros(0, 4),                                 # an image without detections
              torch.tensor([[10.0, 40.0, 70.0, 180.0], [600.4, 50.0, 639.9, 299.9], [200.0, 100.0, 290.0, 200.0],
                            [330.5, 60.5, 352.0, 118.0]])]
    shifts = [[6.2, 14.7, 2.1], [], [30.0, 9.5, 21.3, 4.0]]
    rboxes = []
    for lb, sh in zip(lboxes, shifts):
        rb = lb.clone()
        if len(sh):
            rb[:, [0, 2]] -= torch.tensor(sh)[:, None]
            rb[:, 2] += torch.tensor([1.5, -3.0, 0.0, 2.0][: len(sh)])          # right boxes of another width
        rboxes.append(rb)
    lres = [BoxList(b, (W, H)) for b in lboxes]
    rres = [BoxList(b, (W, H)) for b in rboxes]
    sizes = [(H, W)] * 3
    with torch.no_grad():
        out = model({"left": ImageList(limg.to(dev), sizes), "right": ImageList(rimg.to(dev), sizes)}, {"left": lres, "right": rres})
    assert [len(b) for b in out["left"]] == [3, 0, 4]
    maps = DisparityMapProcessor()(out["left"], out["right"])
    assert len(maps) == 3
    for i in range(3):
        disp = out["left"][i].get_field("disparity").cpu()
        assert tuple(disp.shape) == (len(lboxes[i]), res, res)
        if len(lboxes[i]) == 0:
            assert maps[i].data.abs().sum().item() == 0
            continue
        rois_l, rois_r, geom = [], [], []
        for l, r in zip(lboxes[i].tolist(), rboxes[i].tolist()):
            x1, y1, x1p, y2, mw = R.align_roi_pair(l, r, W, H)
            rois_l.append([i, x1, y1, x1 + mw, y2]); rois_r.append([i, x1p, y1, x1p + mw, y2]); geom.append([x1, x1p, x1 + mw, x1p + mw])
        assert out["left"][i].get_field("roi_geom").cpu().tolist() == geom
        cl = torch.from_numpy(R.crop_and_normalise(limg.numpy(), np.array(rois_l, dtype=np.float32), res))
        cr = torch.from_numpy(R.crop_and_normalise(rimg.numpy(), np.array(rois_r, dtype=np.float32), res))
        with torch.no_grad():
            ref = O.psmnet_forward(sd, cl, cr, 48, -48)
        err = (disp - ref).abs()
        print(f"configs[4] image {i}: {len(geom)} rois, mean/max err px {err.mean().item():.3e} {err.max().item():.3e}")
        assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (i, err.mean().item(), err.max().item())
        refmap = P.disparity_map(lboxes[i], rboxes[i], ref, H, W)
        merr = (maps[i].data.cpu() - refmap).abs().max().item()
        assert merr < 2e-2 * max(W / res, 1.0) + 1e-3, (i, merr)               # per-ROI error scaled by the resize factor mw/224
