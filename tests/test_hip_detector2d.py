"""GPU parity tests of the 2D stage's heads through the C ABI: box decode, Stereo RPN, stereo box head, mask head and
the DispRCNN driver, against the CPU oracle and the fixtures recorded from the imported reference (tests/golden/det_golden.npz).
Tolerances: decode 1e-3 px (expf), objectness 1e-5, conv maps 2e-4 (Winograd fp32 + summation order), FC outputs 2e-3, boxes 5e-3 px,
mask probabilities 5e-4; kept sets identical to the reference's."""
import numpy as np
import pytest
import torch

from oracle import det_oracle as D
from disprcnn_amd.utils import synth
from tests.helpers import check_samples, golden_npz
from tests.test_oracle_det import CASES, POST_NMS, RATIOS, SIZES, STRIDES, det_states

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def z():
    return golden_npz("det_golden.npz")


@pytest.fixture(scope="module")
def model(dev):
    from disprcnn_amd.modeling.detector import default_cfg_2d
    from disprcnn_amd.modeling.roi_heads import build_roi_heads
    from disprcnn_amd.modeling.rpn import build_stereorpn
    cfg = default_cfg_2d("R-50-FPN", post_nms_top_n_test=POST_NMS)
    rpn, heads = build_stereorpn(cfg, 256), build_roi_heads(cfg, 256)
    w_rpn, w_heads = det_states()
    rpn.load_state_dict(w_rpn, strict=True)
    heads.load_state_dict(w_heads, strict=True)
    return rpn.to(dev).eval(), heads.to(dev).eval()


@pytest.mark.parametrize("rows,groups,per,clip", [(1000, 1, 6, None), (777, 2, 4, (320, 160)), (64, 3, 6, (1242, 375)), (0, 2, 4, None), (5, 1, 4, (50, 40))])
def test_box_decode_vs_oracle(dev, rows, groups, per, clip):
    from disprcnn_amd.modeling.box_coder import BoxCoder
    w = (10.0, 10.0, 5.0, 5.0) if per == 4 else (1.0, 1.0, 1.0, 1.0)
    codes = synth.hash_uniform(f"dec{rows}", (rows, groups * per), -3.0, 3.0) * (4.0 if per == 4 else 1.0)
    if rows:
        codes[0, 2] = 50.0                                # exceeds the log(1000/16) clamp
    b = synth.hash_uniform(f"decb{rows}", (rows, 4), 0.0, 1.0)
    boxes = torch.stack([b[:, 0] * 300, b[:, 1] * 150, b[:, 0] * 300 + b[:, 2] * 200, b[:, 1] * 150 + b[:, 3] * 100], 1)
    got = BoxCoder(w).decode(codes.to(dev), boxes.to(dev), clip_to=clip, per=per).cpu()
    assert got.shape == codes.shape
    if rows == 0:
        return
    ref = torch.cat([D.decode(codes[:, g * per:(g + 1) * per], boxes, w) for g in range(groups)], 1)
    if clip is not None:
        ref = ref.reshape(-1, per)
        ref[:, [0, 2] + ([4, 5] if per == 6 else [])] = ref[:, [0, 2] + ([4, 5] if per == 6 else [])].clamp(0, clip[0] - 1)
        ref[:, [1, 3]] = ref[:, [1, 3]].clamp(0, clip[1] - 1)
        ref = ref.reshape(rows, -1)
    scale = ref.abs().clamp(min=1.0)
    assert ((got - ref).abs() / scale).max().item() <= 2e-5          # expf of the device vs libm: a few ulp of a coordinate


@pytest.mark.parametrize("tag,n,h,w", CASES)
def test_stereo_rpn_vs_oracle_and_reference(dev, z, model, tag, n, h, w):
    from disprcnn_amd.structures.image_list import ImageList
    rpn, _ = model
    fl, fr = synth.synth_pyramid(n, h, w, tag="det" + tag)
    gl, gr = [f.to(dev) for f in fl], [f.to(dev) for f in fr]
    images = ImageList(torch.zeros(n, 3, h, w), [(h, w)] * n)
    # raw head maps vs the reference's (objectness is recorded after its pairwise softmax)
    logits, regs = rpn._head(gl, gr)
    for lvl in range(5):
        s = logits[lvl]
        check_samples(z, tag, f"obj{lvl}", s.view(s.shape[0], 2, -1, s.shape[3]).softmax(1).view(*s.shape), 3e-5)
        check_samples(z, tag, f"reg{lvl}", regs[lvl], 3e-4)
    # every anchor's (score, left, right) vs the oracle
    w_rpn, _ = det_states()
    obj, reg = D.srpn_head(fl, fr, w_rpn)
    anchors = D.pyramid_anchors(SIZES, RATIOS, [tuple(f.shape[-2:]) for f in fl], STRIDES)
    sc, left, right = rpn.proposals_dense(images, gl, gr)
    osc = torch.cat([o.permute(0, 2, 3, 1).reshape(n, -1, 2) for o in obj], 1)[:, :, 1]
    org = torch.cat([r.permute(0, 2, 3, 1).reshape(n, -1, 6) for r in reg], 1)
    anc = torch.cat([torch.as_tensor(a, dtype=torch.float32) for a in anchors], 0)
    assert (sc.cpu() - osc).abs().max().item() <= 3e-5
    for i in range(n):
        p = D.decode(org[i], anc, (1.0, 1.0, 1.0, 1.0))
        assert (left[i].cpu() - D.clip(p[:, 0:4], w, h)).abs().max().item() <= 5e-3
        assert (right[i].cpu() - D.clip(p[:, [4, 1, 5, 3]], w, h)).abs().max().item() <= 5e-3
    # final proposals vs the reference
    lp, rp, _ = rpn(images, images, gl, gr)
    for i in range(n):
        assert len(lp[i]) == z[f"{tag}_prop_left{i}"].shape[0] == POST_NMS and lp[i].size == (w, h)
        np.testing.assert_allclose(lp[i].get_field("objectness").cpu().numpy(), z[f"{tag}_prop_score{i}"], rtol=0, atol=3e-5)
        np.testing.assert_allclose(lp[i].bbox.cpu().numpy(), z[f"{tag}_prop_left{i}"], rtol=0, atol=5e-3)
        np.testing.assert_allclose(rp[i].bbox.cpu().numpy(), z[f"{tag}_prop_right{i}"], rtol=0, atol=5e-3)


@pytest.mark.parametrize("tag,n,h,w", CASES)
def test_roi_heads_vs_reference(dev, z, model, tag, n, h, w):
    from disprcnn_amd.structures.bounding_box import BoxList
    _, heads = model
    fl, fr = synth.synth_pyramid(n, h, w, tag="det" + tag)
    gl, gr = [f.to(dev) for f in fl], [f.to(dev) for f in fr]

    def props(side):
        out = []
        for i in range(n):
            b = BoxList(torch.from_numpy(z[f"{tag}_prop_{side}{i}"]).to(dev), (w, h))
            b.add_field("objectness", torch.from_numpy(z[f"{tag}_prop_score{i}"]).to(dev))
            out.append(b)
        return out
    lp, rp = props("left"), props("right")                  # the REFERENCE's proposals: one flipped tie upstream cannot cascade
    x = heads.box.feature_extractor({"left": gl, "right": gr}, {"left": lp, "right": rp})
    logits, deltas = heads.box.predictor(x)
    check_samples(z, tag, "box_x", x, 2e-3)
    np.testing.assert_allclose(logits.cpu().numpy(), z[f"{tag}_box_logits"], rtol=0, atol=3e-3)
    np.testing.assert_allclose(deltas.cpu().numpy(), z[f"{tag}_box_deltas"], rtol=0, atol=3e-3)
    _, ld, rd, _ = heads(gl, gr, lp, rp)
    for i in range(n):
        assert ld[i].bbox.shape == z[f"{tag}_det_left{i}"].shape, "kept set differs from the reference's"
        np.testing.assert_array_equal(ld[i].get_field("labels").cpu().numpy(), z[f"{tag}_det_label{i}"])
        np.testing.assert_allclose(ld[i].get_field("scores").cpu().numpy(), z[f"{tag}_det_score{i}"], rtol=0, atol=1e-3)
        np.testing.assert_allclose(ld[i].bbox.cpu().numpy(), z[f"{tag}_det_left{i}"], rtol=0, atol=5e-2)
        np.testing.assert_allclose(rd[i].bbox.cpu().numpy(), z[f"{tag}_det_right{i}"], rtol=0, atol=5e-2)
        m = ld[i].get_field("mask")
        assert tuple(m.shape) == tuple(z[f"{tag}_det_mask_shape{i}"])
    # mask head alone on the reference's detections (tight)
    dets = []
    for i in range(n):
        b = BoxList(torch.from_numpy(z[f"{tag}_det_left{i}"]).to(dev), (w, h))
        b.add_field("labels", torch.from_numpy(z[f"{tag}_det_label{i}"]).to(dev))
        dets.append(b)
    _, md, _ = heads.mask(gl, dets)
    for i in range(n):
        check_samples(z, tag, f"det_mask{i}", md[i].get_field("mask"), 5e-4)


def test_roi_heads_without_detections(dev, z, model):
    """No proposal passes the score threshold: empty BoxLists with an empty [0,1,28,28] mask field, no launch on empty batches."""
    from disprcnn_amd.structures.bounding_box import BoxList
    _, heads = model
    tag, n, h, w = CASES[1]
    fl, fr = synth.synth_pyramid(n, h, w, tag="det" + tag)
    gl, gr = [f.to(dev) for f in fl], [f.to(dev) for f in fr]
    lp = [BoxList(torch.from_numpy(z[f"{tag}_prop_left{i}"]).to(dev), (w, h)) for i in range(n)]
    rp = [BoxList(torch.from_numpy(z[f"{tag}_prop_right{i}"]).to(dev), (w, h)) for i in range(n)]
    saved = heads.box.post_processor.score_thresh
    heads.box.post_processor.score_thresh = 2.0
    try:
        _, ld, rd, _ = heads(gl, gr, lp, rp)
    finally:
        heads.box.post_processor.score_thresh = saved
    for i in range(n):
        assert len(ld[i]) == 0 and len(rd[i]) == 0
        assert tuple(ld[i].get_field("mask").shape) == (0, 1, 28, 28) and ld[i].get_field("labels").dtype == torch.int64
    # and an image without proposals at all
    x = heads.box.feature_extractor({"left": gl, "right": gr}, {"left": [lp[0][:0]], "right": [rp[0][:0]]})
    assert tuple(x.shape) == (0, 2048)


def test_disprcnn_end_to_end_vs_oracle(dev):
    """The whole 2D stage (R-50-FPN trunk + Stereo RPN + heads) on a small stereo pair: the heads' outputs are checked against the
    oracle fed with the product's own pyramid (the trunk has its own golden tests, tests/test_backbone.py)."""
    from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
    n, h, w = 2, 160, 256
    m = DispRCNN(default_cfg_2d("R-50-FPN", post_nms_top_n_test=40))
    sd = m.state_dict()
    t_rpn = {k[4:]: v for k, v in sd.items() if k.startswith("rpn.")}
    t_heads = {k[10:]: v for k, v in sd.items() if k.startswith("roi_heads.")}
    w_rpn, w_heads = synth.synth_det_state(t_rpn, gain=synth.DET_GAIN), synth.synth_det_state(t_heads, gain=synth.DET_GAIN)
    bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
    m.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **{"rpn." + k: v for k, v in w_rpn.items()},
                       **{"roi_heads." + k: v for k, v in w_heads.items()}}, strict=True)
    m = m.to(dev).eval()
    left, right = synth.synth_images(n, h, w, tag="e2e2d")
    with torch.no_grad():
        out = m({"left": left.to(dev), "right": right.to(dev)})
        feats = m.backbone(torch.cat((left, right), 0).to(dev))
    assert set(out) == {"left", "right"} and len(out["left"]) == n
    fl, fr = [f[:n].cpu() for f in feats], [f[n:].cpu() for f in feats]
    obj, reg = D.srpn_head(fl, fr, w_rpn)
    anchors = D.pyramid_anchors(SIZES, RATIOS, [tuple(f.shape[-2:]) for f in fl], STRIDES)
    props = D.srpn_select(anchors, obj, reg, [(w, h)] * n, 6000, 40, 0.7, 0)
    pl, pr = [p[0] for p in props], [p[1] for p in props]
    _, logits, deltas = D.box_head(fl, fr, pl, pr, h, w_heads)
    dets = D.box_post(logits, deltas, pl, pr, [(w, h)] * n)
    for i in range(n):
        ld, rd = out["left"][i], out["right"][i]
        assert ld.size == (w, h) and set(ld.fields()) >= {"scores", "labels", "mask"}
        assert len(ld) == len(dets[i]["scores"]) == len(rd), (len(ld), len(dets[i]["scores"]))
        # same detections up to order (ties in NMS bookkeeping aside): compare sorted by score
        o1, o2 = torch.argsort(ld.get_field("scores").cpu(), descending=True), torch.argsort(dets[i]["scores"], descending=True)
        assert (ld.get_field("scores").cpu()[o1] - dets[i]["scores"][o2]).abs().max().item() <= 2e-3
        assert (ld.bbox.cpu()[o1] - dets[i]["left"][o2]).abs().max().item() <= 0.1
        assert (rd.bbox.cpu()[o1] - dets[i]["right"][o2]).abs().max().item() <= 0.1
        assert tuple(ld.get_field("mask").shape) == (len(ld), 1, 28, 28)


@pytest.mark.parametrize("n,h,w", [(1, 160, 256), (2, 96, 160), (1, 150, 250)])
def test_stereo_rpn_blocked_head_matches_dense_head(dev, n, h, w):
    """The Stereo-RPN head run on the backbone's blocked pyramid (no layout round trips, the two predictors as one 1x1 convolution)
    gives the maps of the dense path (srpn.py:27-50), for one image pair (one launch over [left, right]) and for a batch."""
    from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
    m = DispRCNN(default_cfg_2d("R-50-FPN", post_nms_top_n_test=40))
    sd = m.state_dict()
    w_rpn = synth.synth_det_state({k[4:]: v for k, v in sd.items() if k.startswith("rpn.")}, gain=synth.DET_GAIN)
    bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
    m.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **{"rpn." + k: v for k, v in w_rpn.items()}}, strict=False)
    m = m.to(dev).eval()
    left, right = synth.synth_images(n, h, w, tag="blkhead")
    with torch.no_grad():
        feats = m.backbone(torch.cat((left, right), 0).to(dev))
        levels = m.backbone._rt.blocked_levels()
        assert [lv.N for lv in levels] == [2 * n] * 5 and all(lv.ph == 1 for lv in levels)
        lg_d, rg_d = m.rpn._head([f[:n] for f in feats], [f[n:] for f in feats])
        lg_b, rg_b = m.rpn._head_blocked(levels)
        lg_b2, _ = m.rpn._head_blocked(levels)                       # cached workspace
    for d, b, b2 in zip(lg_d, lg_b, lg_b2):
        assert d.shape == b.shape and (d - b).abs().max().item() <= 1e-5 * max(1.0, d.abs().max().item())
        assert torch.equal(b, b2)
    for d, b in zip(rg_d, rg_b):
        assert d.shape == b.shape and (d - b).abs().max().item() <= 1e-5 * max(1.0, d.abs().max().item())


def test_pooler_single_launch_equals_per_level_loop(dev):
    """drc_roi_align_fpn_fwd (one launch over the pyramid, round 3) against the per-level nonzero / index_select / ROIAlign / index_copy
    loop of the reference's Pooler (poolers.py:118-149): bit-identical, for both pooler shapes of the 2D stage, incl. empty levels."""
    from disprcnn_amd.modeling.poolers import Pooler
    from disprcnn_amd.structures.bounding_box import BoxList
    fl, _ = synth.synth_pyramid(2, 160, 256, tag="poolfpn")
    feats = [f.to(dev) for f in fl]
    g = torch.Generator().manual_seed(5)
    boxes = []
    for i, r in enumerate((37, 0 + 5)):
        xy = torch.rand(r, 2, generator=g) * torch.tensor([200.0, 120.0])
        wh = torch.rand(r, 2, generator=g) ** 2 * torch.tensor([250.0, 150.0]) + 2.0
        b = torch.cat((xy, (xy + wh).clamp(max=255.0)), 1)
        b[:, 3] = b[:, 3].clamp(max=159.0)
        boxes.append(BoxList(b.to(dev), (256, 160)))
    for res, ratio in ((7, 2), (14, 2)):
        p = Pooler((res, res), (0.25, 0.125, 0.0625, 0.03125), ratio)
        one = p(feats, boxes)
        p.single_launch = False
        ref = p(feats, boxes)
        assert one.shape == ref.shape == (42, 256, res, res) and torch.equal(one, ref)
    small = [BoxList(torch.tensor([[3.0, 4.0, 20.0, 18.0]], device=dev), (256, 160)), BoxList(torch.zeros(0, 4, device=dev), (256, 160))]
    p.single_launch = True
    a = p(feats, small)
    p.single_launch = False
    assert torch.equal(a, p(feats, small))


def test_full_pipeline_images_to_disparity_maps(dev):
    """test_net.py's data flow on the HIP path: stereo pair -> DispRCNN (2D stage) -> DispRCNN3D (instance disparity on the detections)
    -> DisparityMapProcessor (full-image maps).  The 2D stage's output IS the disparity stage's lr_result; each stage has its own parity
    tests, here the hand-over is checked: fields, pairing, shapes, and the disparity stage against its oracle on the same detections."""
    from oracle import psmnet_oracle as O, roi_oracle as R
    from disprcnn_amd.modeling.detector import DispRCNN, DispRCNN3D, default_cfg_2d
    from disprcnn_amd.modeling.detector.disprcnn3d import default_cfg
    from disprcnn_amd.modeling.psmnet.inference import DisparityMapProcessor
    from disprcnn_amd.structures import ImageList
    from tests.helpers import state_for
    n, h, w = 1, 192, 384
    m2 = DispRCNN(default_cfg_2d("R-50-FPN", post_nms_top_n_test=30))
    sd = m2.state_dict()
    heads = synth.synth_det_state({k: v for k, v in sd.items() if not k.startswith("backbone.")},
                                  gain={("rpn." if k.startswith("head.") else "roi_heads.") + k: v for k, v in synth.DET_GAIN.items()})
    bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
    m2.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **heads}, strict=True)
    m2 = m2.to(dev).eval()
    m3 = DispRCNN3D(default_cfg(48, -48, 224))
    m3.dispnet.load_state_dict(state_for("B"), strict=True)
    m3 = m3.to(dev).eval()
    # raw [0,1) images: DispRCNN3D normalises its crops itself (disprcnn3d.py:44-50), so PSMNet sees inputs in its calibrated range
    left = synth.hash_uniform("pipe:L", (n, 3, h, w), 0.0, 1.0)
    right = torch.roll(left, shifts=-5, dims=3) * 0.9 + 0.1 * synth.hash_uniform("pipe:R", (n, 3, h, w), 0.0, 1.0)
    left, right = left.to(dev), right.to(dev)
    with torch.no_grad():
        det = m2({"left": left, "right": right})
        for side in ("left", "right"):                       # keep the six best detections: the CPU oracle below runs PSMNet per ROI
            det[side] = [b[torch.argsort(det["left"][i].get_field("scores"), descending=True)[:6]] for i, b in enumerate(det[side])]
        out = m3({"left": ImageList(left, [(h, w)] * n), "right": ImageList(right, [(h, w)] * n)}, det)
        maps = DisparityMapProcessor()(out["left"], out["right"])
    maps = maps if isinstance(maps, list) else [maps]
    for i in range(n):
        ld, rd = out["left"][i], out["right"][i]
        assert len(ld) == len(rd) > 0 and set(ld.fields()) >= {"scores", "labels", "mask", "disparity"}
        assert tuple(ld.get_field("disparity").shape) == (len(ld), 224, 224) and torch.isfinite(ld.get_field("disparity")).all()
        assert tuple(maps[i].data.shape) == (h, w) and torch.isfinite(maps[i].data).all()
    # the disparity stage vs its oracle on the detections the 2D stage produced (image 0)
    ld, rd = out["left"][0], out["right"][0]
    img_l, img_r = left[0].cpu().numpy(), right[0].cpu().numpy()
    rois_l, rois_r = [], []
    for lb, rb in zip(ld.bbox.cpu().tolist(), rd.bbox.cpu().tolist()):
        x1, y1, x1p, y2, mw = R.align_roi_pair(lb, rb, w, h)
        rois_l.append([0, x1, y1, x1 + mw, y2]); rois_r.append([0, x1p, y1, x1p + mw, y2])
    import numpy as np
    cl = torch.from_numpy(R.crop_and_normalise(img_l[None], np.asarray(rois_l, dtype=np.float32), 224))
    cr = torch.from_numpy(R.crop_and_normalise(img_r[None], np.asarray(rois_r, dtype=np.float32), 224))
    ref = O.psmnet_forward(state_for("B"), cl, cr, 48, -48)
    err = (ld.get_field("disparity").cpu() - ref).abs()
    assert err.mean().item() <= 2e-3 and err.max().item() <= 1e-1, (err.mean().item(), err.max().item())     # crops of ~8 px wide boxes


def test_engine_conv2d_storage_flat_over_unit_counts(dev):
    """The mask head's convolutions see a different detection count on every image: EngineConv2d owns its blocked workspaces
    once per map geometry, sized for a capacity bucket, and runs smaller counts on prefix views -- results equal conv2d at every count,
    and the cached storage stops growing once the largest count was seen."""
    import torch.nn.functional as F
    from disprcnn_amd.modeling.head_ops import EngineConv2d
    conv = torch.nn.Conv2d(24, 40, 3, padding=1)
    conv.weight.data = synth.hash_uniform("ec2d:w", (40, 24, 3, 3), -0.2, 0.2)
    conv.bias.data = synth.hash_uniform("ec2d:b", (40,), -0.5, 0.5)
    ec = EngineConv2d(conv.to(dev), relu=True)
    sizes = []
    for n in (100, 7, 63, 1, 100, 33, 96, 2):
        x = synth.hash_uniform(f"ec2d:x{n}", (n, 24, 14, 14))
        got = ec(x.to(dev)).cpu()
        ref = F.relu(F.conv2d(x, conv.weight.cpu(), conv.bias.cpu(), padding=1))
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-5, n
        sizes.append(ec.nbytes())
    assert len(set(sizes)) == 1, sizes                                   # one allocation, at the first (largest) count
    assert tuple(ec(torch.zeros(0, 24, 14, 14, device=dev)).shape) == (0, 40, 14, 14)


@pytest.mark.parametrize("M,N,K,relu,bias", [(300, 2048, 25088, True, True), (65, 2048, 2048, True, True), (7, 12, 2048, False, True),
                                              (3, 5, 20, False, False), (130, 70, 36, True, True), (0, 8, 16, False, True)])
def test_linear_mfma_gemm_vs_reference(dev, M, N, K, relu, bias):
    """head_ops.linear = drc_linear_fwd (csrc/linear.hip), the hand-written fp32-MFMA GEMM behind the stereo box head's fully connected
    layers (roi_box_feature_extractors.py:85-130: 25088 -> 2048 -> 2048), its predictors and the mask predictor: vs a float64 matmul, incl.
    the split-K path, ragged M / N tiles, a K that is not a multiple of 16, no bias, and an empty ROI batch."""
    from disprcnn_amd.modeling.head_ops import linear
    lin = torch.nn.Linear(K, N, bias=bias)
    lin.weight.data = synth.hash_uniform(f"lin{M}{N}{K}:w", (N, K), -1.0, 1.0) / (K ** 0.5)
    if bias:
        lin.bias.data = synth.hash_uniform(f"lin{N}:b", (N,), -0.5, 0.5)
    x = synth.hash_uniform(f"lin{M}{K}:x", (M, K), -1.0, 1.0)
    got = linear(x.to(dev), lin.to(dev), relu=relu).cpu()
    ref = x.double() @ lin.weight.detach().cpu().double().t()
    if bias:
        ref = ref + lin.bias.detach().cpu().double()
    ref = torch.relu(ref) if relu else ref
    assert got.shape == (M, N) and got.dtype == torch.float32
    if M:
        assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
        again = linear(x.to(dev), lin, relu=relu).cpu()
        assert torch.equal(got, again)                       # split-K partials are added in a fixed order
