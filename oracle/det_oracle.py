"""CPU ORACLE (test infrastructure) for the 2D detection stage around the disparity path: Stereo RPN, stereo box
head, mask head -- inference only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restated from the reference (file:line):
  pyramid anchors                modeling/rpn/anchor_generator.py:300-357 (generate_anchors_all_pyramids / _single_pyramid)
  cell anchors (state_dict)      modeling/rpn/anchor_generator.py:222-298
  BoxCoder.decode (4 / 6 codes)  modeling/box_coder.py:161-244
  SRPNHead                       modeling/rpn/stereo_rpn/srpn.py:14-50
  SRPNPostProcessor.forward      modeling/rpn/stereo_rpn/inference.py:121-196, clip_boxes :287-299
  double_view_boxlist_nms        structures/boxlist_ops.py:36-79
  LevelMapper / Pooler           modeling/poolers.py:10-40, 87-128
  StereoFPN2MLPFeatureExtractor  modeling/roi_heads/box_head/roi_box_feature_extractors.py:85-120
  StereoFPNPredictor             modeling/roi_heads/box_head/roi_box_predictors.py:61-83
  PostProcessor (double view)    modeling/roi_heads/box_head/inference.py:84-120, 213-263
  MaskRCNNFPNFeatureExtractor    modeling/roi_heads/mask_head/roi_mask_feature_extractors.py:16-64
  MaskRCNNC4Predictor            modeling/roi_heads/mask_head/roi_mask_predictors.py:9-29
  MaskPostProcessor              modeling/roi_heads/mask_head/inference.py:27-60
float32 torch-CPU arithmetic like the reference's own CPU path; ROIAlign and NMS come from the pinned roi_oracle / nms_oracle.
PINNED by tests/golden/det_golden.npz, recorded from the imported reference modules (tests/golden/make_golden_det.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nms_oracle, roi_oracle

XFORM_CLIP = math.log(1000.0 / 16)


# ------------------------------------------------------------------------------------------------ anchors
def pyramid_anchors(sizes, ratios, feature_shapes, strides):
    """One float64 [H*W*A, 4] array per level, position-major / ratio-minor, xyxy around the cell ORIGIN (no +stride/2)."""
    out = []
    for size, (h, w), stride in zip(sizes, feature_shapes, strides):
        r = np.asarray(ratios, dtype=np.float64)
        hs, ws = size / np.sqrt(r), size * np.sqrt(r)
        ys, xs = np.arange(h) * stride, np.arange(w) * stride
        cx = np.repeat(np.tile(xs, h), len(r)).astype(np.float64)
        cy = np.repeat(np.repeat(ys, w), len(r)).astype(np.float64)
        bw, bh = np.tile(ws, h * w), np.tile(hs, h * w)
        out.append(np.stack([cx - 0.5 * bw, cy - 0.5 * bh, cx + 0.5 * bw, cy + 0.5 * bh], axis=1))
    return out


def cell_anchors(stride, sizes, ratios):
    """The classic Faster R-CNN cell anchors kept as state_dict buffers (anchor_generator.cell_anchors.N), float64 [A,4]."""
    base = np.array([0, 0, stride - 1, stride - 1], dtype=np.float64)
    w, h = base[2] - base[0] + 1, base[3] - base[1] + 1
    cx, cy = base[0] + 0.5 * (w - 1), base[1] + 0.5 * (h - 1)
    rows = []
    for r in ratios:
        ws = np.round(np.sqrt(w * h / r))
        hs = np.round(ws * r)
        for s in np.asarray(sizes, dtype=np.float64) / stride:
            sw, sh = ws * s, hs * s
            rows.append([cx - 0.5 * (sw - 1), cy - 0.5 * (sh - 1), cx + 0.5 * (sw - 1), cy + 0.5 * (sh - 1)])
    return np.asarray(rows, dtype=np.float64)


# ------------------------------------------------------------------------------------------------ box coder
def decode(codes, boxes, weights):
    """codes [R, K*4] or [R, K*6] (when boxes are [R,4] and 6 | width), boxes [R,4] xyxy -> same shape as codes."""
    codes, boxes = codes.float(), boxes.float()
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0] + 1
    heights = boxes[:, 3] - boxes[:, 1] + 1
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    six = codes.shape[1] % 6 == 0
    n = 6 if six else 4
    dx, dy = codes[:, 0::n] / wx, codes[:, 1::n] / wy
    dw, dh = torch.clamp(codes[:, 2::n] / ww, max=XFORM_CLIP), torch.clamp(codes[:, 3::n] / wh, max=XFORM_CLIP)
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    out = torch.zeros_like(codes)
    out[:, 0::n], out[:, 1::n], out[:, 2::n], out[:, 3::n] = pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph
    if six:
        dxp = codes[:, 4::6] / wx
        dwp = torch.clamp(codes[:, 5::6] / ww, max=XFORM_CLIP)
        pcxp, pwp = dxp * widths[:, None] + ctr_x[:, None], torch.exp(dwp) * widths[:, None]
        out[:, 4::6], out[:, 5::6] = pcxp - 0.5 * pwp, pcxp + 0.5 * pwp
    return out


def clip(boxes, w, h):
    b = boxes.clone()
    b[:, 0::2] = b[:, 0::2].clamp(0, w - 1)
    b[:, 1::2] = b[:, 1::2].clamp(0, h - 1)
    return b


# ------------------------------------------------------------------------------------------------ stereo RPN
def srpn_head(feats_l, feats_r, w):
    """-> (objectness [N,2A,H,W] after the reference's pairwise softmax, box regression [N,6A,H,W]) per level."""
    obj, reg = [], []
    for fl, fr in zip(feats_l, feats_r):
        lt = F.relu(F.conv2d(fl, w["head.conv.weight"], w["head.conv.bias"], 1, 1))
        rt = F.relu(F.conv2d(fr, w["head.conv.weight"], w["head.conv.bias"], 1, 1))
        t = torch.cat((lt, rt), 1)
        s = F.conv2d(t, w["head.cls_logits.weight"], w["head.cls_logits.bias"])
        obj.append(s.view(s.shape[0], 2, -1, s.shape[3]).softmax(1).view(*s.shape))     # pairs channel a with channel A + a
        reg.append(F.conv2d(t, w["head.bbox_pred.weight"], w["head.bbox_pred.bias"]))
    return obj, reg


def double_view_nms(lb, rb, scores, thresh, max_keep, use_keep="joint", strict=True):
    """strict: suppress when IoU > thresh (the reference's CUDA nms) instead of >= (its CPU nms, which recorded the goldens)."""
    kl = nms_oracle.nms(lb.numpy(), scores.numpy(), thresh, strict=strict)
    if use_keep == "left":
        keep = kl
    else:
        kr = nms_oracle.nms(rb.numpy(), scores.numpy(), thresh, strict=strict)
        keep = np.intersect1d(kl, kr)
    if max_keep > 0:
        keep = keep[:max_keep]
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def srpn_select(anchors, objectness, regression, image_sizes, pre_nms_top_n=6000, post_nms_top_n=300, nms_thresh=0.7, min_size=0,
                strict=True):
    """anchors: per-level float32 [H*W*A,4]; objectness/regression: per-level head outputs; image_sizes: [(w, h)] per image.
    -> per image (left [K,4], right [K,4], objectness [K]).  The score of anchor a' at a position is channel 2a'+1 of the
    softmaxed map (the reference flattens [N,H,W,2A] to (-1, 2) and takes column 1)."""
    n = objectness[0].shape[0]
    sc = torch.cat([o.permute(0, 2, 3, 1).reshape(n, -1, 2) for o in objectness], 1)[:, :, 1]
    rg = torch.cat([r.permute(0, 2, 3, 1).reshape(n, -1, 6) for r in regression], 1)
    anc = torch.cat([torch.as_tensor(a, dtype=torch.float32) for a in anchors], 0)
    out = []
    for i in range(n):
        w, h = image_sizes[i]
        p = decode(rg[i], anc, (1.0, 1.0, 1.0, 1.0))
        left, right = clip(p[:, 0:4], w, h), clip(p[:, [4, 1, 5, 3]], w, h)
        order = torch.sort(sc[i], 0, True)[1]
        if 0 < pre_nms_top_n < sc.numel():
            order = order[:pre_nms_top_n]
        left, right, s = left[order], right[order], sc[i][order]
        ok = ((left[:, 2] - left[:, 0] + 1 >= min_size) & (left[:, 3] - left[:, 1] + 1 >= min_size) &
              (right[:, 2] - right[:, 0] + 1 >= min_size) & (right[:, 3] - right[:, 1] + 1 >= min_size))
        left, right, s = left[ok], right[ok], s[ok]      # (the reference filters each side on its own; identical for min_size <= 1)
        keep = double_view_nms(left, right, s, nms_thresh, post_nms_top_n, strict=strict)
        out.append((left[keep], right[keep], s[keep]))
    return out


# ------------------------------------------------------------------------------------------------ pooler
def map_levels(boxes, k_min, k_max, s0=224, lvl0=4, eps=1e-6):
    area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    lv = torch.round(lvl0 + torch.log(torch.sqrt(area) / s0))          # natural log, no eps: as the fork has it
    return torch.clamp(lv, min=k_min, max=k_max).to(torch.int64) - int(k_min)


def pooler(feats, boxes_per_image, image_height, res, scales, sampling_ratio):
    """feats: per level [N,C,H,W] (the extra top level, if given, is ignored like in the fork); -> [R,C,res,res] in box order.
    The ROIAlign scale of a level is feature height / image height (the fork overrides the configured scale)."""
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b.float()], 1) for i, b in enumerate(boxes_per_image)], 0)
    k_min, k_max = -math.log2(scales[0]), -math.log2(scales[-1])
    levels = map_levels(rois[:, 1:], k_min, k_max)
    c = feats[0].shape[1]
    out = torch.zeros(len(rois), c, res, res)
    for lvl in range(len(scales)):
        idx = torch.nonzero(levels == lvl).reshape(-1)
        if len(idx) == 0:
            continue
        f = feats[lvl]
        scale = f.shape[2] / image_height
        out[idx] = torch.from_numpy(roi_oracle.roi_align(f.numpy(), rois[idx].numpy(), scale, res, res, sampling_ratio))
    return out


# ------------------------------------------------------------------------------------------------ stereo box head
def box_head(feats_l, feats_r, props_l, props_r, image_height, w, res=7, scales=(0.25, 0.125, 0.0625, 0.03125), sampling_ratio=0):
    lx = pooler(feats_l, props_l, image_height, res, scales, sampling_ratio)
    rx = pooler(feats_r, props_r, image_height, res, scales, sampling_ratio)
    x = torch.cat([lx, rx], 1)
    x = F.relu(F.conv2d(x, w["box.feature_extractor.RCNN_top.0.weight"], w["box.feature_extractor.RCNN_top.0.bias"], stride=7))
    x = F.relu(F.conv2d(x, w["box.feature_extractor.RCNN_top.3.weight"], w["box.feature_extractor.RCNN_top.3.bias"]))
    x = x.mean(3).mean(2)
    logits = F.linear(x, w["box.predictor.cls_score.weight"], w["box.predictor.cls_score.bias"])
    deltas = F.linear(x, w["box.predictor.bbox_pred.weight"], w["box.predictor.bbox_pred.bias"])
    return x, logits, deltas


def box_post(logits, deltas, props_l, props_r, image_sizes, score_thresh=0.05, nms_thresh=0.5, detections_per_img=100,
             weights=(10.0, 10.0, 5.0, 5.0), strict=True):
    """-> per image dict(left [K,4], right [K,4], scores [K], labels [K]); classes 1.. only; NMS on the left view."""
    prob = F.softmax(logits, -1)
    ncls = prob.shape[1]
    dl = deltas[:, [0, 1, 2, 3, 6, 7, 8, 9]]
    dr = deltas[:, [4, 1, 5, 3, 10, 7, 11, 9]]
    pl = decode(dl, torch.cat(props_l, 0), weights)
    pr = decode(dr, torch.cat(props_r, 0), weights)
    counts = [len(b) for b in props_l]
    out = []
    for prob_i, l_i, r_i, (w, h) in zip(prob.split(counts), pl.split(counts), pr.split(counts), image_sizes):
        l_i = clip(l_i.reshape(-1, 4), w, h).reshape(-1, ncls * 4)
        r_i = clip(r_i.reshape(-1, 4), w, h).reshape(-1, ncls * 4)
        res = {"left": [], "right": [], "scores": [], "labels": []}
        for j in range(1, ncls):
            inds = torch.nonzero(prob_i[:, j] > score_thresh).reshape(-1)
            s, lb, rb = prob_i[inds, j], l_i[inds, j * 4:(j + 1) * 4], r_i[inds, j * 4:(j + 1) * 4]
            keep = double_view_nms(lb, rb, s, nms_thresh, -1, use_keep="left", strict=strict)
            res["left"].append(lb[keep]); res["right"].append(rb[keep]); res["scores"].append(s[keep])
            res["labels"].append(torch.full((len(keep),), j, dtype=torch.int64))
        res = {k: torch.cat(v, 0) for k, v in res.items()}
        nd = len(res["scores"])
        if nd > detections_per_img > 0:
            thr = torch.kthvalue(res["scores"], nd - detections_per_img + 1)[0]
            keep = torch.nonzero(res["scores"] >= thr).reshape(-1)
            res = {k: v[keep] for k, v in res.items()}
        out.append(res)
    return out


# ------------------------------------------------------------------------------------------------ mask head
def mask_head(feats, boxes_per_image, labels_per_image, image_height, w, res=14, scales=(0.25, 0.125, 0.0625, 0.03125), sampling_ratio=2):
    """-> mask probabilities [R,1,28,28] of each detection's own class."""
    x = pooler(feats, boxes_per_image, image_height, res, scales, sampling_ratio)
    for i in (1, 2, 3, 4):
        x = F.relu(F.conv2d(x, w[f"mask.feature_extractor.mask_fcn{i}.weight"], w[f"mask.feature_extractor.mask_fcn{i}.bias"], 1, 1))
    x = F.relu(F.conv_transpose2d(x, w["mask.predictor.conv5_mask.weight"], w["mask.predictor.conv5_mask.bias"], 2))
    logits = F.conv2d(x, w["mask.predictor.mask_fcn_logits.weight"], w["mask.predictor.mask_fcn_logits.bias"])
    prob = logits.sigmoid()
    labels = torch.cat(labels_per_image)
    return prob[torch.arange(len(labels)), labels][:, None]
