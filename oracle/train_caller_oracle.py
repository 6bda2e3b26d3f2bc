"""CPU ORACLE (test infrastructure) for the TRAINING entry of the disparity stage: ROI selection and target preparation.

Restated from the reference (torch-CPU; F.interpolate is the primitive the reference itself calls):
  remove_illegal_detections   modeling/detector/disprcnn3d.py:286-294
  remove_low_score_rois       modeling/detector/disprcnn3d.py:192-207
  truncate_rois               modeling/detector/disprcnn3d.py:223-243   (MAX_ROI_FOR_TRAINING)
  paste_mask                  modeling/roi_heads/mask_head/inference.py:90-150  (Masker(0.7, padding=1))
  roi_targets                 modeling/detector/disprcnn3d.py:60-100 + structures/disparity.py:38-77
Pinned by tests/golden/train_caller_golden.npz, recorded from the imported reference (tests/golden/make_golden_train_caller.py);
tests/test_oracle_train_caller.py holds this file to it.  Only tests/ may import this module.
"""
import torch
import torch.nn.functional as F

from . import roi_oracle as R


def legal_keep(lb, rb):
    return (lb[:, 2] > lb[:, 0] + 1) & (lb[:, 3] > lb[:, 1] + 1) & (rb[:, 2] > rb[:, 0] + 1) & (rb[:, 3] > rb[:, 1] + 1)


def low_score_keep(scores_per_image, thresh):
    """-> list of bool keep masks (the reference's odd branches kept literally: `1 < n < 2` can never hold; one survivor keeps all)."""
    scores = torch.cat(scores_per_image) if scores_per_image else torch.zeros(0)
    keep = scores > thresh
    if 1 < keep.sum() < 2:
        idxs = scores.argsort(descending=True)
        keep[idxs[0]] = keep[idxs[1]] = True
    elif keep.sum() == 1:
        keep.fill_(True)
    return list(torch.split(keep, [len(s) for s in scores_per_image]))


def truncate_counts(counts, max_rois):
    """ROIs per image after keeping only the first max_rois of the batch (reference :229-243)."""
    out, s = [], 0
    for c in counts:
        k = 0 if s >= max_rois else min(max_rois - s, c)
        out.append(k)
        s += k
    return out


def paste_mask(mask, box, im_h, im_w, thresh=0.7, padding=1):
    """[M,M] probabilities + xyxy box -> [im_h,im_w] uint8 (paste_mask_in_image)."""
    mask, box = mask.float(), box.float()
    M = mask.shape[-1]
    scale = float(M + 2 * padding) / M
    padded = mask.new_zeros((1, 1, M + 2 * padding, M + 2 * padding))
    padded[0, 0, padding:-padding, padding:-padding] = mask
    w_half, h_half = (box[2] - box[0]) * .5, (box[3] - box[1]) * .5
    x_c, y_c = (box[2] + box[0]) * .5, (box[3] + box[1]) * .5
    w_half, h_half = w_half * scale, h_half * scale
    b = torch.stack([x_c - w_half, y_c - h_half, x_c + w_half, y_c + h_half]).to(torch.int32)
    w, h = max(int(b[2] - b[0] + 1), 1), max(int(b[3] - b[1] + 1), 1)
    m = F.interpolate(padded, size=(h, w), mode="bilinear", align_corners=False)[0, 0] > thresh
    im = torch.zeros((im_h, im_w), dtype=torch.uint8)
    x0, x1, y0, y1 = max(int(b[0]), 0), min(int(b[2]) + 1, im_w), max(int(b[1]), 0), min(int(b[3]) + 1, im_h)
    im[y0:y1, x0:x1] = m[(y0 - int(b[1])):(y1 - int(b[1])), (x0 - int(b[0])):(x1 - int(b[0]))].to(torch.uint8)
    return im


def roi_targets(lbox, rbox, mask_prob, gt_mask_full, disp_map, res):
    """One ROI: -> (roi, roi_right, target [res,res] f32, mask [res,res] u8)."""
    H, W = disp_map.shape
    x1, y1, x1p, y2, mw = R.align_roi_pair(lbox, rbox, W, H)
    crop = torch.zeros((y2 - y1, mw))
    c = disp_map[y1:y2, x1:x1 + mw]
    crop[:c.shape[0], :c.shape[1]] = c
    crop = crop - (x1 - x1p)
    tgt = F.interpolate(crop[None, None], (res, res), mode="bilinear", align_corners=True)[0, 0] / mw * res
    mp = paste_mask(mask_prob, torch.tensor(lbox), H, W) & gt_mask_full
    mk = F.interpolate(mp[y1:y2, x1:x1 + mw][None, None].float(), (res, res), mode="bilinear", align_corners=True)[0, 0].to(torch.uint8)
    return (x1, y1, x1 + mw, y2), (x1p, y1, x1p + mw, y2), tgt, mk
