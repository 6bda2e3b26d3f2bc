"""CPU ORACLE (test infrastructure) for the step behind the path: per-ROI disparities -> full-image disparity / depth maps.

Restated from the reference (torch-CPU; the resampling primitive is the same F.interpolate the reference calls):
  resize_disparity     structures/disparity.py:38-60  (bilinear, align_corners=True, values scaled by dst_w / src_w)
  crop_disparity       structures/disparity.py:66-77
  disparity_map        modeling/psmnet/inference.py:18-47 (DisparityMapProcessor._forward_single_image); with clamp0 / masks the
                       variant of DispRCNN3D.roi_disp_postprocess (modeling/detector/disprcnn3d.py:161-190)
  roi_depth_maps       modeling/pointnet_module/point_rcnn/lib/net/point_rcnn.py:121-133 (fu*b / (disp + 1e-6), one map per ROI)
Pinned by tests/golden/post_golden.npz, recorded from the imported reference (tests/golden/make_golden_post.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import torch
import torch.nn.functional as F


def expand_box_to_integer(box):
    """utils/stereo_utils.py:219-229"""
    x1, y1, x2, y2 = box
    return math.floor(x1), math.floor(y1), math.ceil(x2), math.ceil(y2)


def resize_disparity(d, dst_w, dst_h):
    t = F.interpolate(d.float()[None, None], (dst_h, dst_w), mode="bilinear", align_corners=True)[0, 0]
    return t / d.shape[1] * dst_w


def crop_disparity(d, box):
    x1, y1, x2, y2 = map(round, box)
    c = d[y1:y2, x1:x2]
    out = torch.zeros((y2 - y1, x2 - x1), dtype=d.dtype)
    out[:c.shape[0], :c.shape[1]] = c
    return out


def _roi_patch(left_box, right_box, disp_roi):
    x1, y1, x2, y2 = expand_box_to_integer(left_box)
    x1p, _, x2p, _ = expand_box_to_integer((right_box[0], left_box[1], right_box[2], left_box[3]))
    patch = crop_disparity(resize_disparity(disp_roi, max(x2 - x1, x2p - x1p), y2 - y1), (0, 0, x2 - x1, y2 - y1))
    return (x1, y1, x2, y2), patch + x1 - x1p


def disparity_map(left_bbox, right_bbox, disparity_preds, height, width, clamp0=False, masks=None):
    """left/right_bbox [R,4] xyxy, disparity_preds [R,S,S] -> [height,width]; max over the per-ROI maps (zero outside a box)."""
    if len(left_bbox) == 0:
        return torch.zeros((height, width))
    maps = []
    for i, (lb, rb, d) in enumerate(zip(left_bbox.tolist(), right_bbox.tolist(), disparity_preds)):
        (x1, y1, x2, y2), patch = _roi_patch(lb, rb, d)
        m = torch.zeros((height, width))
        m[y1:y2, x1:x2] = patch
        if clamp0:
            m = m.clamp(min=0)
        if masks is not None:
            m = m * masks[i].float()
        maps.append(m)
    return torch.stack(maps).max(dim=0)[0]


def roi_depth_maps(left_bbox, right_bbox, disparity_preds, height, width, fuxb):
    """One full-image depth map per ROI: fuxb / (disparity + 1e-6) inside the box, zero elsewhere -> [R,height,width]."""
    out = torch.zeros((len(left_bbox), height, width))
    for i, (lb, rb, d) in enumerate(zip(left_bbox.tolist(), right_bbox.tolist(), disparity_preds)):
        (x1, y1, x2, y2), patch = _roi_patch(lb, rb, d)
        out[i, y1:y2, x1:x2] = fuxb / (patch + 1e-6)
    return out
