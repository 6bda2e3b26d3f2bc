"""CPU ORACLE (test infrastructure) for the step in front of the path: ROIAlign and the left/right ROI pairing.

Restated from the reference sources (numpy / pure Python loops -- small cases only):
  roi_align          csrc/cpu/ROIAlign_cpu.cpp:18-111 (bilinear pre-calc), :113-219 (forward)
  align_roi_pair     modeling/detector/disprcnn3d.py:118-146 + utils/stereo_utils.py:219-229
  crop_and_normalise modeling/detector/disprcnn3d.py:44-50

Pinning: the reference's ROIAlign cannot be built here (csrc/cpu/ROIAlign_cpu.cpp:242 uses the removed
``Tensor::type()`` dispatch; compiling it would need stand-in headers), and the reference ships no test
vectors for it.  PARITY UNPINNED against a reference binary; the restatement is pinned by known-answer tests instead
(tests/test_oracle_roi.py: exactness on affine images, constant images, 1x1 clamp of malformed rois, zero
contribution outside [-1,size], adaptive grid = ceil(roi/pooled)).
"""
import math

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _bilinear(img, y, x):
    """One sample of a [H,W] float32 image; semantics of pre_calc_for_bilinear_interpolate."""
    H, W = img.shape
    f = np.float32
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return f(0.0)
    y = f(max(y, 0.0)); x = f(max(x, 0.0))
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1; y = f(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1; x = f(x_low)
    else:
        x_high = x_low + 1
    ly, lx = f(y - y_low), f(x - x_low)
    hy, hx = f(1.0) - ly, f(1.0) - lx
    return (f(hy * hx) * img[y_low, x_low] + f(hy * lx) * img[y_low, x_high] +
            f(ly * hx) * img[y_high, x_low] + f(ly * lx) * img[y_high, x_high])


def roi_align(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    """inp [B,C,H,W] float32, rois [K,5] -> [K,C,ph,pw] float32 (all arithmetic in float32 like the reference's T=float)."""
    inp = np.asarray(inp, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    K, (B, C, H, W) = rois.shape[0], inp.shape
    out = np.zeros((K, C, ph, pw), dtype=np.float32)
    f = np.float32
    for k in range(K):
        b = int(rois[k, 0])
        rsw, rsh, rew, reh = (f(rois[k, 1] * f(spatial_scale)), f(rois[k, 2] * f(spatial_scale)),
                              f(rois[k, 3] * f(spatial_scale)), f(rois[k, 4] * f(spatial_scale)))
        rw, rh = max(f(rew - rsw), f(1.0)), max(f(reh - rsh), f(1.0))
        bin_h, bin_w = f(rh / f(ph)), f(rw / f(pw))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        for c in range(C):
            for i in range(ph):
                for j in range(pw):
                    acc = f(0.0)
                    for iy in range(gh):
                        yy = f(rsh + f(i * bin_h) + f(f(iy + 0.5) * bin_h / f(gh)))
                        for ix in range(gw):
                            xx = f(rsw + f(j * bin_w) + f(f(ix + 0.5) * bin_w / f(gw)))
                            acc = f(acc + _bilinear(inp[b, c], yy, xx))
                    out[k, c, i, j] = f(acc / f(gh * gw))
    return out


def expand_box_to_integer(box):
    x1, y1, x2, y2 = box
    return math.floor(x1), math.floor(y1), math.ceil(x2), math.ceil(y2)


def align_roi_pair(leftbox, rightbox, width, height):
    """-> (x1, y1, x1p, y2, max_width): the crop geometry of one left/right detection pair."""
    x1, y1, x2, y2 = expand_box_to_integer(leftbox)
    x1p, _, x2p, _ = expand_box_to_integer(rightbox)
    x1, x1p, y1 = max(0, x1), max(0, x1p), max(0, y1)
    y2, x2, x2p = min(y2, height - 1), min(x2, width - 1), min(x2p, width - 1)
    mw = max(x2 - x1, x2p - x1p)
    mw = min(mw, min(width - x1, width - x1p))
    return x1, y1, x1p, y2, mw


def crop_and_normalise(images, rois, res):
    """ROIAlign((res,res), 1.0, 0) on the image, then (x - mean) / std per channel."""
    crop = roi_align(images, rois, 1.0, res, res, 0)
    return (crop - MEAN[None, :, None, None]) / STD[None, :, None, None]
