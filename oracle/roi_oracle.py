"""CPU ORACLE (test infrastructure) for the step in front of the path: ROIAlign and the left/right ROI pairing.

Restated from the reference sources (numpy / pure Python loops -- small cases only):
  roi_align          csrc/cpu/ROIAlign_cpu.cpp:18-111 (bilinear pre-calc), :113-219 (forward)
  align_roi_pair     modeling/detector/disprcnn3d.py:118-146 + utils/stereo_utils.py:219-229
  crop_and_normalise modeling/detector/disprcnn3d.py:44-50

Pinning: PINNED (round 2).  The reference's own CPU kernel (csrc/cpu/ROIAlign_cpu.cpp) is compiled from where it lies
by oracle/build_ref.py (one recorded token patch) into oracle/_ref/; tests/golden/make_golden_roi.py recorded its
outputs for 13 ROI geometries (in-image, out-of-image, >224-px sides with 2x2 and 3-sample grids, malformed 1x1, pedestrian /
cyclist sizes) into tests/golden/roi_golden.npz, and tests/test_oracle_roi.py holds this restatement to them BIT FOR BIT
(plus, when oracle/_ref is present, live against the binary on random rois).  The known-answer tests stay as a second pin.
`roi_align` is the vectorised form (same float32 operation order per sample); `roi_align_loops` the literal loop nest.
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
    """inp [B,C,H,W] float32, rois [K,5] -> [K,C,ph,pw] float32: the loop nest below, vectorised over (c, i, j) per sample
    point -- every float32 operation and the accumulation order over (iy, ix) are those of ROIAlign_cpu.cpp:32-105,186-213."""
    inp = np.asarray(inp, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    K, (B, C, H, W) = rois.shape[0], inp.shape
    out = np.zeros((K, C, ph, pw), dtype=np.float32)
    f = np.float32
    ii = np.arange(ph, dtype=np.float32)[:, None]
    jj = np.arange(pw, dtype=np.float32)[None, :]
    for k in range(K):
        b = int(rois[k, 0])
        rsw, rsh, rew, reh = (f(rois[k, 1] * f(spatial_scale)), f(rois[k, 2] * f(spatial_scale)),
                              f(rois[k, 3] * f(spatial_scale)), f(rois[k, 4] * f(spatial_scale)))
        rw, rh = max(f(rew - rsw), f(1.0)), max(f(reh - rsh), f(1.0))
        bin_h, bin_w = f(rh / f(ph)), f(rw / f(pw))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        img = inp[b].reshape(C, H * W)
        acc = np.zeros((C, ph, pw), dtype=np.float32)
        for iy in range(gh):
            yy = (rsh + ii * bin_h) + f(f(iy + 0.5) * bin_h) / f(gh)                 # [ph,1] float32
            for ix in range(gw):
                xx = (rsw + jj * bin_w) + f(f(ix + 0.5) * bin_w) / f(gw)             # [1,pw]
                y, x = np.broadcast_to(yy, (ph, pw)).copy(), np.broadcast_to(xx, (ph, pw)).copy()
                dead = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
                y[y <= 0] = 0; x[x <= 0] = 0
                y_low, x_low = y.astype(np.int64), x.astype(np.int64)
                ycl, xcl = y_low >= H - 1, x_low >= W - 1
                y_low[ycl] = H - 1; x_low[xcl] = W - 1
                y_high, x_high = np.where(ycl, y_low, y_low + 1), np.where(xcl, x_low, x_low + 1)
                y = np.where(ycl, y_low.astype(np.float32), y); x = np.where(xcl, x_low.astype(np.float32), x)
                y_low[dead] = y_high[dead] = x_low[dead] = x_high[dead] = 0
                ly, lx = (y - y_low.astype(np.float32)).astype(np.float32), (x - x_low.astype(np.float32)).astype(np.float32)
                hy, hx = (f(1.0) - ly).astype(np.float32), (f(1.0) - lx).astype(np.float32)
                w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
                for w_ in (w1, w2, w3, w4):
                    w_[dead] = 0
                v1, v2 = img[:, y_low * W + x_low], img[:, y_low * W + x_high]
                v3, v4 = img[:, y_high * W + x_low], img[:, y_high * W + x_high]
                acc = acc + (((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4)
        out[k] = acc / f(gh * gw)
    return out


def roi_align_loops(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    """The literal loop nest (small cases only): inp [B,C,H,W] float32, rois [K,5] -> [K,C,ph,pw] float32."""
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
