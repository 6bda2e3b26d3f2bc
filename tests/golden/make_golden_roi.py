#!/usr/bin/env python3
"""Golden fixtures for ROIAlign (SURVEY f1 / a11 crop), recorded from the REFERENCE'S OWN CPU KERNEL (authoring container only).

    python oracle/build_ref.py && python tests/golden/make_golden_roi.py

oracle/build_ref.py compiles /root/reference/disprcnn/csrc/cpu/ROIAlign_cpu.cpp (one recorded token patch) into
oracle/_ref/; this script calls its `roi_align_forward` (the function behind `disprcnn._C.roi_align_forward`,
csrc/vision.cpp:9) and stores inputs' specs + outputs.  Images come from disprcnn_amd.utils.synth (rebuilt by the tests).

  small_*  : a 2x3x37x53 image, 6 rois (inside, partly outside, malformed), four (pooled size, sampling_ratio, scale) settings
             -- full output tensors
  crop_*   : the caller's use (disprcnn3d.py:44-50: ROIAlign((224,224), 1.0, 0) on a 375x1242 image pair): 13 roi geometries
             incl. pedestrian/cyclist sizes (BASELINE configs[4]), sides in (224,448] (2 samples per axis) and > 448
             (3 samples), rois sticking out of the image, a malformed roi -- SHA-256 of the full output + 4096 sampled values
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import build_ref  # noqa: E402
from disprcnn_amd.utils import synth  # noqa: E402

SMALL_ROIS = [[0, 3.2, 4.7, 30.9, 28.1], [1, 10.0, 5.0, 24.0, 33.0], [0, -4.0, -3.0, 20.0, 12.0],
              [1, 40.0, 20.0, 60.0, 45.0], [0, 8.0, 8.0, 6.0, 7.0], [1, 0.0, 0.0, 52.0, 36.0]]
SMALL_SETTINGS = [(7, 7, 2, 1.0), (6, 5, 0, 1.0), (14, 14, 0, 0.5), (3, 4, 1, 0.25)]

CROP_ROIS = [  # (batch, x1, y1, x2, y2) on a 1242 x 375 image; integer-aligned like prepare_psmnet_input produces, plus float ones
    [0, 100, 50, 260, 150],          # car: 160 x 100 (1x1 grid)
    [1, 400, 120, 436, 210],         # pedestrian: 36 x 90
    [0, 700, 100, 715, 140],         # far pedestrian: 15 x 40
    [1, 820, 90, 940, 340],          # tall cyclist: 120 x 250 -> 2 samples along y
    [0, 20, 40, 320, 200],           # wide: 300 x 160 -> 2 samples along x
    [1, 500, 10, 960, 370],          # 460 x 360 -> 3 x 2 samples
    [0, -30, -20, 150, 120],         # sticks out top-left
    [1, 1100, 250, 1300, 420],       # sticks out bottom-right (samples beyond +size contribute 0)
    [0, 1241, 374, 1242, 375],       # the last pixel
    [1, 600, 200, 590, 190],         # malformed (x2 < x1): forced to 1 x 1
    [0, 333.25, 77.75, 401.5, 199.125],   # non-integer roi
    [1, 0, 0, 1242, 375],            # the whole image: 6 x 2 samples
    [0, 640, 180, 864, 404],         # exactly 224 x 224, partly below the image
]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = build_ref.load() or (build_ref.build() and build_ref.load())
    assert ref is not None, "run oracle/build_ref.py first (needs /root/reference)"
    out = {}
    img = synth.hash_uniform("roi:img", (2, 3, 37, 53), 0.0, 1.0)
    rois = torch.tensor(SMALL_ROIS, dtype=torch.float32)
    out["small_rois"] = rois.numpy()
    out["small_settings"] = np.array(SMALL_SETTINGS, dtype=np.float64)
    for k, (ph, pw, sr, scale) in enumerate(SMALL_SETTINGS):
        out[f"small_out{k}"] = ref.roi_align_forward(img, rois, scale, ph, pw, sr).numpy()
    pair = synth.hash_uniform("roi:pair", (2, 3, 375, 1242), 0.0, 1.0)
    rois = torch.tensor(CROP_ROIS, dtype=torch.float32)
    crop = ref.roi_align_forward(pair, rois, 1.0, 224, 224, 0).numpy()
    flat = crop.reshape(-1)
    idx = (synth.hash_uniform("roi:idx", (4096,), 0.0, 1.0).double().numpy() * flat.size).astype(np.int64)
    out["crop_rois"] = rois.numpy()
    out["crop_sha"] = np.array(_sha(crop))
    out["crop_idx"], out["crop_val"] = idx, flat[idx]
    out["crop_abssum"] = np.array(np.abs(flat.astype(np.float64)).sum())
    out["crop_roi_sha"] = np.array([_sha(crop[k]) for k in range(len(CROP_ROIS))])
    path = os.path.join(HERE, "roi_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
