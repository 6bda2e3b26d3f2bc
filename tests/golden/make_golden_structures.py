#!/usr/bin/env python3
"""Golden fixtures for the value holders' geometry, recorded from the IMPORTED REFERENCE classes (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_structures.py

Reference entry points exercised (their own Python, torch-CPU):
  DisparityMap.resize (bilinear and use_max_pooling) / .crop / .__sub__   disprcnn/structures/disparity.py:39-83
  BoxList.convert / .resize / .transpose / .crop                          disprcnn/structures/bounding_box.py:119-277
cv2, pycocotools and the compiled disprcnn._C are imported by neighbouring modules but never touched by this code; they are
replaced by inert stand-ins for the import only.  Inputs come from disprcnn_amd.utils.synth (rebuilt by the tests); only the
case parameters and the resulting arrays are stored.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
for name in ("cv2", "pycocotools", "pycocotools.mask", "disprcnn._C"):
    sys.modules[name] = MagicMock()

from disprcnn.structures.bounding_box import BoxList  # noqa: E402  (the reference)
from disprcnn.structures.disparity import DisparityMap  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402

from _structures_cases import BOXES, CROPS, RESIZE, SIZE  # noqa: E402


def main():
    out = {}
    for name, (h, w), dst in RESIZE:
        d = synth.hash_uniform(f"structures:resize:{name}", (h, w), -48.0, 48.0)
        out[f"resize:{name}:bilinear"] = DisparityMap(d).resize(dst).data.numpy()
        out[f"resize:{name}:maxpool"] = DisparityMap(d).resize(dst, use_max_pooling=True).data.numpy()
    for name, (h, w), box in CROPS:
        d = synth.hash_uniform(f"structures:crop:{name}", (h, w), -48.0, 48.0)
        out[f"crop:{name}"] = DisparityMap(d).crop(box).data.numpy()
        out[f"sub:{name}"] = (DisparityMap(d) - 3.25).data.numpy()[::4, ::4].copy()
    b = BoxList(torch.tensor(BOXES), SIZE)
    bw = b.convert("xywh")
    out["box:xywh"] = bw.bbox.numpy()
    out["box:xywh_back"] = bw.convert("xyxy").bbox.numpy()
    for tag, src in (("xyxy", b), ("xywh", bw)):
        out[f"box:{tag}:resize_equal"] = src.resize((640, 192)).bbox.numpy()
        out[f"box:{tag}:resize_unequal"] = src.resize((400, 300)).bbox.numpy()
        out[f"box:{tag}:flip_lr"] = src.transpose(0).bbox.numpy()
        out[f"box:{tag}:flip_tb"] = src.transpose(1).bbox.numpy()
        c = src.crop((40, 10, 250, 80))
        out[f"box:{tag}:crop"] = c.bbox.numpy()
        out[f"box:{tag}:crop_size"] = np.asarray(c.size, dtype=np.int64)
    # maps and non-tensor fields follow the boxes: a DisparityMap in PixelWise_map through resize and crop(crop_map=True)
    d = synth.hash_uniform("structures:map", (SIZE[1], SIZE[0]), 0.0, 64.0)
    bm = BoxList(torch.tensor(BOXES), SIZE)
    bm.add_map("disparity", DisparityMap(d))
    out["map:resize"] = bm.resize((160, 48)).get_map("disparity").data.numpy()
    out["map:crop"] = bm.crop((40, 10, 250, 80), crop_map=True).get_map("disparity").data.numpy()
    out["map:crop_nomap_shape"] = np.asarray(bm.crop((40, 10, 250, 80)).get_map("disparity").data.shape, dtype=np.int64)      # handed over uncropped
    np.savez_compressed(os.path.join(HERE, "structures_golden.npz"), **out)
    print("wrote structures_golden.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
