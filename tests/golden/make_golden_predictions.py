#!/usr/bin/env python3
"""Manifest of the reference's `predictions.pth` format, recorded from the IMPORTED REFERENCE (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_predictions.py

The reference's own BoxList / DisparityMap (disprcnn/structures/bounding_box.py:10-41, structures/disparity.py:12-21) are filled with
synthetic detections and saved the way engine/inference.py:132-133 saves them (plain torch.save).  Recorded -- DATA ONLY, no pickled
reference class travels: the class paths and helper globals the file names, the instance-dict keys of a BoxList / DisparityMap, the dtype
and shape of every tensor, and two cross-checks run here: this package's reader loads the reference's file, and the reference's own
classes load the file this package writes (same fields, identical tensors)."""
import json
import os
import pickletools
import sys
import tempfile
import zipfile
from unittest.mock import MagicMock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
for name in ("cv2", "pycocotools", "pycocotools.mask", "disprcnn._C"):
    sys.modules[name] = MagicMock()

from disprcnn.structures.bounding_box import BoxList as RefBoxList  # noqa: E402  (the reference)
from disprcnn.structures.disparity import DisparityMap as RefDisparityMap  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402
from disprcnn_amd.utils.predictions_io import load_predictions, save_predictions  # noqa: E402
from disprcnn_amd.structures.bounding_box import BoxList  # noqa: E402
from disprcnn_amd.structures.disparity import DisparityMap  # noqa: E402

W, H = 1242, 375


def fields(tag, r):
    return {"scores": synth.hash_uniform(tag + ":s", (r,), 0.0, 1.0), "labels": torch.ones(r, dtype=torch.int64),
            "mask": synth.hash_uniform(tag + ":m", (r, 1, 28, 28), 0.0, 1.0), "disparity": synth.hash_uniform(tag + ":d", (r, 224, 224), -48.0, 48.0)}


def build(cls_box, cls_map):
    out = {"left": [], "right": []}
    for side in ("left", "right"):
        for img, r in enumerate((3, 0, 5)):
            b = cls_box(synth.hash_uniform(f"pred:{side}{img}", (r, 4), 0.0, 300.0), (W, H))
            for k, v in fields(f"pred:{side}{img}", r).items():
                b.add_field(k, v)
            if side == "left" and img == 0:
                b.PixelWise_map["disparity"] = cls_map(synth.hash_uniform("pred:map", (H, W), 0.0, 80.0))
            out[side].append(b)
    return out


def globals_of(path):
    z = zipfile.ZipFile(path)
    data = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    return sorted({a for op, a, _ in pickletools.genops(data) if op.name == "GLOBAL"})


def describe(preds):
    d = {}
    for side, lst in preds.items():
        d[side] = [{"size": list(b.size), "mode": b.mode, "bbox": [str(b.bbox.dtype), list(b.bbox.shape)], "state_keys": sorted(b.__dict__),
                    "fields": {k: [str(v.dtype), list(v.shape)] for k, v in b.extra_fields.items()},
                    "maps": {k: [type(v).__name__, sorted(v.__dict__), str(v.data.dtype), list(v.data.shape)] for k, v in b.PixelWise_map.items()}}
                   for b in lst]
    return d


def same(a, b):
    for side in a:
        for x, y in zip(a[side], b[side]):
            assert tuple(x.size) == tuple(y.size) and x.mode == y.mode and torch.equal(x.bbox, y.bbox)
            assert list(x.extra_fields) == list(y.extra_fields)
            for k in x.extra_fields:
                assert torch.equal(x.extra_fields[k], y.extra_fields[k]), k
            assert list(x.PixelWise_map) == list(y.PixelWise_map)
            for k in x.PixelWise_map:
                assert torch.equal(x.PixelWise_map[k].data, y.PixelWise_map[k].data)
    return True


def main():
    tmp = tempfile.mkdtemp()
    ref_preds = build(RefBoxList, RefDisparityMap)
    ref_path = os.path.join(tmp, "predictions_ref.pth")
    torch.save(ref_preds, ref_path)                                   # engine/inference.py:132-133
    ours = build(BoxList, DisparityMap)
    our_path = os.path.join(tmp, "predictions_ours.pth")
    save_predictions(ours, our_path)
    got = load_predictions(ref_path)                                  # our reader on the reference's file
    assert isinstance(got["left"][0], BoxList) and same(got, ref_preds)
    back = torch.load(our_path, map_location="cpu", weights_only=False)   # the reference's classes on our file
    assert type(back["left"][0]) is RefBoxList and type(back["left"][0].PixelWise_map["disparity"]) is RefDisparityMap and same(back, ours)
    assert back["left"][0].mask_thresh == 0.5
    manifest = {"reference_globals": globals_of(ref_path), "our_globals": globals_of(our_path), "layout": describe(ref_preds),
                "our_reader_loads_reference_file": True, "reference_classes_load_our_file": True,
                "generator": "synth.hash_uniform tags pred:{left,right}{0,1,2}[:s|:m|:d], pred:map; ROI counts (3, 0, 5)"}
    with open(os.path.join(HERE, "predictions_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest["reference_globals"]), "\n", json.dumps(manifest["our_globals"]))


if __name__ == "__main__":
    main()
