"""`predictions.pth` wire format (reference engine/inference.py:125-133): the writer names the reference's class paths and
nothing else, the reader maps them back and refuses foreign globals.  Pinned by tests/golden/predictions_manifest.json, which
tests/golden/make_golden_predictions.py recorded from the reference's own BoxList / DisparityMap saved with plain torch.save (it also
checked there that the reference's classes load our file and our reader loads the reference's file)."""
import json
import os
import pickle
import pickletools
import sys
import zipfile

import pytest
import torch

from disprcnn_amd.structures.bounding_box import BoxList
from disprcnn_amd.structures.disparity import DisparityMap
from disprcnn_amd.utils import synth
from disprcnn_amd.utils.predictions_io import load_predictions, save_predictions

from .conftest import GOLDEN

W, H = 1242, 375


def _build():
    out = {"left": [], "right": []}
    for side in ("left", "right"):
        for img, r in enumerate((3, 0, 5)):
            tag = f"pred:{side}{img}"
            b = BoxList(synth.hash_uniform(tag, (r, 4), 0.0, 300.0), (W, H))
            b.add_field("scores", synth.hash_uniform(tag + ":s", (r,), 0.0, 1.0))
            b.add_field("labels", torch.ones(r, dtype=torch.int64))
            b.add_field("mask", synth.hash_uniform(tag + ":m", (r, 1, 28, 28), 0.0, 1.0))
            b.add_field("disparity", synth.hash_uniform(tag + ":d", (r, 224, 224), -48.0, 48.0))
            if side == "left" and img == 0:
                b.add_map("disparity", DisparityMap(synth.hash_uniform("pred:map", (H, W), 0.0, 80.0)))
            out[side].append(b)
    return out


def _pickle_of(path):
    z = zipfile.ZipFile(path)
    return z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])


def test_written_file_matches_the_reference_layout(tmp_path):
    manifest = json.load(open(os.path.join(GOLDEN, "predictions_manifest.json")))
    assert manifest["our_reader_loads_reference_file"] and manifest["reference_classes_load_our_file"]
    path = str(tmp_path / "predictions.pth")
    preds = _build()
    before = {m: v for m, v in sys.modules.items() if m == "disprcnn" or m.startswith("disprcnn.")}
    save_predictions(preds, path)
    after = {m: v for m, v in sys.modules.items() if m == "disprcnn" or m.startswith("disprcnn.")}
    # the stand-in modules must not outlive the save: whatever was registered under the reference's names before (nothing, or this repo's
    # `disprcnn` alias package) is there again, the very same objects
    assert after.keys() == before.keys() and all(after[m] is before[m] for m in before)
    data = _pickle_of(path)
    got_globals = sorted({a for op, a, _ in pickletools.genops(data) if op.name == "GLOBAL"})
    assert got_globals == manifest["reference_globals"]                  # the same class paths and helpers, nothing of ours
    back = load_predictions(path)
    for side, lst in manifest["layout"].items():
        assert len(back[side]) == len(lst)
        for b, want, src in zip(back[side], lst, preds[side]):
            assert type(b) is BoxList and list(b.size) == want["size"] and b.mode == want["mode"]
            assert sorted(b.__dict__) == want["state_keys"]
            assert [str(b.bbox.dtype), list(b.bbox.shape)] == want["bbox"] and torch.equal(b.bbox, src.bbox)
            assert {k: [str(v.dtype), list(v.shape)] for k, v in b.extra_fields.items()} == want["fields"]
            for k in b.fields():
                assert torch.equal(b.get_field(k), src.get_field(k))
            for k, (cls, keys, dt, shape) in want["maps"].items():
                m = b.get_map(k)
                assert type(m) is DisparityMap and sorted(m.__dict__) == keys and [str(m.data.dtype), list(m.data.shape)] == [dt, shape]
                assert torch.equal(m.data, src.get_map(k).data)


def test_mono_list_round_trip_and_foreign_globals_refused(tmp_path):
    path = str(tmp_path / "p.pth")
    lst = _build()["left"]
    save_predictions(lst, path)                                          # the mono detectors save a plain list (inference.py:120-121)
    back = load_predictions(path)
    assert isinstance(back, list) and len(back) == 3 and len(back[2]) == 5 and back[1].bbox.shape == (0, 4)
    evil = str(tmp_path / "evil.pth")
    torch.save({"left": [os.path.join]}, evil)                           # a file that names a function: data files may not
    with pytest.raises(pickle.UnpicklingError):
        load_predictions(evil)
