"""A/B of engine.TRUNK_S16 (the 3x3 layers of ResNet-FPN and of the RPN head on large maps through the split-f16 kernel, BridgedConv2dS16):
R-50-FPN trunk on one KITTI pair and the whole 2D stage, off / on at several `min_tiles` thresholds; max difference of the pyramid against the
all-fp32-MFMA trunk.      python tools/experiments/exp_trunk_s16.py        (on an MI355X)
"""
import os
import sys
import time
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E  # noqa: E402
from disprcnn_amd.modeling.backbone import build_backbone  # noqa: E402
from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d  # noqa: E402
from disprcnn_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
pair = synth.hash_uniform("benchpair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)


def timeit(fn, n=5):
    with torch.no_grad():
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def trunk():
    bb = build_backbone(NS(MODEL=NS(BACKBONE=NS(CONV_BODY="R-50-FPN"), RESNETS=NS(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256))))
    bsd = synth.synth_backbone_state(bb.state_dict())
    bnf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "bn_stats_backbone.npz")
    if os.path.exists(bnf):
        synth.load_bn_stats(bsd, bnf)
    bb.load_state_dict(bsd)
    return bb.to(dev).eval()


def stage2d():
    m = DispRCNN(default_cfg_2d("R-50-FPN"))
    sd = m.state_dict()
    heads = synth.synth_det_state({k: v for k, v in sd.items() if not k.startswith("backbone.")},
                                  gain={("rpn." if k.startswith("head.") else "roi_heads.") + k: v for k, v in synth.DET_GAIN.items()})
    bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
    m.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **heads}, strict=True)
    return m.to(dev).eval()


ref = None
for tag, en, mt in (("off", False, 96), ("min_tiles 48", True, 48), ("min_tiles 96", True, 96), ("min_tiles 192", True, 192), ("min_tiles 384", True, 384)):
    E.TRUNK_S16["enabled"], E.TRUNK_S16["min_tiles"] = en, mt
    bb = trunk()
    tb, feats = timeit(lambda: bb(pair))
    bridged = sorted(getattr(bb, "_rt", None)._ws[(2, 375, 1242)]["s16"]) if hasattr(bb, "_rt") else "?"
    if ref is None:
        ref = [f.clone() for f in feats]
        diff = ""
    else:
        diff = "  pyramid max|diff| vs off: " + ", ".join(f"{(a - b).abs().max().item():.2e}/{b.abs().max().item():.1f}" for a, b in zip(feats, ref))
    del bb
    m = stage2d()
    t2, _ = timeit(lambda: m({"left": pair[:1], "right": pair[1:]}))
    del m
    print(f"{tag:14s} trunk {tb:6.2f} ms   2D stage {t2:6.2f} ms   bridged trunk layers {bridged}{diff}", flush=True)
