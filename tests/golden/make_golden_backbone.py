#!/usr/bin/env python3
"""Golden fixtures for the ResNet-50-FPN backbone, recorded from the IMPORTED REFERENCE (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_backbone.py

The reference package imports third-party / compiled modules that this image lacks and that the backbone's arithmetic
never touches (cv2, pycocotools, yacs, the compiled disprcnn._C, torch._six); they are replaced by inert stand-ins for
the import only.  The code that produces the recorded numbers is the reference's own
disprcnn/modeling/backbone/{backbone,resnet,fpn}.py run by torch-CPU.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

for name in ("cv2", "pycocotools", "pycocotools.mask", "disprcnn._C"):
    sys.modules[name] = MagicMock()


class CfgNode(dict):
    def __init__(self, init=None, *a, **k):
        super().__init__(init or {})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


yacs, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
yc.CfgNode = CfgNode
yacs.config = yc
sys.modules["yacs"], sys.modules["yacs.config"] = yacs, yc
torch._six = types.SimpleNamespace(PY3=True, PY37=True, string_classes=(str,), int_classes=(int,),
                                   container_abcs=__import__("collections").abc)
sys.modules["torch._six"] = torch._six

from disprcnn.config import cfg  # noqa: E402
from disprcnn.modeling.backbone import build_backbone  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402

torch.set_num_threads(8)


backbone_state = synth.synth_backbone_state


def sample(out, tag, t, k=512):
    flat = t.reshape(-1)
    u = synth.hash_uniform(f"sample:{tag}:{flat.numel()}", (k,), 0.0, 1.0).double()
    idx = (u * flat.numel()).long().clamp(max=flat.numel() - 1).numpy()
    out[tag + "_idx"], out[tag + "_val"] = idx, flat[idx].numpy()
    out[tag + "_abssum"] = np.array(flat.double().abs().sum().item())
    out[tag + "_shape"] = np.array(t.shape)


def main():
    cfg.MODEL.BACKBONE.CONV_BODY = "R-50-FPN"
    cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS = 256
    model = build_backbone(cfg)
    sd = backbone_state(model.state_dict())
    model.load_state_dict(sd, strict=True)
    # calibrate BN running statistics (FrozenBatchNorm2d is a plain BatchNorm2d in this fork) by train-mode passes
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats(); m.momentum = None
    model.train()
    with torch.no_grad():
        for p in range(2):
            model(synth.hash_uniform(f"bb:cal{p}", (2, 3, 192, 320), -2.0, 2.0))
    model.eval()
    bn = {k: v.numpy().copy() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez_compressed(os.path.join(HERE, "bn_stats_backbone.npz"), **bn)
    out = {"keys": np.array(list(model.state_dict().keys()))}
    for tag, shape in (("small", (2, 3, 96, 160)), ("odd", (1, 3, 75, 131)), ("kitti", (2, 3, 375, 1242))):
        x = synth.hash_uniform("bb:" + tag, shape, -2.0, 2.0)
        with torch.no_grad():
            feats = model.body(x)
            outs = model(x)
        for i, f in enumerate(feats):
            sample(out, f"{tag}_c{i + 2}", f)
        for i, o in enumerate(outs):
            sample(out, f"{tag}_p{i + 2}", o)
        print(tag, [tuple(o.shape) for o in outs], float(outs[0].abs().mean()))
    np.savez_compressed(os.path.join(HERE, "backbone_golden.npz"), **out)


def main_r101():
    """R-101-FPN (the conv body of the shipped 2D config, configs/kitti/car/vob/mask.yaml:5): calibrated BN statistics + sampled
    pyramids at 96x160 and 75x131 -> backbone_r101_golden.npz / bn_stats_backbone_r101.npz (the R-50 fixtures stay as they are)."""
    cfg.MODEL.BACKBONE.CONV_BODY = "R-101-FPN"
    cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS = 256
    model = build_backbone(cfg)
    sd = backbone_state(model.state_dict())
    model.load_state_dict(sd, strict=True)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats(); m.momentum = None
    model.train()
    with torch.no_grad():
        for p in range(2):
            model(synth.hash_uniform(f"bb:cal{p}", (2, 3, 192, 320), -2.0, 2.0))
    model.eval()
    bn = {k: v.numpy().copy() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez_compressed(os.path.join(HERE, "bn_stats_backbone_r101.npz"), **bn)
    out = {"keys": np.array(list(model.state_dict().keys())), "n_params": np.array(sum(p.numel() for p in model.parameters()))}
    for tag, shape in (("small", (2, 3, 96, 160)), ("odd", (1, 3, 75, 131))):
        x = synth.hash_uniform("bb:" + tag, shape, -2.0, 2.0)
        with torch.no_grad():
            feats = model.body(x)
            outs = model(x)
        for i, f in enumerate(feats):
            sample(out, f"{tag}_c{i + 2}", f)
        for i, o in enumerate(outs):
            sample(out, f"{tag}_p{i + 2}", o)
        print("R-101", tag, [tuple(o.shape) for o in outs], float(outs[0].abs().mean()))
    np.savez_compressed(os.path.join(HERE, "backbone_r101_golden.npz"), **out)


def main_r101_kitti():
    """R-101-FPN on a KITTI-sized stereo pair (2 x 3 x 375 x 1242), with the BatchNorm statistics main_r101 recorded -> backbone_r101_kitti_golden.npz
    (round 5: the shipped 2D config's conv body at the shipped input size; the other fixtures stay as they are)."""
    cfg.MODEL.BACKBONE.CONV_BODY = "R-101-FPN"
    cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS = 256
    model = build_backbone(cfg)
    sd = backbone_state(model.state_dict())
    synth.load_bn_stats(sd, os.path.join(HERE, "bn_stats_backbone_r101.npz"))
    model.load_state_dict(sd, strict=True)
    model.eval()
    out = {}
    x = synth.hash_uniform("bb:kitti", (2, 3, 375, 1242), -2.0, 2.0)
    with torch.no_grad():
        feats = model.body(x)
        outs = model(x)
    for i, f in enumerate(feats):
        sample(out, f"kitti_c{i + 2}", f)
    for i, o in enumerate(outs):
        sample(out, f"kitti_p{i + 2}", o)
    # the same reference modules in fp64: what the fp32 run above approximates (104 convolutions deep, its own rounding noise reaches
    # 6e-4 * max|ref| at p5) -- recorded at the same sample positions, so that a test can hold an fp32 implementation to the reference's
    # OWN distance from exact arithmetic instead of to a bound between two noisy fp32 runs
    model.double()
    with torch.no_grad():
        outs64 = model(x.double())
    for i, o in enumerate(outs64):
        out[f"kitti_p{i + 2}_val64"] = o.reshape(-1)[torch.from_numpy(out[f"kitti_p{i + 2}_idx"])].numpy()
    print("R-101 kitti", [tuple(o.shape) for o in outs], float(outs[0].abs().mean()))
    np.savez_compressed(os.path.join(HERE, "backbone_r101_kitti_golden.npz"), **out)


if __name__ == "__main__":
    if "--r101-kitti" in sys.argv:
        main_r101_kitti()
    elif "--r101" in sys.argv:
        main_r101()
    else:
        main()
