"""Development tool: Config B (16 ROI crops 224x224, D=96, full PSMNet), the stress shape, the headline step and the R-50-FPN trunk, for
rocprofv3 --kernel-trace (WHAT = psm | psm16 | psm16f | cfgA | anything else: trunk; DRC_LIB = a variant library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):          # development: a variant library from tools/experiments/build_variant_any.sh
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from types import SimpleNamespace as NS
from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
from disprcnn_amd.modeling.backbone import build_backbone
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
what = os.environ.get("WHAT", "psm")
if what in ("psm16", "psm16f"):
    m = PSMNet(48, -48)
    m.load_state_dict(synth.synth_state_dict(m.state_dict()), strict=True)
    m.regressor_storage = "f16"
    if what == "psm16f":
        m.feature_storage = "f16"
    m = m.to(dev).eval()
    l, r = synth.synth_images(64, 224, 224, tag="benchB64")
    l, r = l.to(dev), r.to(dev)
    f = lambda: m((l, r))
elif what == "cfgA":                    # the headline step: Config A regressor from ROI features (ROIS = batch, default 1024)
    import bench
    m, _sd = bench.build_model(dev, 48, 0, "A")
    fl, fr = synth.synth_features(int(os.environ.get("ROIS", "1024")), 32, 28, 28, tag="bench0")
    fl, fr = fl.to(dev), fr.to(dev)
    f = lambda: m.forward_from_features(fl, fr, (112, 112))
elif what == "psm":
    m = PSMNet(48, -48)
    m.load_state_dict(synth.synth_state_dict(m.state_dict()), strict=True)
    m = m.to(dev).eval()
    l, r = synth.synth_images(16, 224, 224, tag="benchB")
    l, r = l.to(dev), r.to(dev)
    f = lambda: m((l, r))
else:
    bb = build_backbone(NS(MODEL=NS(BACKBONE=NS(CONV_BODY="R-50-FPN"), RESNETS=NS(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256))))
    bb.load_state_dict(synth.synth_backbone_state(bb.state_dict()))
    bb = bb.to(dev).eval()
    pair = synth.hash_uniform("benchpair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)
    f = lambda: bb(pair)
with torch.no_grad():
    for _ in range(2):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
print(what, "ms", (time.perf_counter() - t0) / 5 * 1e3)
