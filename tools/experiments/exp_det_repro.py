"""Development tool: run the 2D stage several times on one input and report the detection counts (determinism check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
if os.environ.get("ODD") == "0":
    E.WINO2D["odd"] = False
n, h, w = 2, 160, 256
m = DispRCNN(default_cfg_2d("R-50-FPN", post_nms_top_n_test=40))
sd = m.state_dict()
w_rpn = synth.synth_det_state({k[4:]: v for k, v in sd.items() if k.startswith("rpn.")}, gain=synth.DET_GAIN)
w_heads = synth.synth_det_state({k[10:]: v for k, v in sd.items() if k.startswith("roi_heads.")}, gain=synth.DET_GAIN)
bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
m.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **{"rpn." + k: v for k, v in w_rpn.items()},
                   **{"roi_heads." + k: v for k, v in w_heads.items()}}, strict=True)
m = m.to(dev).eval()
if os.environ.get("BLK") == "0":
    m.backbone._rt = None
    import disprcnn_amd.modeling.backbone.runtime as R
    R.BackboneRuntime.blocked_levels = lambda self: None
left, right = synth.synth_images(n, h, w, tag="e2e2d")
with torch.no_grad():
    for it in range(4):
        out = m({"left": left.to(dev), "right": right.to(dev)})
        feats = m.backbone(torch.cat((left, right), 0).to(dev))
        lv = m.backbone._rt.blocked_levels()
        fl, fr = [f[:n] for f in feats], [f[n:] for f in feats]
        a = m.rpn._head(fl, fr)
        msg = [len(b) for b in out["left"]]
        if lv is not None:
            b = m.rpn._head_blocked(lv)
            msg.append([float((x - y).abs().max()) for x, y in zip(a[0] + a[1], b[0] + b[1])])
        print(it, msg, [float(f.abs().sum()) for f in feats])
