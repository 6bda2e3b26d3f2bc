import os, sys, torch
sys.path.insert(0, os.getcwd())
from disprcnn_amd.utils import synth
from tests.helpers import state_for
from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
dev = torch.device("cuda:0")
def mk(mode, math):
    m = PSMNet(48, 0); m.load_state_dict(state_for("A"), strict=True); m.graph_eval = mode; m.regressor_math = math
    return m.to(dev).eval()
for math in ("f32", "auto"):
    for n in (2, 16):
        me, mg = mk(False, math), mk(True, math)
        fl, fr = synth.synth_features(n, 32, 28, 28, tag="caseA")
        with torch.no_grad():
            a = me.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
            b = mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
            c = mg.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
        print(math, n, "first-call diff", (a - b).abs().max().item(), "replay diff", (a - c).abs().max().item())
        key = [k for k in me._rt._ws if k[0].startswith("3d")][0]
        te, tg = me._rt._ws[key]["t"], mg._rt._ws[key]["t"]
        for name in te:
            x, y = te[name], tg.get(name) if hasattr(tg, "get") else tg[name]
            if y is None: continue
            xs = x.storage if hasattr(x, "storage") and not torch.is_tensor(x) else x
            ys = y.storage if hasattr(y, "storage") and not torch.is_tensor(y) else y
            d = (xs.float() - ys.float()).abs().max().item()
            if d > 0: print("   differs:", name, d)
        print("   plans:", {k: v.kname for k, v in list(me._rt._ws[key]["p"].items())[:6]})
