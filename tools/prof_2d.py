"""Development tool: the 2D stage (DispRCNN, R-50-FPN, synthetic weights) on one KITTI-sized stereo pair, for rocprofv3 / host timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disprcnn_amd.modeling.detector import DispRCNN, default_cfg_2d
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
m = DispRCNN(default_cfg_2d("R-50-FPN"))
sd = m.state_dict()
heads = synth.synth_det_state({k: v for k, v in sd.items() if not k.startswith("backbone.")},
                              gain={("rpn." if k.startswith("head.") else "roi_heads.") + k: v for k, v in synth.DET_GAIN.items()})
bb = synth.synth_backbone_state({k[9:]: v for k, v in sd.items() if k.startswith("backbone.")})
m.load_state_dict({**{"backbone." + k: v for k, v in bb.items()}, **heads}, strict=True)
m = m.to(dev).eval()
pair = synth.hash_uniform("benchpair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)
def run():
    with torch.no_grad():
        return m({"left": pair[:1], "right": pair[1:]})
for _ in range(2):
    run()
torch.cuda.synchronize()
if os.environ.get("STAGES"):
    from disprcnn_amd.structures.image_list import to_image_list
    li = to_image_list(pair[:1])
    def t(fn, n=5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
    with torch.no_grad():
        tb, feats = t(lambda: m.backbone(pair))
        fl, fr = [f[:1] for f in feats], [f[1:] for f in feats]
        th, _ = t(lambda: m.rpn._head(fl, fr))
        td, _ = t(lambda: m.rpn.proposals_dense(li, fl, fr))
        tr, (lp, rp, _) = t(lambda: m.rpn(li, li, fl, fr))
        tx, x = t(lambda: m.roi_heads.box.feature_extractor({"left": fl, "right": fr}, {"left": lp, "right": rp}))
        tbx, (_, ld, rd, _) = t(lambda: m.roi_heads.box({"left": fl, "right": fr}, {"left": lp, "right": rp}))
        tm, _ = t(lambda: m.roi_heads.mask(fl, ld))
    print(f"backbone {tb:.2f}  rpn head {th:.2f}  +proposal kernel {td:.2f}  rpn total {tr:.2f}  box feat {tx:.2f}  box total {tbx:.2f}  mask {tm:.2f} ms;  dets {len(ld[0])}")
t0 = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize()
print("ms/pair", (time.perf_counter() - t0) / 5 * 1e3)
