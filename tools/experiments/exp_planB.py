"""Development tool: the kernel each 2D-CNN site of Config B runs (plan names) and their TIMING."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
m = PSMNet(48, -48); m.load_state_dict(synth.synth_state_dict(m.state_dict()), strict=True); m = m.to(dev).eval()
l, r = synth.synth_images(16, 224, 224, tag="benchB"); l, r = l.to(dev), r.to(dev)
with torch.no_grad():
    for _ in range(2): m((l, r))
    E.TIMING = []
    for _ in range(3): m((l, r))
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for k, fl, e0, e1 in E.TIMING:
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:34s} calls/step {n/3:5.1f}  ms/step {t/3:7.3f}")
