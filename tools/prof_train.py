"""Development tool: a few Config-A train steps (from the feature boundary) for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):          # development: a variant library from tools/experiments/build_variant2.sh
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
from disprcnn_amd.utils import synth
from disprcnn_amd.utils.loss_utils import PSMLoss
dev = torch.device("cuda:0")
if os.environ.get("WG_CAP"):
    from disprcnn_amd.modeling.psmnet import train as _T
    _T.WGRAD_TILE["lds_cap"] = int(os.environ["WG_CAP"]) * 1024
n = int(os.environ.get("N", "64"))
CFG_B = bool(os.environ.get("CFG_B"))       # Config B: full PSMNet on N 224x224 crops, D = 96 (-48..48)
m = PSMNet(48, -48) if CFG_B else PSMNet(48, 0)
m.load_state_dict(synth.synth_state_dict(m.state_dict()), strict=True)
m = m.to(dev).train()
crit = PSMLoss()
if CFG_B:
    li, ri = synth.synth_images(n, 224, 224, tag="trainB")
    li, ri = li.to(dev), ri.to(dev)
    tgt = synth.hash_uniform("trainB:t", (n, 224, 224), -47.0, 47.0).to(dev)
    msk = torch.ones_like(tgt, dtype=torch.uint8)
    def step():
        for p in m.parameters():
            p.grad = None
        loss = crit(m({"left": li, "right": ri}), {"disparity": tgt, "mask": msk})
        loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
    sys.exit(0)
fl, fr = synth.synth_features(n, 32, 28, 28, tag="trainA")
fl, fr = fl.to(dev), fr.to(dev)
tgt = synth.hash_uniform("trainA:t", (n, 112, 112), 0.0, 47.0).to(dev)
msk = torch.ones_like(tgt, dtype=torch.uint8)
def step():
    for p in m.parameters():
        p.grad = None
    loss = crit(m.forward_from_features(fl, fr, (112, 112)), {"disparity": tgt, "mask": msk})
    loss.backward()
if os.environ.get("FWD_ONLY"):
    for _ in range(6):
        with torch.no_grad():
            m.forward_from_features(fl, fr, (112, 112))
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.no_grad():
    pass
for _ in range(3):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
# forward only
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    with torch.no_grad():
        m.forward_from_features(fl, fr, (112, 112))
torch.cuda.synchronize()
print("fwd-only ms", (time.perf_counter() - t0) / 3 * 1e3)
