"""Development tool: per-tensor error of the fp16-storage feature CNN against the fp32 HIP path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from tests.helpers import state_for
sd = state_for('B')
def model(storage, feat):
    m = PSMNet(48, -48); m.load_state_dict(sd, strict=True); m.regressor_storage = storage; m.feature_storage = feat
    return m.to(dev).eval()
left, right = synth.synth_images(4, 224, 224, tag="feat16")
m32, m16, m16f = model("f32", "f32"), model("f16", "f32"), model("f16", "f16")
with torch.no_grad():
    ref = m32((left.to(dev), right.to(dev))).cpu()
    a = m16((left.to(dev), right.to(dev))).cpu()
    b = m16f((left.to(dev), right.to(dev))).cpu()
print("disp err f16-3D only: mean", (a - ref).abs().mean().item(), " f16 all: mean", (b - ref).abs().mean().item())
ws16 = [w for k, w in m16f._rt._ws.items() if k[0] == "2d16"][0]
ws32 = m32._rt._ws[("2d", 8, 224, 224)]
for name in ws32["t"]:
    if name not in ws16["t"]:
        continue
    x32, x16 = ws32["t"][name], ws16["t"][name]
    if isinstance(x32, E.BlockedSlice) or isinstance(x16, E.Blocked16Slice):
        continue
    d32 = x32.to_dense()[:, :, 0].cpu(); d16 = x16.to_dense()[:, :, 0].cpu()
    print(f"{name:24s} range {d32.abs().max().item():9.3f}  max err {(d16 - d32).abs().max().item():9.4f}  mean err {(d16 - d32).abs().mean().item():9.5f}")
d32 = ws32["t"]["feat"].to_dense()[:, :, 0].cpu(); d16 = ws16["t"]["feat"].to_dense()[:, :, 0].cpu()
e = (d16 - d32).abs()
print("feat err: border rows", e[:, :, :2].mean().item(), e[:, :, -2:].mean().item(), "border cols", e[:, :, :, :2].mean().item(), e[:, :, :, -2:].mean().item(),
      "interior", e[:, :, 8:-8, 8:-8].mean().item(), "per-unit", [round(e[i].mean().item(), 5) for i in range(e.shape[0])])
# the same fp16 features fed to the f16 regressor through the fp32->f16 cost volume path (rounding only), vs the fp16-feature path
de = (b - ref).abs()
print("disp err by region: border", de[:, :8].mean().item(), de[:, -8:].mean().item(), de[:, :, :8].mean().item(), de[:, :, -8:].mean().item(), "interior", de[:, 32:-32, 32:-32].mean().item())
print("disp err per ROI", [round(de[i].mean().item(), 3) for i in range(de.shape[0])], "f16-3D only per ROI", [round((a - ref).abs()[i].mean().item(), 3) for i in range(4)])
# inject the fp32 features rounded to f16 into the fp16-feature pipeline: isolates the 2D CNN's error from the plumbing
from disprcnn_amd.modeling.psmnet import runtime as R
rt = m16f._rt
f16t = ws16["t"]["feat"]
v = f16t.view6()
v32 = ws32["t"]["feat"].view6()     # [N, 2 blocks of 16, 1, Hp, Wp, 16]
N_, _, _, Hp, Wp, _ = v32.shape
v[:, 0, 0] = v32[:, :, 0].permute(0, 2, 3, 1, 4).reshape(N_, Hp, Wp, 32).half()
ws3 = [w for k, w in rt._ws.items() if k[0] == "3d16"][0]
with torch.no_grad():
    E.cost_volume16_from16(f16t, 4, ws3["t"]["cost"], -12, 12)
    c = rt._heads(rt._regress16(ws3, rt._compile()), 4, 224, 224, 48, -48, False).cpu()
print("disp err with fp32 features rounded to f16 through the new cost-volume kernel:", (c - ref).abs().mean().item())
