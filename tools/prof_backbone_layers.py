"""Development tool: per-plan timing of the R-50-FPN trunk at 2x3x375x1242."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from disprcnn_amd import engine as E, _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd.modeling.backbone import build_backbone
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
if os.environ.get("W2D_MIN_CHUNKS"):
    E.WINO2D["min_chunks"] = int(os.environ["W2D_MIN_CHUNKS"])
if os.environ.get("T2D_TILE"):
    E.TAP2D["tile"] = tuple(int(v) for v in os.environ["T2D_TILE"].split(","))
bb = build_backbone(NS(MODEL=NS(BACKBONE=NS(CONV_BODY="R-50-FPN"), RESNETS=NS(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256))))
bb.load_state_dict(synth.synth_backbone_state(bb.state_dict()))
bb = bb.to(dev).eval()
pair = synth.hash_uniform("benchpair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)
with torch.no_grad():
    for _ in range(2):
        bb(pair)
    E.TIMING = []
    for _ in range(3):
        bb(pair)
    torch.cuda.synchronize()
rt = bb._rt if hasattr(bb, "_rt") else None
agg = collections.OrderedDict()
names = []
ws = list(rt._ws.values())[0]
order = ["stem"] + [s[0] for s in ws["sched"]] + [f[1] for f in ws["fsched"] if f[0] == "conv"]
per = len(E.TIMING) // 3
for i, (kname, flops, e0, e1) in enumerate(E.TIMING):
    nm = order[i % per]
    pl = ws["p"][nm]
    k0 = pl.p.cls[0]
    key = f"k{k0.nh}x{k0.nw} s{pl.p.in_mul}"
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
tot = sum(a[1] for a in agg.values())
for k, a in agg.items():
    print(f"{k:10s} calls/iter {a[0]//3:3d}  ms/iter {a[1]/3:6.2f}  {100*a[1]/tot:5.1f}%  {a[2]/a[1]/1e9:6.1f} TF")
print("total conv ms/iter", tot / 3)
# per-layer list (sorted by time): name, kernel, geometry, us, TF
rows = collections.OrderedDict()
for i, (kname, flops, e0, e1) in enumerate(E.TIMING):
    nm = order[i % per]
    r = rows.setdefault(nm, [kname, 0.0, flops])
    r[1] += e0.elapsed_time(e1) / 3
if os.environ.get("LAYERS"):
    for nm, (kname, ms, flops) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(os.environ["LAYERS"])]:
        pl = ws["p"][nm]
        print(f"{nm:28s} {kname:30s} N={pl.p.N} {pl.p.cb_in*16:4d}->{pl.p.cout_pad:4d} {pl.p.OH}x{pl.p.OW} R={pl.p.R} WT={pl.p.WT}: {ms*1e3:7.1f} us {flops/ms/1e9:6.1f} TF")
