"""Winograd 3x3x3 kernel against the direct kernel on random layers (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E, ops

torch.manual_seed(0)
dev = torch.device("cuda:0")

E.SLIDE["min_od"], E.SLIDE["min_units"] = 2, 1
worst = 0.0
for (n, cin, cout, d, h, w) in [(1, 16, 16, 2, 2, 2), (2, 32, 32, 4, 6, 6), (3, 20, 40, 6, 4, 10), (5, 64, 32, 12, 28, 28), (2, 64, 64, 6, 14, 14),
                                 (1, 7, 33, 2, 2, 30), (4, 32, 32, 2, 14, 2)]:
    x = torch.randn(n, cin, d, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    res = torch.randn(n, cout, d, h, w, device=dev)
    for relu, r in [(False, None), (True, res)]:
        E.WINO["enabled"] = True
        a = ops.conv3d_bn(x, wt, sc, sh, 1, relu, r)
        E.WINO["enabled"] = False
        b = ops.conv3d_bn(x, wt, sc, sh, 1, relu, r)
        ref = torch.nn.functional.conv3d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
        if r is not None:
            ref = ref + r.double()
        if relu:
            ref = ref.clamp_min(0)
        ea = (a.double() - ref).abs().max().item(); eb = (b.double() - ref).abs().max().item()
        worst = max(worst, ea)
        print(f"{(n, cin, cout, d, h, w)} relu={relu} res={r is not None}: wino err {ea:.2e}  direct err {eb:.2e}  scale {ref.abs().max().item():.1f}", flush=True)
print("worst", worst)
if os.environ.get("ONLY3D"):
    sys.exit(0 if worst < 1e-3 else 1)

# ---- 2D: wino2d against the direct kernel and torch fp64
E.WINO2D["min_chunks"] = 0
worst2 = 0.0
for (n, cin, cout, h, w) in [(2, 16, 16, 2, 2), (3, 32, 32, 6, 10), (2, 20, 40, 8, 4), (5, 7, 33, 2, 30), (4, 32, 32, 112, 112), (4, 64, 64, 56, 56),
                             (3, 128, 128, 56, 56), (9, 64, 32, 4, 4)]:
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    res = torch.randn(n, cout, h, w, device=dev)
    for relu, r in [(False, None), (True, res)]:
        E.WINO2D["enabled"] = True
        a = ops.conv2d_bn(x, wt, sc, sh, 1, 1, 1, relu, r)
        E.WINO2D["enabled"] = False
        b = ops.conv2d_bn(x, wt, sc, sh, 1, 1, 1, relu, r)
        ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
        if r is not None:
            ref = ref + r.double()
        if relu:
            ref = ref.clamp_min(0)
        ea = (a.double() - ref).abs().max().item(); eb = (b.double() - ref).abs().max().item()
        worst2 = max(worst2, ea)
        print(f"2D {(n, cin, cout, h, w)} relu={relu} res={r is not None}: wino err {ea:.2e}  direct err {eb:.2e}", flush=True)
print("worst2d", worst2)


def time2d(n, c, hw, wino):
    E.WINO2D["enabled"] = wino
    x = E.Blocked(n, c, 1, *hw, 0, 1, 1, dev); x.view6().normal_()
    y = E.Blocked(n, c, 1, *hw, 0, 1, 1, dev)
    plan = E.plan_conv2d(x, y, 3, 1, 1, 1, c, True)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
    wp, w16 = E.pack_conv_weight(wt), plan.pack16(wt)
    s1, s0 = torch.ones(E.cout_pad_of(c), device=dev), torch.zeros(E.cout_pad_of(c), device=dev)
    for _ in range(3):
        plan.run(x, wp, s1, s0, y, w16=w16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.run(x, wp, s1, s0, y, w16=w16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"time2d N={n} C={c} {hw} {plan.kname}: {us:8.1f} us  {plan.flops / us / 1e6:6.1f} TF", flush=True)


for (n, c, hw) in [(32, 32, (112, 112)), (32, 64, (56, 56)), (32, 128, (56, 56))]:
    time2d(n, c, hw, True); time2d(n, c, hw, False)
