"""Winograd 3x3x3 kernel against the direct kernel on random layers (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
