"""Development tool: 3x3 Conv2d tile choice on the trunk's small maps (engine.CONV2D_FILL on / off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
dev = torch.device("cuda:0")
for (n, c, h, w, st) in ((2, 512, 12, 39, 1), (2, 256, 24, 78, 1), (2, 256, 47, 155, 2), (2, 512, 24, 78, 2), (600, 256, 14, 14, 1)):
    for en in (False, True):
        E.CONV2D_FILL["enabled"] = en
        x = E.Blocked(n, c, 1, h, w, 0, 1, 1, dev); x.from_dense(torch.randn(n, c, 1, h, w, device=dev))
        oh, ow = (h - 1) // st + 1, (w - 1) // st + 1
        y = E.Blocked(n, c, 1, oh, ow, 0, 1, 1, dev)
        pl = E.plan_conv2d(x, y, 3, st, 1, 1, c, True)
        wt = torch.randn(c, c, 3, 3, device=dev) * 0.02
        wp = E.pack_conv_weight(wt); w16 = pl.pack16(wt)
        sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev)
        for _ in range(3): pl.run(x, wp, sc, sh, y, None, w16=w16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): pl.run(x, wp, sc, sh, y, None, w16=w16)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f"{(n, c, h, w, st)} fill={en}: {pl.kname} R={pl.p.R} WT={pl.p.WT}: {us:7.1f} us {pl.flops/us/1e6:6.1f} TF", flush=True)
