"""Development tool: time layers.nms / nms_pair on RPN-sized proposal sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd.layers import nms, nms_pair
dev = torch.device("cuda:0")
for n in (1000, 6000, 12000, 30000):
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(n, 2, generator=g) * torch.tensor([1200.0, 330.0]); wh = 16 + torch.rand(n, 2, generator=g) * torch.tensor([200.0, 120.0])
    d = torch.cat((xy, xy + wh), 1).to(dev); s = torch.rand(n, generator=g).to(dev)
    for _ in range(3): k = nms(d, s, 0.7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): k = nms(d, s, 0.7)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n}: nms {e0.elapsed_time(e1)*100:.1f} us per call, kept {len(k)}")
