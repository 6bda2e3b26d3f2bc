"""Development tool: time the rb kernel (default lib or DRC_LIB variant) on the three layer shapes; no checks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E

def one(N, cin, cout, dims, res, relu, reps=20):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = E.Blocked(N, cin, *dims, 1, 1, 1, dev)
    x.from_dense(torch.randn(N, cin, *dims, device=dev))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.randn(cout, device=dev)
    r = None
    if res:
        r = E.Blocked(N, cout, *dims, 1, 1, 1, dev)
        r.from_dense(torch.randn(N, cout, *dims, device=dev))
    E.WINO["rb"] = os.environ.get("RB", "1") != "0"
    E.WINO["rb_min_chunks"] = 1
    y = E.Blocked(N, cout, *dims, 1, 1, 1, dev)
    plan = E.plan_conv3d(x, y, 1, cout, relu)
    wp = E.pack_weight(w)
    w16 = plan.pack16(w)
    for _ in range(3):
        plan.run(x, wp, sc, sh, y, r, w16=w16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run(x, wp, sc, sh, y, r, w16=w16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ex = plan.flops * 64 / 216
    return f"{us:8.1f} us {ex/us/1e6/157.3:.3f}"

if __name__ == "__main__":
    N = int(os.environ.get("N", "256"))
    name = os.environ.get("DRC_LIB", "default").split("/")[-1]
    print(f"{name:32s} 32->32: {one(N, 32, 32, (12, 28, 28), True, True)} | 64->32: {one(N, 64, 32, (12, 28, 28), False, True)} | 64->64h: {one(N, 64, 64, (6, 14, 14), False, True)}", flush=True)
