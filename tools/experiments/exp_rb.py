"""Development tool: wino3d_rb (two waves per SIMD, row brick) vs wino3d -- bit-exactness and time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E


def one(N, cin, cout, dims, res, relu, reps=20, check=True):
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
    out = {}
    for rb in (False, True):
        E.WINO["rb"] = rb
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
        out[rb] = (plan.kname, us, y.to_dense().clone() if check else None, plan.flops)
    (k0, u0, y0, fl), (k1, u1, y1, _) = out[False], out[True]
    msg = ""
    if check:
        d = (y0 - y1).abs().max().item()
        msg = f"max|rb - wino| = {d:.3e} {'BIT-EXACT' if torch.equal(y0, y1) else 'DIFFERENT'} finite={bool(torch.isfinite(y1).all())}"
    ex = fl * 64 / 216
    print(f"N={N} {cin}->{cout} {dims} res={res} relu={relu}: {k0} {u0:8.1f} us ({ex/u0/1e6:5.1f} TF exec) | {k1} {u1:8.1f} us ({ex/u1/1e6:5.1f} TF exec = {ex/u1/1e6/157.3:.3f} of peak)  {msg}", flush=True)


if __name__ == "__main__":
    N = int(os.environ.get("N", "256"))
    one(7, 32, 32, (12, 28, 28), False, False, reps=3)
    one(5, 64, 64, (6, 14, 14), True, True, reps=3)
    one(3, 32, 32, (4, 28, 28), True, False, reps=3)
    one(N, 32, 32, (12, 28, 28), True, True)
    one(N, 64, 32, (12, 28, 28), False, True)
    one(N, 64, 64, (6, 14, 14), False, True)
    if os.environ.get("BIG"):
        one(1024, 32, 32, (12, 28, 28), True, True, check=False)
        one(1024, 64, 64, (6, 14, 14), False, True, check=False)
