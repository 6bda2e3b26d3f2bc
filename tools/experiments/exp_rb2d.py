"""Development tool: the 2D row-brick Winograd kernel vs wino2d on the PSMNet feature CNN's layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
def one(N, cin, cout, hw, res, reps=20):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = E.Blocked(N, cin, 1, *hw, 0, 1, 1, dev); x.from_dense(torch.randn(N, cin, 1, *hw, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev)
    r = None
    if res:
        r = E.Blocked(N, cout, 1, *hw, 0, 1, 1, dev); r.from_dense(torch.randn(N, cout, 1, *hw, device=dev))
    out = {}
    for rb in (False, True):
        E.WINO2D["rb"] = rb; E.WINO2D["rb_min_chunks"] = 1
        y = E.Blocked(N, cout, 1, *hw, 0, 1, 1, dev)
        plan = E.plan_conv2d(x, y, 3, 1, 1, 1, cout, True)
        wp = E.pack_conv_weight(w); w16 = plan.pack16(w)
        for _ in range(3): plan.run(x, wp, sc, sh, y, r, w16=w16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan.run(x, wp, sc, sh, y, r, w16=w16)
        e1.record(); torch.cuda.synchronize()
        out[rb] = (plan.kname, e0.elapsed_time(e1) * 1e3 / reps, y.to_dense().clone(), plan.flops)
    (k0, u0, y0, fl), (k1, u1, y1, _) = out[False], out[True]
    print(f"N={N} {cin}->{cout} {hw} res={res}: {k0} {u0:7.1f} us ({fl/u0/1e6:5.1f} TF direct-equiv) | {k1} {u1:7.1f} us ({fl/u1/1e6:5.1f} TF, exec {fl*16/36/u1/1e6/157.3:.3f} of peak) {'BIT-EXACT' if torch.equal(y0, y1) else 'DIFFERENT %.3e' % (y0-y1).abs().max().item()}", flush=True)
if __name__ == "__main__":
    N = int(os.environ.get("N", "32"))
    one(N, 32, 32, (112, 112), True)
    one(N, 64, 64, (56, 56), True)
    one(N, 128, 128, (56, 56), True)
    one(N, 320, 128, (56, 56), False)
