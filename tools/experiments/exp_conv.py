"""Micro-benchmark of one tapconv layer (development tool; not part of the product or the tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E

def run(N, cin, cout, dims, stride=1, deconv=False, reps=20):
    dev = torch.device("cuda:0")
    x = E.Blocked(N, cin, *dims, 1, 1, 1, dev)
    x.view6().normal_()
    od = tuple(2 * d for d in dims) if deconv else tuple(-(-d // stride) for d in dims)
    y = E.Blocked(N, cout, *od, 1, 1, 1, dev)
    plan = E.plan_deconv3d(x, y, cout, True) if deconv else E.plan_conv3d(x, y, stride, cout, True)
    w = torch.randn((cin, cout, 3, 3, 3) if deconv else (cout, cin, 3, 3, 3), device=dev) * 0.05
    wp = E.pack_weight(w, deconv)
    w16 = plan.pack16(w, deconv)
    sc = torch.ones(wp.shape[3], device=dev); sh = torch.zeros(wp.shape[3], device=dev)
    for _ in range(3):
        plan.run(x, wp, sc, sh, y, w16=w16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run(x, wp, sc, sh, y, w16=w16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{os.environ.get('DRC_LIB','default').split('/')[-1]:28s} N={N} {cin}->{cout} {dims} s{stride} dc={deconv} {plan.kname} R={plan.p.R} WT={plan.p.WT}: {us:8.1f} us  {plan.flops/us/1e6:6.1f} TF")

if __name__ == "__main__":
    N = int(os.environ.get("N", "128"))
    if os.environ.get("SLIDE") is not None:
        E.SLIDE["enabled"] = os.environ["SLIDE"] != "0"
    if os.environ.get("SLIDE_SLOTS"):
        E.SLIDE["max_slots"] = int(os.environ["SLIDE_SLOTS"])
    if os.environ.get("DN_TILE"):
        E.DOWN["tile"] = tuple(int(v) for v in os.environ["DN_TILE"].split(","))
        run(N, 32, 64, (12, 28, 28), stride=2)
        run(N, 64, 64, (6, 14, 14), stride=2)
        sys.exit(0)
    if os.environ.get("DC_TILE"):
        E.DECONV_TILE = tuple(int(v) for v in os.environ["DC_TILE"].split(","))
        run(N, 64, 32, (6, 14, 14), deconv=True)
        sys.exit(0)
    if os.environ.get("MAX_SLOTS"):
        E.MAX_SLOTS = int(os.environ["MAX_SLOTS"])
    if os.environ.get("DECONV_DIRECT"):
        E.DECONV_DIRECT["enabled"] = os.environ["DECONV_DIRECT"] != "0"
    if os.environ.get("DC_CT"):
        E.DECONV_DIRECT["ct"] = int(os.environ["DC_CT"])
    if os.environ.get("DC_B"):
        for n_ in ((int(os.environ["DC_N"]),) if os.environ.get("DC_N") else (16, 64)):
            run(n_, 64, 64, (6, 14, 14), deconv=True)
            run(n_, 64, 32, (12, 28, 28), deconv=True)
        sys.exit(0)
    if os.environ.get("WINO"):
        E.WINO["enabled"] = os.environ["WINO"] != "0"
    if os.environ.get("DIRECT"):
        E.DIRECT["enabled"] = os.environ["DIRECT"] != "0"
    if os.environ.get("DN_MIN_GROUPS"):
        E.DOWN["min_groups"] = int(os.environ["DN_MIN_GROUPS"])
    if os.environ.get("DIRECT_DOWN"):
        E.DIRECT["down"] = os.environ["DIRECT_DOWN"] != "0"
    if os.environ.get("SLIDE_MIN_OD"):
        E.SLIDE["min_od"] = int(os.environ["SLIDE_MIN_OD"]); E.SLIDE["min_share"] = 1
    if os.environ.get("SLIDE_MIN_UNITS"):
        E.SLIDE["min_units"] = int(os.environ["SLIDE_MIN_UNITS"])
    if os.environ.get("SLIDE_CT"):
        E.SLIDE["ct"] = int(os.environ["SLIDE_CT"])
    if os.environ.get("QUARTER"):
        run(N, 64, 64, (3, 7, 7))
        sys.exit(0)
    run(N, 32, 32, (12, 28, 28))
    if os.environ.get("ALL"):
        run(N, 64, 32, (12, 28, 28))
        run(N, 32, 64, (12, 28, 28), stride=2)
        run(N, 64, 64, (6, 14, 14))
        run(N, 64, 64, (6, 14, 14), stride=2)
        run(N, 64, 64, (3, 7, 7))
        run(N, 64, 64, (3, 7, 7), deconv=True)
        run(N, 64, 32, (6, 14, 14), deconv=True)
        run(16, 32, 32, (24, 56, 56))


def run_cout1(N, dims, reps=20):
    dev = torch.device("cuda:0")
    x = E.Blocked(N, 32, *dims, 1, 1, 1, dev)
    x.view6()[:, :, 1:-1, 1:-1, 1:-1].normal_()
    w = E.pack_weight_cout1(torch.randn(1, 32, 3, 3, 3, device=dev) * 0.05)
    out = torch.empty(N, *dims, device=dev)
    res = torch.randn(N, *dims, device=dev)
    for _ in range(3):
        E.conv3d_cout1(x, w, res, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        E.conv3d_cout1(x, w, res, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byts = x.numel * 4
    print(f"cout1 N={N} {dims}: {us:8.1f} us  input {byts/1e6:.0f} MB -> {byts/us/1e6:.2f} TB/s (input bytes / time)")
