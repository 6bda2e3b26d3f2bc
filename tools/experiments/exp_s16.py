"""Round-5 experiment: the split-f16 3x3x3 convolution (csrc/convs16.hip) against an fp64 reference, next to the fp32 paths' own error,
and its launch time at the headline batch.  Run on the GPU box:  python tools/experiments/exp_s16.py [--time-only] [--n 1024]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E  # noqa: E402
from disprcnn_amd import s16  # noqa: E402


def ref_costvol(L, R, lo4, D):
    """stackhourglass.py:115-128 restated (the oracle's closed form): [N,2C,D,H,W]."""
    N, Cc, H, W = L.shape
    cost = torch.zeros(N, 2 * Cc, D, H, W, dtype=L.dtype)
    for j in range(D):
        i = lo4 + j
        if i > 0:
            cost[:, :Cc, j, :, i:] = L[:, :, :, i:]
            cost[:, Cc:, j, :, i:] = R[:, :, :, :-i]
        elif i == 0:
            cost[:, :Cc, j] = L
            cost[:, Cc:, j] = R
        else:
            cost[:, :Cc, j, :, :i] = L[:, :, :, :i]
            cost[:, Cc:, j, :, :i] = R[:, :, :, -i:]
    return cost


def stats(name, got, ref):
    err = (got.double() - ref).abs()
    m = ref.abs().max().item()
    print(f"  {name:34s} max|err| {err.max().item():.3e}  mean|err| {err.mean().item():.3e}  rel-to-max {err.max().item() / m:.3e}  (max|ref| {m:.3f})", flush=True)
    return err.max().item() / m


def one_case(dev, N, cin, cout, D, H, W, relu, with_res, cv=False, lo4=0, in_scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(N, cout, D, H, W, generator=g) if with_res else None
    if cv:
        L = torch.randn(N, 32, H, W, generator=g) * in_scale
        R = torch.randn(N, 32, H, W, generator=g) * in_scale
        x = ref_costvol(L, R, lo4, D)
    else:
        x = torch.randn(N, cin, D, H, W, generator=g) * in_scale
    ref = F.conv3d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    print(f"case N={N} cin={cin} cout={cout} D={D} H={H} W={W} relu={relu} res={with_res} cv={cv} lo4={lo4} in_scale={in_scale}", flush=True)
    # fp32 chains: torch CPU (oneDNN) and torch GPU (MIOpen)
    y_cpu = F.conv3d(x, w, padding=1) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    if with_res:
        y_cpu = y_cpu + res
    if relu:
        y_cpu = y_cpu.clamp_min(0)
    stats("fp32 torch CPU", y_cpu, ref)
    # split-f16 kernel
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    sh = shift.to(dev).contiguous()
    y16 = torch.zeros(N, cout // 32, D + 2, H + 2, 8, W + 2, 8, dtype=torch.float16, device=dev)
    yb = E.Blocked(N, cout, D, H, W, 1, 1, 1, dev)
    r16 = s16.rs16_from_dense(res.to(dev)) if with_res else None
    b = 0.0
    if cv:
        l16 = s16.rs16_from_dense(L.to(dev))
        r_16 = s16.rs16_from_dense(R.to(dev))
        s16.conv3d_k3(None, wp, sc, sh, D, H, W, cin, cout, relu, y16=y16, res=r16, left=l16, right=r_16, lo4=lo4)
    else:
        x16 = s16.rs16_from_dense(x.to(dev))
        s16.conv3d_k3(x16, wp, sc, sh, D, H, W, cin, cout, relu, y16=y16, res=r16)
        if cin == 32 and not with_res and W % 28 == 0:
            s16.conv3d_k3(x16, wp, sc, sh, D, H, W, cin, cout, relu, y32=yb.storage)
            torch.cuda.synchronize()
            b = stats("split-f16 kernel, blocked fp32 out", yb.to_dense().cpu(), ref)
    torch.cuda.synchronize()
    a = stats("split-f16 kernel, RS16 output", s16.rs16_to_dense(y16).cpu(), ref)
    halo = y16.clone()
    halo[:, :, 1:D + 1, 1:H + 1, :, 1:W + 1] = 0
    print(f"  halo untouched: {bool((halo == 0).all())}; wexp {wexp}", flush=True)
    return max(a, b)


def timing(dev, N, cin, cout, D, H, W, cv=False, res=False, y32=False, y16=True, iters=10):
    w = torch.randn(cout, cin, 3, 3, 3) * (2.0 / (27 * cin)) ** 0.5
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = torch.full((cout,), 2.0 ** -wexp, device=dev)
    sh = torch.zeros(cout, device=dev)

    def rnd(cb, d, pd):
        t = torch.zeros(N, cb, d + 2 * pd, H + 2, 8, W + 2, 8, dtype=torch.float16, device=dev)
        t[:, :, pd:pd + d, 1:H + 1, :, 1:W + 1].normal_()
        t[:, :, pd:pd + d, 1:H + 1, 4:, 1:W + 1] *= 2.0 ** -11
        return t
    x16 = None if cv else rnd(cin // 32, D, 1)
    l16 = rnd(1, 1, 0) if cv else None
    r_16 = rnd(1, 1, 0) if cv else None
    yo = torch.zeros(N, cout // 32, D + 2, H + 2, 8, W + 2, 8, dtype=torch.float16, device=dev) if y16 else None
    yb = E.Blocked(N, cout, D, H, W, 1, 1, 1, dev).storage if y32 else None
    r16 = rnd(cout // 32, D, 1) if res else None

    def run():
        s16.conv3d_k3(x16, wp, sc, sh, D, H, W, cin, cout, True, y16=yo, y32=yb, res=r16, left=l16, right=r_16, lo4=0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    flops = 2.0 * 27 * cin * cout * N * D * H * W
    med = ts[len(ts) // 2]
    print(f"time N={N} {cin}->{cout} D={D} H={H} W={W} cv={cv} res={res} y16={y16} y32={y32}: median {med * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us  "
          f"= {flops / med / 1e9:.1f} TFLOP/s fp32-equivalent ({3 * flops / med / 1e9:.0f} executed f16)", flush=True)
    return med


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--one", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if not a.time_only and not a.one:
        worst = 0.0
        worst = max(worst, one_case(dev, 2, 32, 32, 12, 28, 28, True, False))
        worst = max(worst, one_case(dev, 3, 32, 32, 12, 28, 28, False, True, seed=1))
        worst = max(worst, one_case(dev, 2, 32, 32, 12, 28, 28, False, False, in_scale=1e-3, seed=2))
        worst = max(worst, one_case(dev, 9, 32, 32, 6, 4, 56, True, True, seed=3))
        worst = max(worst, one_case(dev, 2, 64, 32, 12, 28, 28, True, False, seed=4))
        worst = max(worst, one_case(dev, 2, 64, 64, 6, 28, 28, True, True, seed=5))
        worst = max(worst, one_case(dev, 2, 64, 32, 12, 28, 28, True, False, cv=True, lo4=0, seed=6))
        worst = max(worst, one_case(dev, 2, 64, 32, 12, 28, 28, True, False, cv=True, lo4=-6, seed=7))
        worst = max(worst, one_case(dev, 1, 64, 32, 24, 56, 56, True, False, cv=True, lo4=-12, seed=8))
        print(f"worst rel-to-max error of the split-f16 kernel over all cases: {worst:.3e}", flush=True)
    N = a.n
    if a.one:
        timing(dev, N, 32, 32, 12, 28, 28)
        return
    timing(dev, N, 32, 32, 12, 28, 28)
    timing(dev, N, 32, 32, 12, 28, 28, res=True)
    timing(dev, N, 32, 32, 12, 28, 28, y32=True, y16=False)
    timing(dev, N, 64, 32, 12, 28, 28)
    timing(dev, N, 64, 32, 12, 28, 28, cv=True)
    timing(dev, 16, 32, 32, 12, 28, 28)
    timing(dev, 64, 32, 32, 24, 56, 56)


if __name__ == "__main__":
    main()
