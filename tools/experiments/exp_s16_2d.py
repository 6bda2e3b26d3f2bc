"""convs16r.hip (split-f16 2D 3x3 conv, row walk): correctness against fp64 next to the fp32 chain, and timing per layer shape of PSMNet's
feature CNN for the forms of the kernel (cin 64: one tile / two tiles per workgroup; the fp32 Winograd kernel's times for the same layers are in
profiles/r5_configB_kernel_stats.md: wino2d_rb_kernel<14>).      python tools/experiments/exp_s16_2d.py        (on an MI355X)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E  # noqa: E402
from disprcnn_amd import s16  # noqa: E402

dev = torch.device("cuda:0")


def check(N, cin, cout, H, W, relu, with_res, form, seed=0, dil=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(N, cout, H, W, generator=g) if with_res else None

    def chain(dt):
        y = F.conv2d(x.to(dt), w.to(dt), padding=dil, dilation=dil) * scale.to(dt).view(1, -1, 1, 1) + shift.to(dt).view(1, -1, 1, 1)
        if with_res:
            y = y + res.to(dt)
        return y.clamp_min(0) if relu else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = (scale * 2.0 ** -wexp).to(dev).contiguous()
    x16 = E.RS16(N, cin, 1, H, W, 0, dev).from_dense(x.to(dev))
    y16 = E.RS16(N, cout, 1, H, W, 0, dev)
    r16 = E.RS16(N, cout, 1, H, W, 0, dev).from_dense(res.to(dev)) if with_res else None
    s16.conv2d_k3(x16.storage, wp, sc, shift.to(dev), N, H, W, cin, cout, relu, y16.storage, res=None if r16 is None else r16.storage, form=form, dil=dil)
    got = y16.to_dense().cpu()[:, :, 0]
    err = (got.double() - ref).abs().max().item()
    v = y16.view7().clone()
    v[:, :, :, 1:H + 1, :, 1:W + 1] = 0
    halo = bool(v.any())
    bad = (got.double() - ref).abs() > 1e-3
    print(f"N={N} {cin}->{cout} {H}x{W} relu={relu} res={with_res} form={form} dil={dil}: err {err:.3e} (fp32 chain {e32:.3e}) halo_dirty={halo} bad={int(bad.sum())}", flush=True)
    if bad.any():
        idx = bad.nonzero()
        print("   first bad (n, c, y, x):", idx[:4].tolist(), " rows:", sorted(set(idx[:, 2].tolist()))[:12], " cols:", sorted(set(idx[:, 3].tolist()))[:12])
    return err <= 2 * e32 + 1e-6 and not halo


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def bench(N, cin, cout, H, W, with_res, forms, dil=1):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    scale, shift = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    wp, wexp = s16.pack_weight_s16(w)
    sc = (scale * 2.0 ** -wexp).contiguous()
    x16 = E.RS16(N, cin, 1, H, W, 0, dev).from_dense(x)
    y16 = E.RS16(N, cout, 1, H, W, 0, dev)
    r16 = E.RS16(N, cout, 1, H, W, 0, dev).from_dense(torch.randn(N, cout, H, W, device=dev)) if with_res else None
    gf = 2.0 * N * cin * cout * 9 * H * W / 1e9
    out = []
    for form in forms:
        us = timeit(lambda: s16.conv2d_k3(x16.storage, wp, sc, shift, N, H, W, cin, cout, True, y16.storage, res=None if r16 is None else r16.storage, form=form, dil=dil))
        out.append(f"form {form}: {us:7.1f} us ({gf / us * 1e3:6.1f} TF)")
    print(f"N={N} {cin}->{cout} {H}x{W} dil={dil} res={with_res} ({gf:.2f} GF):  " + "   ".join(out), flush=True)


if __name__ == "__main__":
    ok = True
    for case in [(2, 32, 32, 28, 56, True, False, 0), (3, 32, 32, 56, 112, False, True, 0), (9, 64, 64, 28, 28, True, True, 1), (2, 64, 64, 56, 56, True, False, 2),
                 (17, 64, 64, 28, 56, False, True, 2), (2, 64, 128, 28, 28, True, False, 1), (9, 128, 128, 28, 28, False, True, 0), (2, 128, 128, 56, 56, True, False, 0),
                 (33, 64, 64, 28, 28, True, True, 1), (40, 32, 32, 28, 56, True, True, 0)]:
        ok &= check(*case)
    for case in [(2, 128, 128, 56, 56, True, False, 0), (9, 128, 128, 56, 28, False, True, 0), (3, 128, 128, 112, 56, True, True, 0)]:
        ok &= check(*case, dil=2)
    print("ALL OK" if ok else "MISMATCH", flush=True)
    if os.environ.get("BENCH", "1") == "1":
        for N in (32, 128):
            # forms: bit 2 (4) = one workgroup per CU instead of two (the KS == 1 kernels)
            bench(N, 32, 32, 112, 112, True, (0, 4))
            bench(N, 64, 64, 56, 56, True, (1, 5, 2))
            bench(N, 64, 128, 56, 56, False, (1, 5, 2))
            bench(N, 128, 128, 56, 56, True, (0,))
            bench(N, 128, 128, 56, 56, True, (0,), dil=2)
