"""A/B of the hourglass kernels' forms at the bench's shapes (Config A, 1024 and 256 ROIs; Config B, 16 and 64 ROIs): the product forms
(lo4 = 0: row-major tile lanes, de-interleaved slab rows and the cout split of convs16d.hip) against the alternatives (lo4 bits 0x100 the
bank-conflict-free tile lanes of s16_tilemap.h, 0x200 interleaved rows, 0x400 no cout split).  Prints us per launch and the results' identity
(the forms compute the same sums in the same order: bit-identical outputs expected).  Results of the run that set the defaults:
profiles/r5_exp_s16_forms.log (there 0x100 still meant ROW-MAJOR lanes: the grouped order was the default under test).
    python tools/experiments/exp_s16_forms.py            (on an MI355X)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E  # noqa: E402
from disprcnn_amd import s16  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


# (kind, cin, cout, input dims, residual, forms)
A = [("s2", 32, 64, (12, 28, 28), False, (0, 0x100, 0x200, 0x400, 0x600, 0x700)),       # hourglass conv1
     ("s1", 64, 64, (6, 14, 14), True, (0, 0x100)),                                      # conv2
     ("s2", 64, 64, (6, 14, 14), False, (0, 0x100, 0x200, 0x300)),                       # conv3
     ("s1", 64, 64, (3, 7, 7), False, (0, 0x100)),                                       # conv4
     ("up", 64, 64, (3, 7, 7), True, (0, 0x100)),                                        # conv5
     ("up", 64, 32, (6, 14, 14), True, (0, 0x100))]                                      # conv6
B = [("s2", 32, 64, (24, 56, 56), False, (0, 0x200, 0x400, 0x600)),
     ("s1", 64, 64, (12, 28, 28), True, (0,)),
     ("s2", 64, 64, (12, 28, 28), False, (0, 0x200)),
     ("s1", 64, 64, (6, 14, 14), False, (0, 0x100)),
     ("up", 64, 64, (6, 14, 14), True, (0, 0x100)),
     ("up", 64, 32, (12, 28, 28), True, (0,))]

for tag, layers, batches in (("A", A, (1024, 256)), ("B", B, (64, 16))):
    for N in batches:
        for kind, cin, cout, (D, H, W), with_res, forms in layers:
            g = torch.Generator(device=dev).manual_seed(N + cin)
            x = torch.randn(N, cin, D, H, W, generator=g, device=dev)
            w = torch.randn(*((cin, cout) if kind == "up" else (cout, cin)), 3, 3, 3, generator=g, device=dev) * 0.05
            od = {"s1": (D, H, W), "s2": (D // 2, H // 2, W // 2), "up": (2 * D, 2 * H, 2 * W)}[kind]
            wp, wexp = s16.pack_weight_s16(w.transpose(0, 1).contiguous() if kind == "up" else w)
            sc = torch.full((cout,), 2.0 ** -wexp, device=dev)
            sh = torch.zeros(cout, device=dev)
            x16 = E.RS16(N, cin, D, H, W, 1, dev).from_dense(x)
            r16 = E.RS16(N, cout, *od, 1, dev).from_dense(torch.randn(N, cout, *od, generator=g, device=dev)) if with_res else None
            del x
            plan = E.ConvPlanS16(N, cin, cout, D, H, W, True, device=dev, kind=kind)
            out, base = [], None
            for f in forms:
                y16 = E.RS16(N, cout, *od, 1, dev)
                us = timeit(lambda: plan.run(x16, wp, sc, sh, y16=y16, res=r16, lo4=f))
                same = ""
                if base is None:
                    base = y16.storage.clone()
                else:
                    same = " =" if torch.equal(base, y16.storage) else " DIFFERENT"
                out.append(f"{f:#05x}: {us:7.1f} us{same}")
                del y16
            print(f"{tag} N={N} {kind} {cin}->{cout} in {D}x{H}x{W} ({plan.kname}):  " + "   ".join(out), flush=True)
            del x16, r16, base
