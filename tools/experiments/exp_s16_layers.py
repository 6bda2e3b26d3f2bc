"""Round-5 debugging aid: every split-f16 layer kind at Config-A shapes and a given batch against torch's own GPU fp32 convolution."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E
from disprcnn_amd import s16

dev = torch.device("cuda:0")
LAYERS = [("s1", 32, 32, (12, 28, 28)), ("s1", 64, 64, (6, 14, 14)), ("s1", 64, 64, (3, 7, 7)), ("s2", 32, 64, (12, 28, 28)), ("s2", 64, 64, (6, 14, 14)),
          ("up", 64, 64, (3, 7, 7)), ("up", 64, 32, (6, 14, 14))]
for N in [int(a) for a in sys.argv[1:]] or [16, 64]:
    for kind, cin, cout, (D, H, W) in LAYERS:
        g = torch.Generator(device=dev).manual_seed(N + cin)
        x = torch.randn(N, cin, D, H, W, generator=g, device=dev)
        w = torch.randn(*((cin, cout) if kind == "up" else (cout, cin)), 3, 3, 3, generator=g, device=dev) * 0.05
        od = {"s1": (D, H, W), "s2": (D // 2, H // 2, W // 2), "up": (2 * D, 2 * H, 2 * W)}[kind]
        res = torch.randn(N, cout, *od, generator=g, device=dev) if kind != "s2" else None
        if kind == "up":
            ref = F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)
        else:
            ref = F.conv3d(x, w, padding=1, stride=1 if kind == "s1" else 2)
        if res is not None:
            ref = ref + res
        wp, wexp = s16.pack_weight_s16(w.transpose(0, 1).contiguous() if kind == "up" else w)
        sc = torch.full((cout,), 2.0 ** -wexp, device=dev)
        y16 = E.RS16(N, cout, *od, 1, dev)
        plan = E.ConvPlanS16(N, cin, cout, D, H, W, False, device=dev, kind=kind)
        r16 = E.RS16(N, cout, *od, 1, dev).from_dense(res) if res is not None else None
        x16 = E.RS16(N, cin, D, H, W, 1, dev).from_dense(x)
        worst = 0.0
        for rep in range(3):
            y16.storage.zero_()
            plan.run(x16, wp, sc, torch.zeros(cout, device=dev), y16=y16, res=r16)
            err = (y16.to_dense() - ref).abs()
            worst = max(worst, err.max().item())
        bad = (err > 1e-3).nonzero()
        print(f"N={N} {plan.kname}: max|err| {worst:.3e} bad voxels {bad.shape[0]}" + (f" first {bad[0].tolist()} last {bad[-1].tolist()} units {sorted(set(bad[:,0].tolist()))[:20]} z {sorted(set(bad[:,2].tolist()))} y {sorted(set(bad[:,3].tolist()))}" if bad.shape[0] else ""), flush=True)
