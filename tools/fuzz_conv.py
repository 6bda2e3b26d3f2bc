"""Development tool: randomized shape sweep of the convolution kernels against torch-CPU (run on an MI355X)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from disprcnn_amd import ops, engine as E

dev = torch.device("cuda:0")
if os.environ.get("DIRECT") == "0":
    E.DIRECT["enabled"] = False          # sweep the generic kernel (tapconv.hip) instead
WINO_ONLY = os.environ.get("WINO") == "1"
if WINO_ONLY:
    E.SLIDE["min_od"] = 2
rng = random.Random(int(os.environ.get("SEED", "1")))
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "1")))
bad = 0
n_cases = int(os.environ.get("CASES", "80"))
for case in range(n_cases):
    kind = rng.choice(["3d_s1", "3d_s1", "3d_s2", "deconv", "2d_k3", "2d_k3s2", "2d_k1", "2d_k1s2", "2d_dil2"])
    if WINO_ONLY:
        kind = "3d_s1"
    cin, cout = rng.choice([3, 8, 16, 24, 32, 40, 64, 96]), rng.choice([8, 16, 24, 32, 48, 64, 80, 128])
    n = rng.choice([33, 64, 100, 130]) if os.environ.get("BIGN") else rng.choice([1, 2, 3, 5])   # BIGN: many groups per wave, larger cout tiles per wave
    force_slide = rng.random() < 0.7 or WINO_ONLY
    if WINO_ONLY:
        n = rng.choice([1, 2, 3, 5, 7, 19, 33, 64])
    if kind.startswith("3d") or kind == "deconv":
        if kind == "3d_s2":
            dims = (2 * rng.randint(1, 5), 2 * rng.randint(1, 12), 2 * rng.randint(1, 20)) if not os.environ.get("BIGN") else (2 * rng.randint(1, 3), 2 * rng.randint(1, 5), 2 * rng.randint(1, 8))
        else:
            dims = (rng.randint(1, 9), rng.randint(1, 20), rng.randint(1, 33)) if not os.environ.get("BIGN") else (rng.randint(1, 5), rng.randint(1, 9), rng.randint(1, 15))
        if WINO_ONLY:      # even output dims: the Winograd kernel (tile groups of every fill level, 1..N cout groups)
            dims = (2 * rng.randint(1, 6), 2 * rng.randint(1, 10), 2 * rng.randint(1, 16)) if n < 19 else (2 * rng.randint(1, 3), 2 * rng.randint(1, 5), 2 * rng.randint(1, 7))
        x = torch.randn((n, cin) + dims, generator=g)
        w = torch.randn((cin, cout, 3, 3, 3) if kind == "deconv" else (cout, cin, 3, 3, 3), generator=g) * 0.1
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
        if kind == "deconv":
            ref = F.conv_transpose3d(x, w, None, 2, 1, 1)
        else:
            ref = F.conv3d(x, w, None, 2 if kind == "3d_s2" else 1, 1)
        res = torch.randn(ref.shape, generator=g)
        ref = F.relu(ref * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1) + res)
        saved = E.SLIDE["min_units"]
        if force_slide:
            E.SLIDE["min_units"] = 1
        try:
            got = ops.conv3d_bn(x.to(dev), w.to(dev), sc.to(dev), sh.to(dev), 2 if kind == "3d_s2" else 1, True, res.to(dev), kind == "deconv")
        finally:
            E.SLIDE["min_units"] = saved
    else:
        k = 1 if "k1" in kind else 3
        stride = 2 if kind.endswith("s2") else 1
        dil = 2 if kind == "2d_dil2" else 1
        pad = 0 if k == 1 else dil
        hw = (rng.randint(4, 40), rng.randint(4, 70)) if not os.environ.get("BIGN") else (rng.randint(4, 14), rng.randint(4, 20))
        x = torch.randn((n, cin) + hw, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * 0.1
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
        ref = F.conv2d(x, w, None, stride, pad, dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        res = torch.randn(ref.shape, generator=g)
        ref = F.relu(ref + res)
        got = ops.conv2d_bn(x.to(dev), w.to(dev), sc.to(dev), sh.to(dev), stride, pad, dil, True, res.to(dev), in_halo=max(pad, 1))
    err = (got.cpu() - ref).abs().max().item()
    tol = 2e-5 * ref.abs().max().item() + 1e-5
    ok = err <= tol and got.shape == ref.shape
    if not ok:
        bad += 1
        print("FAIL", case, kind, n, cin, cout, tuple(x.shape[2:]), "slide" if force_slide else "", err, tol)
print(f"{n_cases - bad}/{n_cases} cases ok")
