"""Development tool: randomized sweep of the per-site backward (conv + batch-stat BN (+res) (+ReLU): dx, dres, dw, dgamma, dbeta vs
torch fp64 autograd), reusing the parity test's body.  Run on an MI355X."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_train import test_site_backward_vs_torch_autograd as site_test

dev = torch.device("cuda:0")
rng = random.Random(int(os.environ.get("SEED", "1")))
bad = 0
n_cases = int(os.environ.get("CASES", "40"))
for case in range(n_cases):
    kind = rng.choice(["conv3d", "conv3d", "deconv3d", "conv2d", "conv2d"])
    cin, cout = rng.choice([3, 16, 24, 32, 64]), rng.choice([16, 32, 48, 64])
    relu, with_res = rng.random() < 0.5, rng.random() < 0.5
    if kind == "conv3d":
        stride = rng.choice([1, 1, 2])
        dims = tuple(stride * rng.randint(1, m) for m in (5, 8, 12))
        args = (kind, cin, cout, 3, stride, 1, 1, dims, relu, with_res)
    elif kind == "deconv3d":
        dims = (rng.randint(1, 4), rng.randint(1, 7), rng.randint(1, 9))
        args = (kind, cin, cout, 3, 2, 1, 1, dims, relu, with_res)
    else:
        k = rng.choice([1, 3])
        stride = rng.choice([1, 2])
        dil = rng.choice([1, 2]) if (k == 3 and stride == 1) else 1
        hw = tuple(stride * rng.randint(2, m) for m in (12, 16))
        args = (kind, cin, cout, k, stride, 0 if k == 1 else dil, dil, hw, relu, with_res)
    try:
        site_test(dev, *args)
    except AssertionError as ex:
        bad += 1
        print("FAIL", case, args, str(ex)[:200])
    except Exception as ex:
        bad += 1
        print("ERROR", case, args, repr(ex)[:300])
print(f"{n_cases - bad}/{n_cases} cases ok")
