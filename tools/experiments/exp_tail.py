"""Development tool: time the head kernels (32->1 classifier conv, upsample + softmax + soft-argmin) at the bench batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import ops, _lib, engine as E
from oracle import psmnet_oracle as O
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "1024"))
torch.manual_seed(0)
cost = torch.randn(N, 1, 12, 28, 28, device=dev) * 3
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
us = t(lambda: ops.upsample_softargmin(cost, 48, 0, 112, 112))
got = ops.upsample_softargmin(cost[:4], 48, 0, 112, 112).cpu()
ref = O.upsample_softargmin(cost[:4].cpu(), 48, 0, 112, 112)
err = (got - ref).abs()
print(f"upsample_softargmin N={N}: {us:.1f} us  ({(4*9408+4*12544)*N/us/1e6:.2f} TB/s algorithmic)  err mean {err.mean():.2e} max {err.max():.2e}")
c2 = torch.randn(16, 1, 24, 56, 56, device=dev) * 3
us = t(lambda: ops.upsample_softargmin(c2, 48, -48, 224, 224))
got = ops.upsample_softargmin(c2[:2], 48, -48, 224, 224).cpu(); ref = O.upsample_softargmin(c2[:2].cpu(), 48, -48, 224, 224); err = (got - ref).abs()
print(f"upsample_softargmin B N=16: {us:.1f} us  err mean {err.mean():.2e} max {err.max():.2e}")
