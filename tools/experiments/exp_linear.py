"""Development tool: drc_linear_fwd vs torch.addmm (hipBLASLt) on the stereo box head's fully connected layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import _lib
if os.environ.get("DRC_LIB"):
    _lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd.modeling.head_ops import linear
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, N, K in ((300, 2048, 25088), (600, 2048, 25088), (300, 2048, 2048), (300, 12, 2048), (100, 1024, 256)):
    lin = torch.nn.Linear(K, N).to(dev)
    x = torch.randn(M, K, device=dev)
    a = t(lambda: linear(x, lin, relu=True))
    b = t(lambda: torch.relu_(torch.addmm(lin.bias, x, lin.weight.t())))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: drc_linear {a:8.1f} us ({fl/a/1e6:6.1f} TF) | addmm+relu {b:8.1f} us ({fl/b/1e6:6.1f} TF)")
