"""Development tool: per-step s_memtime timeline of the two waves of one SIMD in the rb kernel (needs a -DRB_TRACE variant lib built from attic/wino3d_rb_abl.hip)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from disprcnn_amd import _lib
_lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E
dev = torch.device("cuda:0")
N, cin, cout, dims = 256, 32, 32, (12, 28, 28)
x = E.Blocked(N, cin, *dims, 1, 1, 1, dev); x.from_dense(torch.randn(N, cin, *dims, device=dev))
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
E.WINO["rb"] = True; E.WINO["rb_min_chunks"] = 1
y = E.Blocked(N, cout, *dims, 1, 1, 1, dev)
plan = E.plan_conv3d(x, y, 1, cout, True)
wp = E.pack_weight(w); w16 = plan.pack16(w)
for _ in range(3):
    plan.run(x, wp, sc, sh, y, None, w16=w16)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["DRC_LIB"])
buf = np.zeros((2, 64, 16), dtype=np.uint64)
st = lib.drc_rb_trace_read(buf.ctypes.data_as(C.c_void_p))
assert st == 0, st
names = ["h0 start", "fill issued", "consumed", "item A finished", "barrier passed", "fill issued", "consumed", "item B finished", "phase end done", "barrier passed", "-", "-"]
t = buf.astype(np.int64)
for wv in range(2):
    d = np.diff(t[wv, :, :12], axis=1)
    step = t[wv, 1:, 0] - t[wv, :-1, 0]
    print(f"wave {wv*4}: mean cycles per step {step.mean():.0f} (min {step.min()}, max {step.max()})")
    for k in range(11):
        print(f"   {names[k]:24s} -> {names[k+1]:24s}: mean {d[:, k].mean():8.0f}  median {np.median(d[:, k]):8.0f}")
    print("   odd steps (cb=1, phase end):", np.round(d[1::2].mean(axis=0)).astype(int).tolist())
    print("   even steps (cb=0)          :", np.round(d[0::2].mean(axis=0)).astype(int).tolist())
print("wave4 - wave0 start skew per step (cycles):", (t[1, :8, 0] - t[0, :8, 0]).tolist())
