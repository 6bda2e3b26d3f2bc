"""Development tool: per-step s_memtime timeline of the two waves of one SIMD in the ping-pong rb kernel (needs a -DRB_TRACE variant lib)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from disprcnn_amd import _lib
_lib.LIB_PATH = os.environ["DRC_LIB"]
from disprcnn_amd import engine as E
dev = torch.device("cuda:0")
N, cin, cout, dims = 256, int(os.environ.get("CIN", "32")), 32, (12, 28, 28)
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    plan.run(x, wp, sc, sh, y, None, w16=w16)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
lib = C.CDLL(os.environ["DRC_LIB"])
buf = np.zeros((2, 64, 16), dtype=np.uint64)  # low 32 bits of s_memtime per mark
st = lib.drc_rb_trace_read(buf.ctypes.data_as(C.c_void_p))
assert st == 0, st
names = ["L0", "bar", "M0", "vmw", "bar", "L1", "bar", "M1", "vmw", "bar"]
t = (buf & 0xffffffff).astype(np.int64)
cb = cin // 16
print(f"{os.environ['DRC_LIB'].split('/')[-1]}: {us:.0f} us per launch (traced build)")
for wv in range(2):
    d = np.diff(t[wv, :, :11], axis=1) % (1 << 32)
    step = (t[wv, 1:, 0] - t[wv, :-1, 0]) % (1 << 32)
    if os.environ.get("VERBOSE"):
        for k in range(10):
            print(f"   {names[k]:6s}: mean {d[:, k].mean():7.0f}  first-of-phase steps {d[0::cb, k].mean():7.0f}  others {d[1::cb, k].mean() if cb > 1 else 0:7.0f}")
    f = d[0::cb].mean(axis=0); o = d[1::cb].mean(axis=0) if cb > 1 else f * 0
    fmt = lambda v: " ".join(f"{n}={x:5.0f}" for n, x in zip(names, v))
    print(f" wave {wv*4}: step {step.mean():6.0f} | PE steps: {fmt(f)} | others: {fmt(o)}")
    m0 = t[wv, :, [2, 11, 12, 13, 14, 15, 3]].T
    dm = np.diff(m0, axis=1) % (1 << 32)
    print("   inside M0: MFMA 0-3 (fma pieces) | 4-9 (w butterfly + LDS writes) | 10-13 (ring copies) | 14-21 (4 loads) | 22-29 (4 loads) | 30-31:", np.round(dm.mean(axis=0)).astype(int).tolist())
if os.environ.get("RAW"):
    base = t[0, 0, 0]
    for s_ in range(int(os.environ["RAW"])):
        for wv in range(2):
            print(f"step {s_} wave {wv*4}: " + " ".join(f"{int((t[wv, s_, k] - base) % (1 << 32)):6d}" for k in range(11)))
per = 4 * cb
for wv in range(2):
    d = np.diff(t[wv, :, :11], axis=1) % (1 << 32)
    # stepno of row k = 8 + k
    for ph in range(per):
        rows = [k for k in range(64) if (8 + k) % per == ph]
        print(f"   wave {wv*4} step%{per}={ph}: L0={d[rows, 0].mean():6.0f} M0={d[rows, 2].mean():6.0f} L1={d[rows, 5].mean():6.0f} M1={d[rows, 7].mean():6.0f}")
