"""Round 6 A/B: idle MFMA columns (lanes 28..31 of a 28-voxel row tile) over-reading their neighbours' voxels (default) against reading zeros
(p.dil bit 12 of the KW = 2 forms of convs16.hip): same launch, interleaved rounds, outputs bit-identical (idle columns are never stored).
The plain 32 -> 32 layer and the fused-head form at 1024 Config-A ROIs, activations post-ReLU like in the bench.
Result (round 6, one box, five interleaved rounds, us per launch): plain 1278-1297 over-reading | 1260-1284 zeros; fused head 1395-1415 | 1385-1418 --
no difference beyond the run-to-run spread; the switch (ZIDLE, a 48 KiB zero region behind the exchange buffers) was removed from convs16.hip
again (git history: "zidle"), so this script documents the measurement and needs that variant to run."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from disprcnn_amd import engine as E, s16, _lib
from disprcnn_amd._lib import DrcS16ConvParams
dev = torch.device("cuda:0")
lib = _lib.lib()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
N, D, H, W = 1024, 12, 28, 28
g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(32, 32, 3, 3, 3, generator=g, device=dev) * 0.06
wp, wexp = s16.pack_weight_s16(w)
sc = torch.full((32,), 2.0 ** -wexp, device=dev); sh = torch.randn(32, generator=g, device=dev) * 0.1
x = E.RS16(N, 32, D, H, W, 1, dev).from_dense(torch.randn(N, 32, D, H, W, generator=g, device=dev).relu())
hp, _ = s16.pack_head_weight_s16(torch.randn(1, 32, 3, 3, 3) * 0.2)
hp = hp.to(dev)
for form in ("plain", "head"):
    ys = [E.RS16(N, 32, D, H, W, 1, dev) for _ in range(2)] if form == "plain" else [torch.zeros(N, D, H, W, 12, device=dev) for _ in range(2)]
    def mk(i, dil):
        if form == "plain":
            return DrcS16ConvParams(P(x.storage), P(wp), P(sc), P(sh), None, P(ys[i].storage), None, None, None, N, D, H, W, 32, 32, 1, 0, dil)
        return DrcS16ConvParams(P(x.storage), P(wp), P(sc), P(sh), None, None, None, None, None, N, D, H, W, 32, 32, 1, 0, dil, P(ys[i]), P(hp))
    pa, pb = mk(0, 1), mk(1, 1 | 0x1000)
    assert lib.drc_conv3d_k3_s16_fwd(C.byref(pa), st()) == 0 and lib.drc_conv3d_k3_s16_fwd(C.byref(pb), st()) == 0
    torch.cuda.synchronize()
    same = torch.equal(ys[0].storage if form == "plain" else ys[0], ys[1].storage if form == "plain" else ys[1])
    ta, tb = [], []
    for _ in range(5):
        ta.append(timeit(lambda: lib.drc_conv3d_k3_s16_fwd(C.byref(pa), st())))
        tb.append(timeit(lambda: lib.drc_conv3d_k3_s16_fwd(C.byref(pb), st())))
    print(f"{form}: over-read {' '.join(f'{t:.0f}' for t in ta)} us | zeros {' '.join(f'{t:.0f}' for t in tb)} us | identical={same}", flush=True)
