"""A/B of the two-tiles-per-wave kernel (convs16w.hip) against convs16.hip: bit-identity and time, the plain 32->32 layer and the cost-volume layer."""
import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo")
from disprcnn_amd import engine as E, s16, _lib
from disprcnn_amd._lib import DrcS16ConvParams
dev = torch.device("cuda:0")
lib = _lib.lib()
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for N, D, H, W in ((1024, 12, 28, 28), (256, 12, 28, 28), (64, 24, 56, 56), (16, 24, 56, 56), (37, 8, 12, 40)):
    for cv in (False, True):
        cin = 64 if cv else 32
        g = torch.Generator(device=dev).manual_seed(N + D)
        w = torch.randn(32, cin, 3, 3, 3, generator=g, device=dev) * 0.05
        wp, wexp = s16.pack_weight_s16(w)
        sc = torch.full((32,), 2.0 ** -wexp, device=dev); sh = torch.randn(32, generator=g, device=dev) * 0.1
        if cv:
            L = E.RS16(N, 32, 1, H, W, 0, dev).from_dense(torch.randn(N, 32, H, W, generator=g, device=dev).relu())
            R = E.RS16(N, 32, 1, H, W, 0, dev).from_dense(torch.randn(N, 32, H, W, generator=g, device=dev).relu())
            x = None
        else:
            x = E.RS16(N, 32, D, H, W, 1, dev).from_dense(torch.randn(N, 32, D, H, W, generator=g, device=dev).relu())
        ya, yb = E.RS16(N, 32, D, H, W, 1, dev), E.RS16(N, 32, D, H, W, 1, dev)
        def mk(y, lo4, dil=1):
            return DrcS16ConvParams(P(x.storage) if x is not None else None, P(wp), P(sc), P(sh), None, P(y.storage), None, P(L.storage) if cv else None,
                                    P(R.storage) if cv else None, N, D, H, W, cin, 32, 1, lo4, dil)
        pa, pb = mk(ya, -3 if cv else 0), mk(yb, -3 if cv else 0)
        # A: the library's own dispatch (may or may not pick the wide kernel); B: the wide kernel forced; C: the one-tile kernel (experiment bit; for cv: compare A vs B only)
        assert lib.drc_conv3d_k3_s16_fwd(C.byref(pa), st()) == 0
        if not cv:
            print('(the plain form is not instantiated any more: measured 1277 us wide against 1278 us one-tile at 1024 ROIs, round 6)'); continue
        assert lib.drc_conv3d_k3_s16_wide_fwd(C.byref(pb), st()) == 0
        torch.cuda.synchronize()
        same = torch.equal(ya.storage, yb.storage)
        pc = mk(ya, -3 if cv else 0, 0x800)                      # the one-tile kernel (experiment bit)
        ta = tb = tc = 0.0
        for _ in range(3):                                       # interleaved rounds: clock / thermal state hits all three alike
            ta += timeit(lambda: lib.drc_conv3d_k3_s16_fwd(C.byref(pa), st())) / 3
            tb += timeit(lambda: lib.drc_conv3d_k3_s16_wide_fwd(C.byref(pb), st())) / 3
            tc += timeit(lambda: lib.drc_conv3d_k3_s16_fwd(C.byref(pc), st())) / 3
        print(f"N={N} {D}x{H}x{W} cv={cv}: dispatch picks wide={bool(lib.drc_conv3d_k3_s16_wide(C.byref(pa)))}  dispatch {ta:8.1f} us  wide {tb:8.1f} us  one-tile {tc if tc is None else round(tc, 1)} us  identical={same}", flush=True)
