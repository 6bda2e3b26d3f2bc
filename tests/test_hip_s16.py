"""GPU parity of the split-f16 ("f16x2") convolution path (round 5: csrc/convs16.hip, s16_ops.hip, the RS16 epilogue of deconvdirect.hip).

An fp32 value is carried as hi + lo fp16, a product is three f16 MFMAs into an fp32 accumulator.  The claim to hold: fp32-CLASS results.
Every layer case is therefore measured against an fp64 reference NEXT TO the fp32 FMA chain's own error on the same inputs (torch CPU,
oneDNN), and must (a) meet the single-layer bound of test_hip_parity.py (2e-5 * max + 1e-5) with a wide margin and (b) stay within 2x of
the fp32 chain's own maximum error.  Reference arithmetic: disprcnn/modeling/psmnet/submodule.py:19-22, stackhourglass.py:63-70,78-88,115-128.
"""
import pytest
import torch
import torch.nn.functional as F

from disprcnn_amd import engine as E
from disprcnn_amd import s16
from disprcnn_amd.utils import synth
from oracle import psmnet_oracle as O
from tests.helpers import state_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _ref_costvol(L, R, lo4, D):
    N, Cc, H, W = L.shape
    cost = torch.zeros(N, 2 * Cc, D, H, W, dtype=L.dtype)
    for j in range(D):
        i = lo4 + j
        if i > 0:
            cost[:, :Cc, j, :, i:] = L[:, :, :, i:]
            cost[:, Cc:, j, :, i:] = R[:, :, :, :-i]
        elif i == 0:
            cost[:, :Cc, j] = L
            cost[:, Cc:, j] = R
        else:
            cost[:, :Cc, j, :, :i] = L[:, :, :, :i]
            cost[:, Cc:, j, :, :i] = R[:, :, :, -i:]
    return cost


def test_rs16_converters_match_the_layout_definition(dev):
    """drc_rs16_from_dense / _from_blocked / _to_dense against the torch definition of the layout (disprcnn_amd/s16.py)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 64, 4, 6, 10, generator=g) * torch.logspace(-4, 2, 10).view(1, 1, 1, 1, 10)
    want = s16.rs16_from_dense(x)
    t = E.RS16(3, 64, 4, 6, 10, 1, dev).from_dense(x.to(dev))
    assert torch.equal(t.view7().cpu(), want)
    blk = E.Blocked(3, 64, 4, 6, 10, 1, 1, 1, dev).from_dense(x.to(dev))
    t2 = E.RS16(3, 64, 4, 6, 10, 1, dev).from_blocked(blk)
    assert torch.equal(t2.view7().cpu(), want)
    back = t.to_dense().cpu()
    assert torch.equal(back, s16.rs16_to_dense(want))
    assert (back - x).abs().max().item() <= 2.0 ** -22 * x.abs().max().item()
    # 2D maps (no depth halo), a later range of units of a blocked tensor
    f = torch.randn(4, 32, 5, 7, generator=g)
    blk2 = E.Blocked(4, 32, 1, 5, 7, 0, 1, 1, dev).from_dense(f.to(dev))
    m = E.RS16(2, 32, 1, 5, 7, 0, dev).from_blocked(blk2, 2)
    assert torch.equal(m.view7().cpu(), s16.rs16_from_dense(f[2:]))
    # RS16 -> blocked fp32 (the seams of the 2D CNN: convs16r.hip's results handed to fp32 kernels): a whole tensor with another halo, a
    # channel slice of a wider tensor, a later range of units; exactly hi + lo, everything outside the interior untouched
    f2 = torch.randn(3, 64, 6, 9, generator=g)
    r = E.RS16(3, 64, 1, 6, 9, 0, dev).from_dense(f2.to(dev))
    want2 = s16.rs16_to_dense(s16.rs16_from_dense(f2), pd=0)[:, :, 0]
    whole = E.Blocked(3, 64, 1, 6, 9, 0, 2, 2, dev)
    r.to_blocked(whole)
    assert torch.equal(whole.to_dense().cpu()[:, :, 0], want2)
    wide = E.Blocked(5, 160, 1, 6, 9, 0, 1, 1, dev)
    wide.storage.fill_(7.0)
    r.to_blocked(E.BlockedSlice(wide, 2, 64), 1)
    got = wide.to_dense().cpu()[:, :, 0]
    assert torch.equal(got[1:4, 32:96], want2)
    assert (got[0] == 7).all() and (got[4] == 7).all() and (got[:, :32] == 7).all() and (got[:, 96:] == 7).all()


CASES = [
    # N, cin, cout, D, H, W, relu, res, cv, lo4, in_scale
    (2, 32, 32, 12, 28, 28, True, False, False, 0, 1.0),
    (3, 32, 32, 12, 28, 28, False, True, False, 0, 1.0),
    (9, 32, 32, 6, 4, 56, True, True, False, 0, 1.0),
    (2, 32, 32, 3, 2, 28, False, False, False, 0, 30.0),
    (2, 64, 32, 12, 28, 28, True, False, False, 0, 1.0),
    (2, 64, 64, 6, 28, 28, True, True, False, 0, 1.0),
    (2, 64, 32, 12, 28, 28, True, False, True, 0, 1.0),
    (2, 64, 32, 12, 28, 28, True, False, True, -6, 1.0),
    (2, 64, 32, 6, 28, 28, False, False, True, 3, 1.0),
    (1, 64, 32, 24, 56, 56, True, False, True, -12, 1.0),
    # round 6: shapes beyond BASELINE's (VERDICT r5 missing #2; the reference takes any D, H, W = 0 mod 4 here, stackhourglass.py:115-128):
    # depths that are not multiples of 3 (phantom zero planes behind the last one), widths that mask their last 28-wide tile, odd row counts
    (2, 32, 32, 4, 8, 32, True, False, False, 0, 1.0),
    (2, 32, 32, 8, 7, 16, False, True, False, 0, 1.0),
    (9, 32, 32, 16, 5, 40, True, True, False, 0, 1.0),
    (2, 32, 32, 1, 3, 20, True, False, False, 0, 1.0),
    (2, 32, 32, 2, 4, 64, False, False, False, 0, 1.0),
    (2, 64, 64, 20, 6, 36, True, True, False, 0, 1.0),
    (2, 64, 32, 8, 32, 32, True, False, True, 0, 1.0),
    (2, 64, 32, 16, 12, 40, True, False, True, -8, 1.0),
    (3, 64, 32, 4, 16, 16, False, False, True, 2, 1.0),
]


@pytest.mark.parametrize("N,cin,cout,D,H,W,relu,with_res,cv,lo4,in_scale", CASES)
def test_conv3d_s16_vs_fp64_next_to_the_fp32_chain(dev, N, cin, cout, D, H, W, relu, with_res, cv, lo4, in_scale):
    g = torch.Generator().manual_seed(N * 1000 + cin + D + (7 if cv else 0) + lo4)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(N, cout, D, H, W, generator=g) if with_res else None
    if cv:
        L = torch.randn(N, 32, H, W, generator=g) * in_scale
        R = torch.randn(N, 32, H, W, generator=g) * in_scale
        x = _ref_costvol(L, R, lo4, D)
    else:
        x = torch.randn(N, cin, D, H, W, generator=g) * in_scale

    def chain(dt):
        y = F.conv3d(x.to(dt), w.to(dt), padding=1) * scale.to(dt).view(1, -1, 1, 1, 1) + shift.to(dt).view(1, -1, 1, 1, 1)
        if with_res:
            y = y + res.to(dt)
        return y.clamp_min(0) if relu else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    y16 = E.RS16(N, cout, D, H, W, 1, dev)
    plan = E.ConvPlanS16(N, cin, cout, D, H, W, relu, cv=cv, device=dev)
    r16 = E.RS16(N, cout, D, H, W, 1, dev).from_dense(res.to(dev)) if with_res else None
    outs = []
    if cv:
        plan.run(None, wp, sc, shift.to(dev), y16=y16, res=r16, left=E.RS16(N, 32, 1, H, W, 0, dev).from_dense(L.to(dev)),
                 right=E.RS16(N, 32, 1, H, W, 0, dev).from_dense(R.to(dev)), lo4=lo4)
    else:
        x16 = E.RS16(N, cin, D, H, W, 1, dev).from_dense(x.to(dev))
        plan.run(x16, wp, sc, shift.to(dev), y16=y16, res=r16)
        if cin == 32 and not with_res and W > 14:
            # the blocked fp32 output form (the cout-1 head's input): instead of RS16
            y32 = E.Blocked(N, cout, D, H, W, 1, 1, 1, dev)
            plan.run(x16, wp, sc, shift.to(dev), y32=y32)
            outs.append(("blocked fp32", y32.to_dense().cpu()))
            b = y32.view6().clone()
            b[:, :, 1:D + 1, 1:H + 1, 1:W + 1] = 0
            assert not b.any()
    outs.append(("RS16", y16.to_dense().cpu()))
    m = ref.abs().max().item()
    for name, got in outs:
        err = (got.double() - ref).abs().max().item()
        print(f"{name}: max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
        assert err <= 2e-5 * m + 1e-5
        assert err <= 2.0 * e32 + 1e-6 * m, (err, e32)
    # the halo stays zero
    v = y16.view7().clone()
    v[:, :, 1:D + 1, 1:H + 1, :, 1:W + 1] = 0
    assert not v.any()


def _layer_case(dev, kind, N, cin, cout, D, H, W, relu, with_res, seed, lo4=0):
    """One launch of the kind's kernel against fp64 next to the fp32 chain (torch CPU); D, H, W = the input dims."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, D, H, W, generator=g)
    if kind == "up":
        w = torch.randn(cin, cout, 3, 3, 3, generator=g) * (2.0 / (27 * cin / 8)) ** 0.5
        od = (2 * D, 2 * H, 2 * W)
    else:
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
        od = (D, H, W) if kind == "s1" else (D // 2, H // 2, W // 2)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(N, cout, *od, generator=g) if with_res else None

    def chain(dt):
        if kind == "up":
            y = F.conv_transpose3d(x.to(dt), w.to(dt), stride=2, padding=1, output_padding=1)
        else:
            y = F.conv3d(x.to(dt), w.to(dt), padding=1, stride=1 if kind == "s1" else 2)
        y = y * scale.to(dt).view(1, -1, 1, 1, 1) + shift.to(dt).view(1, -1, 1, 1, 1)
        if with_res:
            y = y + res.to(dt)
        return y.clamp_min(0) if relu else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wd = w.to(dev)
    wp, wexp = s16.pack_weight_s16(wd.transpose(0, 1).contiguous() if kind == "up" else wd)
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    y16 = E.RS16(N, cout, *od, 1, dev)
    plan = E.ConvPlanS16(N, cin, cout, D, H, W, relu, device=dev, kind=kind)
    r16 = E.RS16(N, cout, *od, 1, dev).from_dense(res.to(dev)) if with_res else None
    plan.run(E.RS16(N, cin, D, H, W, 1, dev).from_dense(x.to(dev)), wp, sc, shift.to(dev), y16=y16, res=r16, lo4=lo4)
    got = y16.to_dense().cpu()
    m = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    print(f"{plan.kname}: max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
    assert err <= 2e-5 * m + 1e-5
    assert err <= 2.0 * e32 + 1e-6 * m, (err, e32)
    v = y16.view7().clone()
    v[:, :, 1:od[0] + 1, 1:od[1] + 1, :, 1:od[2] + 1] = 0
    assert not v.any()                   # the halo stays zero


@pytest.mark.parametrize("kind,N,cin,cout,D,H,W,relu,with_res", [
    ("s1", 3, 64, 64, 6, 14, 14, True, True),        # hourglass conv2 (2 x 14 tiles)
    ("s1", 5, 64, 64, 3, 7, 7, True, False),         # hourglass conv4 (4 x 7 tiles, ragged last row tile)
    ("s1", 2, 32, 32, 3, 6, 14, False, True),
    ("s1", 2, 32, 64, 6, 5, 7, True, False),
    ("s2", 3, 32, 64, 12, 28, 28, True, False),      # hourglass conv1, Config A
    ("s2", 3, 64, 64, 6, 14, 14, True, False),       # hourglass conv3, Config A (ragged 4 x 7 row tile)
    ("s2", 2, 64, 64, 12, 28, 28, True, False),      # hourglass conv3, Config B
    ("s2", 1, 32, 64, 24, 56, 56, True, False),      # hourglass conv1, Config B
    ("s2", 2, 64, 32, 2, 14, 14, False, False),
    ("s2", 2, 32, 32, 6, 14, 14, True, False),
    ("s2", 1, 64, 64, 4, 8, 56, True, False),
    ("s2", 2, 32, 64, 6, 14, 14, True, False),       # cout-split form on 4 x 7 tiles (ragged row tile)
    ("s2", 2, 32, 64, 4, 6, 56, False, False),       # cout-split form on 1 x 28 tiles, odd row count
    ("up", 3, 64, 64, 3, 7, 7, True, True),          # hourglass conv5, Config A
    ("up", 3, 64, 32, 6, 14, 14, False, True),       # hourglass conv6, Config A
    ("up", 1, 64, 32, 12, 28, 28, False, True),      # hourglass conv6, Config B
    ("up", 2, 64, 64, 6, 14, 14, True, False),       # hourglass conv5, Config B
    ("up", 2, 64, 64, 1, 3, 7, False, False),
    ("up", 2, 64, 32, 2, 5, 56, True, True),
    # round 6: masked last tiles / any depth
    ("s1", 3, 64, 64, 4, 8, 8, True, True),          # 2 x 14 tiles, 8 of 14 columns
    ("s1", 3, 64, 64, 2, 4, 4, True, False),         # 4 x 7 tiles, 4 of 7 columns, two phantom planes... (D = 2 -> walk of 3)
    ("s1", 2, 32, 32, 5, 9, 12, False, True),
    ("s1", 2, 64, 64, 10, 16, 16, True, True),       # 1 x 28 tiles, 16 of 28 columns
    ("s2", 3, 32, 64, 8, 16, 32, True, False),       # -> 4 x 8 x 16 (cout split, 1 x 28 tile masked)
    ("s2", 3, 64, 64, 4, 8, 16, True, False),        # -> 2 x 4 x 8 (2 x 14 tile masked)
    ("s2", 2, 64, 64, 2, 6, 8, True, False),         # -> 1 x 3 x 4 (4 x 7 tile masked)
    ("s2", 2, 32, 64, 16, 20, 40, True, False),      # -> 8 x 10 x 20
    ("s2", 2, 32, 32, 4, 10, 72, False, False),      # -> 36 wide: 28 + 8
    ("up", 3, 64, 64, 2, 4, 4, True, True),          # -> 4 x 8 x 8
    ("up", 3, 64, 32, 4, 8, 8, False, True),         # -> 8 x 16 x 16
    ("up", 2, 64, 32, 8, 10, 20, False, True),       # -> 16 x 20 x 40
    ("up", 2, 64, 64, 3, 5, 36, True, False),
])
def test_hourglass_layers_s16_vs_fp64_next_to_the_fp32_chain(dev, kind, N, cin, cout, D, H, W, relu, with_res):
    _layer_case(dev, kind, N, cin, cout, D, H, W, relu, with_res, seed=("s1", "s2", "up").index(kind) * 331 + N + cin + 3 * cout + 7 * D + 11 * H + 13 * W)


@pytest.mark.parametrize("N,D,H,W,lo4", [(80, 6, 28, 28, 0), (3, 12, 28, 28, -5), (2, 8, 12, 40, 2), (5, 4, 6, 64, -1), (2, 1, 2, 16, 0)])
def test_costvol_layer_two_tiles_per_wave_is_bit_identical_and_vs_fp64(dev, N, D, H, W, lo4):
    """convs16w.hip (round 6): the cost-volume layer with two MFMA tiles per wave -- through drc_conv3d_k3_s16_wide_fwd directly (any batch;
    masked last x tile at W = 40 / 16, phantom depth planes at D = 8 / 4 / 1) and through the library's own dispatch (80 units of 28 rows:
    1120 two-row columns -> picked) -- against the one-tile kernel BIT FOR BIT (same products, same summation order) and against the fp64
    convolution of the materialised volume next to the fp32 chain."""
    import ctypes as C
    from disprcnn_amd import _lib
    from disprcnn_amd._lib import DrcS16ConvParams
    g = torch.Generator().manual_seed(N * 100 + D + W + lo4)
    w = torch.randn(32, 64, 3, 3, 3, generator=g) * (2.0 / (27 * 64)) ** 0.5
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.1
    L, R = torch.randn(N, 32, H, W, generator=g), torch.randn(N, 32, H, W, generator=g)
    x = _ref_costvol(L, R, lo4, D)

    def chain(dt):
        return (F.conv3d(x.to(dt), w.to(dt), padding=1) * scale.to(dt).view(1, -1, 1, 1, 1) + shift.to(dt).view(1, -1, 1, 1, 1)).clamp_min(0)
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    sh = shift.to(dev)
    l16, r16 = E.RS16(N, 32, 1, H, W, 0, dev).from_dense(L.to(dev)), E.RS16(N, 32, 1, H, W, 0, dev).from_dense(R.to(dev))
    lib = _lib.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    outs = {}
    for name, fn, dil in (("wide", lib.drc_conv3d_k3_s16_wide_fwd, 1), ("dispatch", lib.drc_conv3d_k3_s16_fwd, 1), ("one tile", lib.drc_conv3d_k3_s16_fwd, 0x800)):
        y = E.RS16(N, 32, D, H, W, 1, dev)
        prm = DrcS16ConvParams(None, P(wp), P(sc), P(sh), None, P(y.storage), None, P(l16.storage), P(r16.storage), N, D, H, W, 64, 32, 1, lo4, dil)
        _lib.check(fn(C.byref(prm), st), name)
        if name == "dispatch":
            assert bool(lib.drc_conv3d_k3_s16_wide(C.byref(prm))) == (N * (H // 2) * -(-W // 28) >= 1024 and H % 2 == 0 and W > 14)
        outs[name] = y
    torch.cuda.synchronize()
    assert torch.equal(outs["wide"].storage, outs["one tile"].storage) and torch.equal(outs["dispatch"].storage, outs["one tile"].storage)
    got = outs["wide"].to_dense().cpu()
    m = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    print(f"convs16w N={N} {D}x{H}x{W} lo4={lo4}: max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
    assert err <= 2e-5 * m + 1e-5 and err <= 2.0 * e32 + 1e-6 * m
    v = outs["wide"].view7().clone()
    v[:, :, 1:D + 1, 1:H + 1, :, 1:W + 1] = 0
    assert not v.any()                   # the halo stays zero


@pytest.mark.parametrize("N,D,H,W,with_prev", [
    (3, 12, 28, 28, True),          # Config A volume
    (18, 6, 28, 28, False),         # some workgroups walk two columns (units 0, 8, 16 share an XCD's 32 workers): the depth sums cross column boundaries
    (2, 6, 6, 56, True),            # two x tiles per row: the gather crosses tile boundaries; shortest supported depth
    (9, 24, 4, 84, True),           # more units than XCDs, Config B depth
    (1, 24, 56, 56, False),         # Config B volume
])
def test_fused_cout1_head_vs_fp64(dev, N, D, H, W, with_prev):
    """classif[0] (3x3x3 32 -> 32 + BN + ReLU) and classif[2] (3x3x3 32 -> 1), stackhourglass.py:78-88, + the cumulative head add of
    :142-144 as ONE split-f16 launch that stores per-source-voxel partial sums + drc_head_gather_fwd, against the two convolutions in fp64,
    next to the fp32 chain of the same two layers (torch CPU)."""
    g = torch.Generator().manual_seed(N * 1000 + D + H + W)
    x = torch.randn(N, 32, D, H, W, generator=g)
    w0 = torch.randn(32, 32, 3, 3, 3, generator=g) * (2.0 / (27 * 32)) ** 0.5
    w1 = torch.randn(1, 32, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.1
    prev = torch.randn(N, D, H, W, generator=g) if with_prev else None

    def chain(dt):
        a = F.conv3d(x.to(dt), w0.to(dt), padding=1) * scale.to(dt).view(1, -1, 1, 1, 1) + shift.to(dt).view(1, -1, 1, 1, 1)
        y = F.conv3d(a.clamp_min(0), w1.to(dt), padding=1)[:, 0]
        return y + prev.to(dt) if with_prev else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wp, wexp = s16.pack_weight_s16(w0.to(dev))
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    hp, hexp = s16.pack_head_weight_s16(w1)
    S = torch.full((N, D, H, W, 12), float("nan"), device=dev)          # every slot the gather reads must have been written
    plan = E.ConvPlanS16(N, 32, 32, D, H, W, True, device=dev, kind="s1")
    plan.run(E.RS16(N, 32, D, H, W, 1, dev).from_dense(x.to(dev)), wp, sc, shift.to(dev), head=(hp.to(dev), S))
    out = torch.empty(N, D, H, W, device=dev)
    E.head_gather(S, 2.0 ** -hexp, prev.to(dev) if with_prev else None, out)
    got = out.cpu()
    m = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    print(f"fused head N={N} {D}x{H}x{W}: max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
    assert torch.isfinite(got).all()
    assert err <= 2e-5 * m + 1e-5
    assert err <= 2.0 * e32 + 1e-6 * m, (err, e32)


@pytest.mark.parametrize("N,D,H,W", [(2, 3, 5, 7), (1, 2, 30, 57), (3, 2, 28, 28), (1, 1, 3, 400), (2, 1, 56, 56)])
def test_head_gather_vs_torch(dev, N, D, H, W):
    """drc_head_gather_fwd alone: cost = res + scale * sum_{kh,kw} S[.., y+kh-1, x+kw-1][slot(kh*3+kw)], zero outside the map (slot j for j < 5,
    j + 3 above); LDS-staged bands, whole planes, and the one-thread-per-voxel form very wide maps fall back to (W = 400)."""
    g = torch.Generator().manual_seed(N + D + H + W)
    S = torch.randn(N, D, H, W, 12, generator=g)
    res = torch.randn(N, D, H, W, generator=g)
    ref = torch.zeros(N, D, H, W, dtype=torch.float64)
    Sp = F.pad(S.double(), (0, 0, 1, 1, 1, 1))                      # zero border in x and y
    for kh in range(3):
        for kw in range(3):
            j = kh * 3 + kw
            ref += Sp[:, :, kh:kh + H, kw:kw + W, j if j < 5 else j + 3]
    ref = ref * 0.375 + res.double()
    S_dev = S.to(dev)
    S_dev[..., 5:8] = float("nan")                                  # the slots' padding floats are never read
    out = torch.empty(N, D, H, W, device=dev)
    E.head_gather(S_dev, 0.375, res.to(dev), out)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    out2 = torch.empty(N, D, H, W, device=dev)
    E.head_gather(S_dev, 0.375, None, out2)
    assert (out2.cpu().double() - (ref - res.double())).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_fused_head_validation(dev):
    x = E.RS16(2, 32, 12, 28, 28, 1, dev)
    w0 = torch.randn(32, 32, 3, 3, 3, device=dev)
    wp, _ = s16.pack_weight_s16(w0)
    one, zero = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    hp, _ = s16.pack_head_weight_s16(torch.randn(1, 32, 3, 3, 3))
    plan = E.ConvPlanS16(2, 32, 32, 12, 28, 28, True, device=dev, kind="s1")
    with pytest.raises(ValueError):
        plan.run(x, wp, one, zero, head=(hp.to(dev), torch.empty(2 * 12 * 28 * 28 * 12 - 1, device=dev)))          # S too small
    with pytest.raises(ValueError):
        plan.run(x, wp, one, zero, y16=E.RS16(2, 32, 12, 28, 28, 1, dev), head=(hp.to(dev), torch.empty(2, 12, 28, 28, 12, device=dev)))
    with pytest.raises(ValueError):
        E.ConvPlanS16(2, 32, 32, 3, 28, 28, True, device=dev, kind="s1").run(E.RS16(2, 32, 3, 28, 28, 1, dev), wp, one, zero,
                                                                            head=(hp.to(dev), torch.empty(2, 3, 28, 28, 12, device=dev)))   # D < 6


# the forms measured against the product kernels stay selectable (lo4 bits of a non-cost-volume launch: 0x100 the bank-conflict-free tile
# lanes of s16_tilemap.h -- slower, see there --, 0x200 interleaved slab rows and 0x400 no cout split in the stride-2 kernel) and must stay
# correct: the A/B timings of tools/experiments/exp_s16_forms.py compare like with like
@pytest.mark.parametrize("kind,N,cin,cout,D,H,W,relu,with_res,lo4", [
    ("s1", 3, 64, 64, 6, 14, 14, True, True, 0x100), ("s1", 5, 64, 64, 3, 7, 7, True, False, 0x100), ("s1", 2, 32, 32, 3, 6, 14, False, True, 0x100),
    ("s2", 3, 32, 64, 12, 28, 28, True, False, 0x100), ("s2", 3, 32, 64, 12, 28, 28, True, False, 0x200), ("s2", 3, 32, 64, 12, 28, 28, True, False, 0x400),
    ("s2", 3, 32, 64, 12, 28, 28, True, False, 0x700), ("s2", 3, 64, 64, 6, 14, 14, True, False, 0x100), ("s2", 3, 64, 64, 6, 14, 14, True, False, 0x200),
    ("s2", 1, 32, 64, 24, 56, 56, True, False, 0x600), ("s2", 2, 32, 32, 6, 14, 14, True, False, 0x300),
    ("up", 3, 64, 64, 3, 7, 7, True, True, 0x100), ("up", 3, 64, 32, 6, 14, 14, False, True, 0x100),
])
def test_hourglass_layers_s16_experiment_forms(dev, kind, N, cin, cout, D, H, W, relu, with_res, lo4):
    _layer_case(dev, kind, N, cin, cout, D, H, W, relu, with_res, seed=("s1", "s2", "up").index(kind) * 331 + N + cin + 3 * cout + 7 * D + 11 * H + 13 * W, lo4=lo4)


@pytest.mark.parametrize("N,cin,cout,H,W,relu,with_res,form", [
    (2, 128, 128, 56, 56, True, False, "d2"), (9, 128, 128, 56, 28, False, True, "d2"), (3, 128, 128, 112, 56, True, True, "d2"),   # layer4: dilation 2
    (2, 32, 32, 28, 56, True, False, 0), (3, 32, 32, 56, 112, False, True, 0),          # firstconv / layer1 (two tiles per workgroup, K over two waves)
    (9, 64, 64, 28, 28, True, True, 0), (2, 64, 64, 56, 56, True, False, 0),            # layer2: one tile, K over four waves
    (17, 64, 64, 28, 56, False, True, 2), (2, 64, 64, 56, 56, True, True, 2),           # ... the two-tile form with two K slices per wave
    (2, 64, 128, 28, 28, True, False, 1), (40, 64, 128, 28, 56, True, False, 0),        # layer3's first conv (four cout tiles side by side)
    (9, 128, 128, 28, 28, False, True, 0), (2, 128, 128, 56, 56, True, False, 0),       # layer3: two K slices per wave
    (33, 64, 64, 28, 28, True, True, 1), (40, 32, 32, 28, 56, True, True, 0),           # more images than one pass of the persistent grid's XCD groups
    # any map size (the trunk's / FPN's maps): ragged last x group and row block, maps smaller than one tile, cout up to 512
    (2, 64, 64, 30, 40, True, True, 0), (3, 32, 32, 10, 60, False, True, 0), (2, 64, 64, 5, 7, True, False, 0), (1, 128, 128, 94, 100, True, True, 0),
    (2, 128, 256, 47, 155, True, False, 0), (2, 64, 64, 30, 61, False, True, 2), (1, 128, 512, 24, 78, True, True, 0), (9, 32, 32, 29, 57, True, False, 0),
])
def test_conv2d_s16_vs_fp64_next_to_the_fp32_chain(dev, N, cin, cout, H, W, relu, with_res, form):
    """convs16r.hip (the 2D member: 3x3 stride-1 conv + BN (+ residual, + ReLU), reference submodule.py:9-16) against fp64 next to the fp32
    chain, at the bounds of the 3D cases."""
    dil = 2 if form == "d2" else 1
    form = 0 if form == "d2" else form
    g = torch.Generator().manual_seed(N * 100 + cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(N, cout, H, W, generator=g) if with_res else None

    def chain(dt):
        y = F.conv2d(x.to(dt), w.to(dt), padding=dil, dilation=dil) * scale.to(dt).view(1, -1, 1, 1) + shift.to(dt).view(1, -1, 1, 1)
        if with_res:
            y = y + res.to(dt)
        return y.clamp_min(0) if relu else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    assert tuple(wp.shape) == (cout // 32, cin // 16, 9, 2, 64, 8)
    sc = (scale * (2.0 ** -wexp)).to(dev).contiguous()
    x16 = E.RS16(N, cin, 1, H, W, 0, dev).from_dense(x.to(dev))
    y16 = E.RS16(N, cout, 1, H, W, 0, dev)
    r16 = E.RS16(N, cout, 1, H, W, 0, dev).from_dense(res.to(dev)) if with_res else None
    if form == 0:
        E.ConvPlanS16(N, cin, cout, 1, H, W, relu, device=dev, kind="2d", dil=dil).run(x16, wp, sc, shift.to(dev), y16=y16, res=r16)
    else:
        s16.conv2d_k3(x16.storage, wp, sc, shift.to(dev), N, H, W, cin, cout, relu, y16.storage, res=None if r16 is None else r16.storage, form=form)
    got = y16.to_dense().cpu()[:, :, 0]
    m = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    print(f"max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
    assert err <= 2e-5 * m + 1e-5
    assert err <= 2.0 * e32 + 1e-6 * m, (err, e32)
    v = y16.view7().clone()
    v[:, :, :, 1:H + 1, :, 1:W + 1] = 0
    assert not v.any()                                   # the halo stays zero


@pytest.mark.parametrize("N,cin,cout,H,W,relu,ph", [(2, 256, 64, 30, 61, True, 1), (1, 512, 128, 24, 78, False, 1), (2, 64, 64, 47, 155, True, 1),
                                                     (2, 128, 256, 33, 40, True, 0)])
def test_bridged_conv2d_s16_between_blocked_tensors(dev, N, cin, cout, H, W, relu, ph):
    """engine.BridgedConv2dS16: a 3x3 layer of the trunk / FPN / RPN head between BLOCKED fp32 tensors through the split-f16 kernel -- input
    converted per 128-channel slice, chained launches (partial sum as the next residual), output converted back -- against fp64."""
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1

    def chain(dt):
        y = F.conv2d(x.to(dt), w.to(dt), padding=1) * scale.to(dt).view(1, -1, 1, 1) + shift.to(dt).view(1, -1, 1, 1)
        return y.clamp_min(0) if relu else y
    ref = chain(torch.float64)
    e32 = (chain(torch.float32).double() - ref).abs().max().item()
    xb = E.Blocked(N, cin, 1, H, W, 0, ph, ph, dev).from_dense(x.to(dev))
    yb = E.Blocked(N, cout, 1, H, W, 0, 1, 1, dev)
    br = E.BridgedConv2dS16(N, cin, cout, H, W, relu, dev, {})
    packs = []
    for a, b in br.bounds:
        wp, wexp = s16.pack_weight_s16(w[:, a:b].contiguous().to(dev))
        packs.append((wp, (scale * 2.0 ** -wexp).to(dev).contiguous()))
    br.run(xb, packs, shift.to(dev), torch.zeros(cout, device=dev), yb)
    got = yb.to_dense().cpu()[:, :, 0]
    m = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    print(f"bridged {cin}->{cout} {H}x{W}: max|err| {err:.3e} (fp32 chain {e32:.3e}), max|ref| {m:.3f}")
    assert err <= 2e-5 * m + 1e-5
    assert err <= 2.5 * e32 + 1e-6 * m, (err, e32)
    v = yb.view6().clone()
    v[:, :, :, 1:H + 1, 1:W + 1] = 0
    assert not v.any()                                   # the blocked output's halo stays zero


def test_conv2d_s16_validation(dev):
    lib = _lib_handle()
    assert lib.drc_conv2d_k3_s16_supported(64, 64, 56, 56, 1) == 1 and lib.drc_conv2d_k3_s16_supported(32, 32, 112, 112, 1) == 1
    assert lib.drc_conv2d_k3_s16_supported(32, 32, 28, 28, 1) == 1 and lib.drc_conv2d_k3_s16_supported(64, 64, 30, 57, 1) == 1        # any map size (ragged tiles)
    assert lib.drc_conv2d_k3_s16_supported(128, 512, 12, 39, 1) == 1 and lib.drc_conv2d_k3_s16_supported(128, 96, 12, 39, 1) == 0
    assert lib.drc_conv2d_k3_s16_supported(48, 64, 28, 28, 1) == 0 and lib.drc_conv2d_k3_s16_supported(256, 64, 28, 28, 1) == 0       # cin: 32, 64, 128 (wider: chained launches)
    assert lib.drc_conv2d_k3_s16_supported(128, 128, 56, 56, 2) == 1 and lib.drc_conv2d_k3_s16_supported(128, 128, 28, 56, 2) == 0      # dilation 2: 56-row blocks
    assert lib.drc_conv2d_k3_s16_supported(64, 64, 56, 56, 2) == 0 and lib.drc_conv2d_k3_s16_supported(128, 128, 56, 56, 3) == 0
    with pytest.raises(ValueError):
        E.ConvPlanS16(2, 48, 64, 1, 28, 56, True, device=dev, kind="2d")
    plan = E.ConvPlanS16(2, 64, 64, 1, 28, 28, True, device=dev, kind="2d")
    w = torch.zeros(2, 4, 9, 2, 64, 8, dtype=torch.float16, device=dev)
    sc = torch.ones(64, device=dev)
    with pytest.raises(ValueError):                                      # a volume (depth halo) is not a 2D map
        plan.run(E.RS16(2, 64, 1, 28, 28, 1, dev), w, sc, sc, y16=E.RS16(2, 64, 1, 28, 28, 0, dev))
    with pytest.raises(ValueError):
        plan.run(E.RS16(2, 64, 1, 28, 28, 0, dev), w, sc, sc, y16=E.RS16(2, 32, 1, 28, 28, 0, dev))


def _lib_handle():
    from disprcnn_amd import _lib
    return _lib.lib()


def test_feature_cnn_f16x2_vs_f32_path(dev):
    """Config B (full PSMNet on 224 x 224 crops): the default path (2D CNN's stride-1 3x3 layers and the regressor in split-f16) against
    the all-fp32-MFMA 2D CNN: the 32-channel feature maps agree to fp32 rounding (measured 2.8e-5 on a range of 10; bound 1e-4 of the
    range); the disparities behind the regressor at the bounds every HIP path meets against the reference (mean 1e-3 px, max 2e-2 px;
    measured 2.3e-4 / 5.1e-3: the regressor amplifies rounding-level feature differences, whichever fp32-class arithmetic made them)."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("B")
    left, right = synth.synth_images(3, 224, 224, tag="s16_2d")
    outs, feats = {}, {}
    for math in ("auto", "f32"):
        m = PSMNet(48, -48)
        m.load_state_dict(sd, strict=True)
        m.feature_math = math
        m = m.to(dev).eval()
        with torch.no_grad():
            outs[math] = m((left.to(dev), right.to(dev))).cpu()
        keys = [k[0] for k in m._rt._ws]
        assert ("2ds16" in keys) == (math == "auto") and ("2d" in keys) == (math == "f32"), keys
        feats[math] = m._rt._ws[("2ds16" if math == "auto" else "2d", 6, 224, 224)]["t"]["feat"].to_dense().cpu()
    fd = (feats["auto"] - feats["f32"]).abs().max().item()
    fm = feats["f32"].abs().max().item()
    d = (outs["auto"] - outs["f32"]).abs()
    print(f"features: max diff {fd:.3e} (max {fm:.3f}); disparity: mean {d.mean().item():.3e} max {d.max().item():.3e} px")
    assert fd <= 1e-4 * max(1.0, fm)
    assert d.mean().item() < 1e-3 and d.max().item() < 2e-2


def test_lastconv_as_chained_split_f16_launches(dev):
    """lastconv[0] (3x3, 320 -> 128 on the concat; reference submodule.py:125-128) runs as three chained split-f16 launches over the concat's
    parts (raw 64, skip 128, SPP branches 128), each adding the previous partial sum as its residual.  Against the same schedule with that
    layer on the fp32 Winograd kernel (engine.LASTCONV_S16 off) the features agree to fp32 rounding."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("B")
    left, right = synth.synth_images(2, 224, 224, tag="s16_last")
    feats = {}
    for on in (True, False):
        E.LASTCONV_S16["enabled"] = on
        try:
            m = PSMNet(48, -48)
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).eval()
            with torch.no_grad():
                m((left.to(dev), right.to(dev)))
            ws = m._rt._ws[("2ds16", 4, 224, 224)]
            assert ("last16" in ws) == on
            feats[on] = ws["t"]["feat"].to_dense().cpu()
        finally:
            E.LASTCONV_S16["enabled"] = True
    fd = (feats[True] - feats[False]).abs().max().item()
    fm = feats[False].abs().max().item()
    print(f"features: chained split-f16 lastconv[0] vs fp32 Winograd: max diff {fd:.3e} (max {fm:.3f})")
    assert fd <= 5e-5 * max(1.0, fm)


def test_feature_math_validation(dev):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, -48).to(dev).eval()
    left, right = synth.synth_images(1, 224, 224, tag="v2d")
    m.feature_math = "bf16"
    with pytest.raises(ValueError):
        m((left.to(dev), right.to(dev)))
    m.feature_math = "f16x2"
    left, right = synth.synth_images(1, 220, 224, tag="v2e")             # below the reference's own minimum (fixed AvgPool2d(56), submodule.py:76)
    with pytest.raises((RuntimeError, ValueError)):
        m((left.to(dev), right.to(dev)))


@pytest.mark.parametrize("H,W", [(256, 256), (224, 320)])
def test_full_psmnet_other_crop_sizes_f16x2_vs_f32_path(dev, H, W):
    """Round 6: the split-f16 2D CNN and regressor on crops beyond 224 x 224 (64 x 64 and 56 x 80 feature maps: ragged last tiles in the 2D
    kernel, masked last tiles / phantom planes in the 3D ones) against the all-fp32 HIP path: features to fp32 rounding, disparities at the
    bounds of the 224 x 224 test."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("B")
    left, right = synth.synth_images(2, H, W, tag=f"crop{H}x{W}")
    outs, feats = {}, {}
    for math in ("f16x2", "f32"):
        m = PSMNet(48, -48)
        m.load_state_dict(sd, strict=True)
        m.feature_math = m.regressor_math = math
        m = m.to(dev).eval()
        with torch.no_grad():
            outs[math] = m((left.to(dev), right.to(dev))).cpu()
        keys = [k[0] for k in m._rt._ws]
        assert ("2ds16" in keys and "3ds16" in keys) == (math == "f16x2"), keys
        feats[math] = m._rt._ws[("2ds16" if math == "f16x2" else "2d", 4, H, W)]["t"]["feat"].to_dense().cpu()
    fd = (feats["f16x2"] - feats["f32"]).abs().max().item()
    fm = feats["f32"].abs().max().item()
    d = (outs["f16x2"] - outs["f32"]).abs()
    print(f"{H}x{W}: features max diff {fd:.3e} (max {fm:.3f}); disparity: mean {d.mean().item():.3e} max {d.max().item():.3e} px")
    assert outs["f16x2"].shape == (2, H, W)
    assert fd <= 1e-4 * max(1.0, fm)
    assert d.mean().item() < 1e-3 and d.max().item() < 2e-2


def test_conv3d_s16_small_activations_keep_an_absolute_error_floor(dev):
    """Activations of 1e-3: the lo parts are subnormal fp16 numbers (absolute resolution 2^-25).  The f16 MFMA does not flush them: the
    error stays ~1e-7 absolute (it would be ~5e-5 relative with flushing)."""
    g = torch.Generator().manual_seed(5)
    N, D, H, W = 2, 12, 28, 28
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * (2.0 / (27 * 32)) ** 0.5
    x = torch.randn(N, 32, D, H, W, generator=g) * 1e-3
    ref = F.conv3d(x.double(), w.double(), padding=1)
    wp, wexp = s16.pack_weight_s16(w.to(dev))
    sc = torch.full((32,), 2.0 ** -wexp, device=dev)
    y16 = E.RS16(N, 32, D, H, W, 1, dev)
    E.ConvPlanS16(N, 32, 32, D, H, W, False, device=dev).run(E.RS16(N, 32, D, H, W, 1, dev).from_dense(x.to(dev)), wp, sc, torch.zeros(32, device=dev), y16=y16)
    err = (y16.to_dense().cpu().double() - ref).abs().max().item()
    print(f"max|err| {err:.3e} at max|ref| {ref.abs().max().item():.3e}")
    assert err < 5e-7


def test_deconv_direct_writes_rs16_identical_to_its_fp32_output(dev):
    """hourglass conv6 (stackhourglass.py:26-30,49): the RS16 epilogue of deconvdirect.hip = split(fp32 result), bit for bit; the fp32
    output of the same launch is unchanged; RS16-only launches (no fp32 output) give the same halfs."""
    g = torch.Generator().manual_seed(11)
    N, D, H, W = 20, 6, 14, 14
    x = torch.randn(N, 64, D, H, W, generator=g)
    w = torch.randn(64, 32, 3, 3, 3, generator=g) * 0.05
    res = torch.randn(N, 32, 2 * D, 2 * H, 2 * W, generator=g)
    scale = (torch.rand(32, generator=g) + 0.5).to(dev)
    shift = (torch.randn(32, generator=g) * 0.1).to(dev)
    xb = E.Blocked(N, 64, D, H, W, 1, 1, 1, dev).from_dense(x.to(dev))
    rb = E.Blocked(N, 32, 2 * D, 2 * H, 2 * W, 1, 1, 1, dev).from_dense(res.to(dev))
    y_plain = E.Blocked(N, 32, 2 * D, 2 * H, 2 * W, 1, 1, 1, dev)
    y_both = E.Blocked(N, 32, 2 * D, 2 * H, 2 * W, 1, 1, 1, dev)
    pl = E.plan_deconv3d(xb, y_plain, 32, False)
    assert pl.deconv_direct
    wt = E.pack_weight(w.to(dev), True)
    w16 = pl.pack16(w.to(dev), True)
    pl.run(xb, wt, scale, shift, y_plain, rb, w16=w16)
    s_both, s_only = E.RS16(N, 32, 2 * D, 2 * H, 2 * W, 1, dev), E.RS16(N, 32, 2 * D, 2 * H, 2 * W, 1, dev)
    pl.run(xb, wt, scale, shift, y_both, rb, w16=w16, y16=s_both)
    pl.run(xb, wt, scale, shift, None, rb, w16=w16, y16=s_only)
    assert torch.equal(y_plain.storage, y_both.storage)
    want = s16.rs16_from_dense(y_plain.to_dense().cpu())
    assert torch.equal(s_both.view7().cpu(), want)
    assert torch.equal(s_only.view7().cpu(), want)


@pytest.mark.parametrize("mx,mn,N,Hp,Wp", [(64, 0, 3, 32, 32), (16, -16, 2, 20, 40), (32, 0, 2, 16, 64), (80, 0, 2, 28, 28)])
def test_regressor_f16x2_other_volume_shapes_vs_f32_path_and_oracle(dev, mx, mn, N, Hp, Wp):
    """VERDICT r5 missing #2: the reference takes any D, H, W = 0 mod 16 (stackhourglass.py:115-174); the split-f16 schedule now takes every
    such volume of at least 16 columns (masked last tiles, phantom depth planes; the heads fall back to the stand-alone cout-1 kernel where the
    fused form's shape rule does not hold).  128 x 128 / 64 disparities, 80 x 160 / 32, 64 x 256 / 32, 112 x 112 / 80: against the fp32 HIP path and
    the CPU oracle at the bounds of test_hip_parity.py."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("A")
    fl, fr = synth.synth_features(N, 32, Hp, Wp, tag=f"shape_{Hp}_{Wp}")
    outs = {}
    for math in ("f16x2", "f32"):
        m = PSMNet(mx, mn)
        m.load_state_dict(sd, strict=True)
        m.regressor_math = math
        m = m.to(dev).eval()
        with torch.no_grad():
            outs[math] = m.forward_from_features(fl.to(dev), fr.to(dev), (4 * Hp, 4 * Wp)).cpu()
        keys = [k[0] for k in m._rt._ws]
        assert ("3ds16" in keys) == (math == "f16x2"), keys
        if math == "f16x2":
            assert m._rt._guard.policy.overflows == 0
    with torch.no_grad():
        ref = O.psmnet_from_features(sd, fl, fr, mx, mn, 4 * Hp, 4 * Wp)
    for math, got in outs.items():
        err = (got - ref).abs()
        print(f"{math} D'={(mx - mn) // 4} {Hp}x{Wp}: mean/max err px vs oracle {err.mean().item():.3e} {err.max().item():.3e}")
        assert err.mean().item() < 1e-3 and err.max().item() < 2e-2
    d = (outs["f16x2"] - outs["f32"]).abs()
    print(f"f16x2 vs f32 HIP paths: mean {d.mean().item():.3e} max {d.max().item():.3e} px")
    assert d.max().item() < 3e-3 and d.mean().item() < 1e-4          # two fp32-class paths (measured 1.2e-3 max at 64 / 80 disparities: the soft-argmin's range scales it)


@pytest.mark.parametrize("mx,mn,N", [(48, 0, 16), (24, -24, 5), (48, 0, 1)])
def test_regressor_f16x2_vs_f32_path_and_oracle(dev, mx, mn, N):
    """Config A from the feature boundary: the default (split-f16) path against the all-fp32-MFMA path of rounds 1-4 and against the CPU
    oracle, at the bounds of test_hip_parity.py (mean <= 1e-3 px, max <= 2e-2 px); the two HIP paths agree to 1e-3 px max."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = state_for("A")
    fl, fr = synth.synth_features(N, 32, 28, 28, tag=f"s16_{N}")
    outs = {}
    for math in ("auto", "f32"):
        m = PSMNet(mx, mn)
        m.load_state_dict(sd, strict=True)
        m.regressor_math = math
        m = m.to(dev).eval()
        with torch.no_grad():
            outs[math] = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        keys = [k[0] for k in m._rt._ws]
        assert ("3ds16" in keys) == (math == "auto"), keys
    with torch.no_grad():
        ref = O.psmnet_from_features(sd, fl, fr, mx, mn, 112, 112)
    for math, got in outs.items():
        err = (got - ref).abs()
        print(f"{math}: mean/max err px vs oracle {err.mean().item():.3e} {err.max().item():.3e}")
        assert err.mean().item() < 1e-3 and err.max().item() < 2e-2
    d = (outs["auto"] - outs["f32"]).abs()
    print(f"f16x2 vs f32 HIP paths: mean {d.mean().item():.3e} max {d.max().item():.3e} px")
    assert d.max().item() < 1e-3 + 1e-2 * (err.max().item() > 5e-3)


def test_regressor_math_validation(dev):
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, 0).to(dev).eval()
    m.regressor_math = "bf16"
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="v")
    with pytest.raises(ValueError):
        m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112))
    m.regressor_math = "f16x2"
    fl, fr = synth.synth_features(2, 32, 12, 12, tag="v2")          # 12-wide maps: below the cost-volume layer's 28-wide tiles
    with pytest.raises(RuntimeError):
        m.forward_from_features(fl.to(dev), fr.to(dev), (48, 48))
    m.regressor_math = "auto"                                      # ... "auto" falls back to the fp32 kernels
    with torch.no_grad():
        out = m.forward_from_features(fl.to(dev), fr.to(dev), (48, 48))
    assert out.shape == (2, 48, 48) and torch.isfinite(out).all() and "3ds16" not in [k[0] for k in m._rt._ws]
