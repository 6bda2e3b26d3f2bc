"""CPU-only checks: C-ABI library loads and exports every declared symbol, state-dict layout, host-side planning."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    from disprcnn_amd import _lib
    from disprcnn_amd.csrc import build
    build.build(verbose=False)
    header = open(os.path.join(ROOT, "include", "disprcnn_hip.h")).read()
    declared = set(re.findall(r"\b(drc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/disprcnn_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert b"gfx950" in _lib.lib().drc_version()


def test_round4_fp16_entries_reject_bad_arguments_without_launching():
    """The conv16x.hip entries validate before they launch (no GPU needed): null block / pointers -> -1, empty grid -> -2, a parameter block
    that is not the layer the entry implements -> -4 (`_supported` -> 0); an empty batch is a no-op (0)."""
    from disprcnn_amd import _lib, engine as E
    L = _lib.lib()
    p = _lib.DrcTapconvParams()
    for fn in (L.drc_conv16_k3s2_tile_fwd, L.drc_deconv16_k3s2_tile_fwd):
        assert fn(None, None) == -1 and fn(ctypes.byref(p), None) == -1
    assert L.drc_conv16_k3_costvol_fwd(None, 0, None) == -1 and L.drc_conv16_k3_costvol_fwd(ctypes.byref(p), 0, None) == -1
    assert L.drc_conv16_k3s2_tile_supported(None) == 0 and L.drc_deconv16_k3s2_tile_supported(None) == 0
    buf = (ctypes.c_float * 64)()
    addr = ctypes.addressof(buf)
    for f in ("x", "w", "y", "scale", "shift"):
        setattr(p, f, addr)
    p.N, p.OD, p.OH, p.OW = 1, 0, 4, 4
    assert L.drc_conv16_k3s2_tile_fwd(ctypes.byref(p), None) == -2 and L.drc_deconv16_k3s2_tile_fwd(ctypes.byref(p), None) == -2
    assert L.drc_conv16_k3_costvol_fwd(ctypes.byref(p), 0, None) == -2
    p.OD = 4
    p.N = 0
    assert L.drc_conv16_k3s2_tile_fwd(ctypes.byref(p), None) == 0 and L.drc_conv16_k3_costvol_fwd(ctypes.byref(p), 0, None) == 0
    p.N = 1                                                   # a zeroed class table is none of the three layers
    assert L.drc_conv16_k3s2_tile_supported(ctypes.byref(p)) == 0 and L.drc_deconv16_k3s2_tile_supported(ctypes.byref(p)) == 0
    assert L.drc_conv16_k3s2_tile_fwd(ctypes.byref(p), None) == -4 and L.drc_deconv16_k3s2_tile_fwd(ctypes.byref(p), None) == -4
    assert L.drc_conv16_k3_costvol_fwd(ctypes.byref(p), 0, None) == -4
    # the class tables the engine builds ARE recognised (stride-2 conv and transposed conv), and not by the other entry
    def fill(classes, in_mul, out_mul, cb_in, cout_pad):
        q = _lib.DrcTapconvParams()
        q.n_classes, q.in_mul, q.out_mul, q.cb_in, q.cout_pad = len(classes), in_mul, out_mul, cb_in, cout_pad
        for ci, c in enumerate(classes):
            k = q.cls[ci]
            k.nd, k.nh, k.nw = c["n"]; k.dd0, k.dh0, k.dw0 = c["first"]; k.sd, k.sh, k.sw = c["step"]
            k.wbase = c["wbase"]; k.wsd, k.wsh, k.wsw = c["wstep"]; k.out_off_d, k.out_off_h, k.out_off_w = c["off"]
        return q
    down = fill(E.taps_conv((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)), 2, 1, 1, 64)
    up = fill(E.taps_deconv3d_k3s2(), 1, 2, 2, 32)
    assert L.drc_conv16_k3s2_tile_supported(ctypes.byref(down)) == 1 and L.drc_deconv16_k3s2_tile_supported(ctypes.byref(down)) == 0
    assert L.drc_deconv16_k3s2_tile_supported(ctypes.byref(up)) == 1 and L.drc_conv16_k3s2_tile_supported(ctypes.byref(up)) == 0
    up.cb_in = 5                                              # more input blocks than the instantiated stages: conv16.hip keeps the layer
    assert L.drc_deconv16_k3s2_tile_supported(ctypes.byref(up)) == 0


def test_params_struct_matches_header_size():
    """sizeof(drc_tapconv_params) computed by gcc must equal the ctypes mirror."""
    import subprocess, tempfile
    from disprcnn_amd import _lib
    src = '#include <stdio.h>\n#include "disprcnn_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu", sizeof(drc_tapconv_params), sizeof(drc_tap_class), sizeof(int32_t), sizeof(drc_costvol_src), sizeof(drc_wgrad_params), sizeof(drc_s16conv_params));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b, t, cvs, wg, s16p = map(int, subprocess.check_output([exe]).split())
    assert a == ctypes.sizeof(_lib.DrcTapconvParams)
    assert b == ctypes.sizeof(_lib.DrcTapClass) and t == 4
    assert cvs == ctypes.sizeof(_lib.DrcCostvolSrc)
    assert wg == ctypes.sizeof(_lib.DrcWgradParams)
    assert s16p == ctypes.sizeof(_lib.DrcS16ConvParams)            # (round 6: + the range-guard word `ovf`)


def test_state_dict_layout():
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    sd = PSMNet(48, -48).state_dict()
    assert len(sd) == 514
    assert sum(v.numel() for v in sd.values()) == 5236053
    assert tuple(sd["dres0.0.0.weight"].shape) == (32, 64, 3, 3, 3)
    assert tuple(sd["dres2.conv5.0.weight"].shape) == (64, 64, 3, 3, 3)
    assert tuple(sd["dres2.conv6.0.weight"].shape) == (64, 32, 3, 3, 3)
    assert tuple(sd["classif1.2.weight"].shape) == (1, 32, 3, 3, 3)
    assert sum(1 for k in sd if k.startswith("feature_extraction")) == 361
    n_params = sum(p.numel() for p in PSMNet(48, -48).parameters())
    assert n_params == 5224768


def test_no_cpu_fallback():
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, 0).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward_from_features(torch.zeros(1, 32, 28, 28), torch.zeros(1, 32, 28, 28), (112, 112))


def test_pack_weight_layout():
    from disprcnn_amd import engine as E
    w = torch.arange(2 * 3 * 27, dtype=torch.float32).view(2, 3, 3, 3, 3)
    p = E.pack_weight(w)
    assert tuple(p.shape) == (27, 1, 2, 16, 8)          # [tap][cb][half][cout][8 ch]
    for t in (0, 13, 26):
        for co in range(2):
            for ci in range(3):
                assert p[t, 0, ci // 8, co, ci % 8] == w[co, ci].reshape(-1)[t]
    assert p[:, :, :, 2:, :].abs().sum() == 0 and p[:, :, 0, :, 3:].abs().sum() == 0 and p[:, :, 1].abs().sum() == 0
    w2 = torch.arange(4 * 20 * 27, dtype=torch.float32).view(4, 20, 3, 3, 3)
    p2 = E.pack_weight(w2)
    assert tuple(p2.shape) == (27, 2, 2, 16, 8)
    assert p2[7, 1, 0, 3, 2] == w2[3, 18].reshape(-1)[7] and p2[7, 0, 1, 3, 5] == w2[3, 13].reshape(-1)[7]
    wt = torch.arange(3 * 2 * 27, dtype=torch.float32).view(3, 2, 3, 3, 3)     # ConvTranspose: [Cin,Cout,...]
    pt = E.pack_weight(wt, transposed=True)
    assert pt[5, 0, 0, 1, 2] == wt[2, 1].reshape(-1)[5]


def test_deconv_parity_classes_cover_all_taps():
    from disprcnn_amd import engine as E
    cls = E.taps_deconv3d_k3s2()
    assert len(cls) == 8
    widx = sorted(t[3] for c in cls for t in E.class_taps(c))
    assert widx == list(range(27))
    assert sorted(len(E.class_taps(c)) for c in cls) == [1, 2, 2, 2, 4, 4, 4, 8]
    # o = 2i - 1 + k: class parity p, tap (offset, k) must satisfy 2*(j+off) - 1 + k == 2j + p per dimension
    for c in cls:
        for (dd, dh, dw, w) in E.class_taps(c):
            ks = (w // 9, (w // 3) % 3, w % 3)
            for off, k, par in zip((dd - 1, dh - 1, dw - 1), ks, c["off"]):
                assert 2 * off - 1 + k == par


@pytest.mark.parametrize("shape", [(28, 28, 1, 2, 2), (56, 56, 1, 2, 2), (14, 14, 2, 2, 2), (7, 7, 1, 1, 1), (112, 112, 2, 2, 2),
                                   (112, 112, 1, 2, 2), (56, 56, 1, 4, 4), (375, 1242, 2, 6, 6)])
def test_tile_choice_fits_lds(shape):
    from disprcnn_amd import engine as E
    OH, OW, im, sh, sw = shape
    R, WT, lds = E.choose_tile(OH, OW, im, sh, sw)
    assert R * WT <= 112 and lds <= E.LDS_PER_WAVE_MAX
    assert lds == 2 * (im * (R - 1) + sh + 1) * (im * (WT - 1) + sw + 1) * 32


def test_stride1_conv3d_kernel_selection():
    """Which kernel the engine plans for a stride-1 3x3x3 layer: Winograd for even output dims with enough work, the direct
    kernel for odd dims or with the switch off, the generic kernel for tiny batches; and a plan refuses the wrong packing."""
    from disprcnn_amd import engine as E
    dev = torch.device("cpu")

    def plan(n, c, dims):
        x, y = E.Blocked(n, c, *dims, 1, 1, 1, dev), E.Blocked(n, c, *dims, 1, 1, 1, dev)
        return E.plan_conv3d(x, y, 1, c, True)

    pl = plan(256, 32, (12, 28, 28))
    assert pl.wino and pl.rb and pl.direct and pl.kname == "wino3d_rb_kernel<14>"       # round 3: the two-waves-per-SIMD row-brick kernel
    assert plan(12, 32, (12, 28, 28)).kname == "wino3d_kernel<2>"                      # fewer 64-tile chunks than CUs: wino3d.hip
    pl = plan(256, 64, (3, 7, 7))                       # conv4 of Config A: odd dims
    assert not pl.wino and pl.direct and pl.kname.startswith("tapdirect")
    saved = E.WINO["enabled"]
    E.WINO["enabled"] = False
    try:
        assert not plan(256, 32, (12, 28, 28)).wino
    finally:
        E.WINO["enabled"] = saved
    pl = plan(2, 32, (12, 28, 28))                      # below the sliding kernels' unit threshold, 36 Winograd chunks (< 48):
    assert not pl.wino and pl.direct and pl.kname == "tapdirect_kernel<7,1>"            # the direct kernel, one cout tile per wave (round 3)
    pl = plan(4, 64, (12, 28, 28))                      # ... 73 chunks: Winograd (rounds 1-2: the generic kernel)
    assert pl.wino and pl.kname == "wino3d_kernel<2>"
    pl = plan(1, 32, (2, 4, 4))                         # a handful of voxels: the generic kernel
    assert not pl.wino and not pl.direct
    # a Winograd plan wants the 64-point packing, not the 27-tap one
    pl = plan(256, 32, (12, 28, 28))
    x, y = E.Blocked(1, 32, 2, 2, 2, 1, 1, 1, dev), E.Blocked(1, 32, 2, 2, 2, 1, 1, 1, dev)
    w = torch.zeros(32, 32, 3, 3, 3)
    with pytest.raises(ValueError):
        pl.run(x, E.pack_weight(w), torch.ones(32), torch.zeros(32), y, w16=E.pack_weight_t16(w))


def test_conv2d_3x3_kernel_selection():
    """2D Winograd for stride-1 undilated 3x3 layers with enough tile groups; the direct kernel for strides, dilations and small
    problems (the backbone's deep stages)."""
    from disprcnn_amd import engine as E
    dev = torch.device("cpu")

    def plan(n, c, hw, stride=1, dil=1):
        x = E.Blocked(n, c, 1, *hw, 0, max(dil, 1), max(dil, 1), dev)
        oh, ow = (hw[0] - 1) // stride + 1, (hw[1] - 1) // stride + 1
        return E.plan_conv2d(x, E.Blocked(n, c, 1, oh, ow, 0, 1, 1, dev), 3, stride, dil, dil, c, True)

    assert plan(32, 32, (112, 112)).kname == "wino2d_rb_kernel<14>"      # <= 128 input channels on a 28-multiple width: the row-brick form (round 3)
    assert plan(2, 256, (56, 56)).kname == "wino2d_kernel<2>"            # more input channels (or too few chunks): wino2d.hip
    assert plan(32, 128, (56, 56)).wino
    assert plan(2, 64, (94, 311)).wino                # odd width: half-used last tile column (round 3)
    saved_odd = E.WINO2D["odd"]
    E.WINO2D["odd"] = False
    try:
        assert not plan(2, 64, (94, 311)).wino
    finally:
        E.WINO2D["odd"] = saved_odd
    assert plan(2, 256, (24, 78)).wino                # a deep backbone stage: few tile groups, still ahead of the direct kernel
    assert not plan(2, 512, (12, 38)).wino            # too few tile groups per cout group
    assert not plan(32, 32, (112, 112), stride=2).wino
    assert plan(32, 128, (56, 56), dil=2).wino            # dilated, 2d | H, W: d*d interleaved sub-grids on the Winograd kernel (round 3)
    assert not plan(32, 128, (54, 56), dil=2).wino        # ... otherwise the direct kernel
    saved_dil = E.WINO2D["dilated"]
    E.WINO2D["dilated"] = False
    try:
        assert not plan(32, 128, (56, 56), dil=2).wino
    finally:
        E.WINO2D["dilated"] = saved_dil
    saved = E.WINO2D["enabled"]
    E.WINO2D["enabled"] = False
    try:
        assert plan(32, 32, (112, 112)).kname.startswith("conv2ddirect")
    finally:
        E.WINO2D["enabled"] = saved


def test_workspace_pool_bucket_and_prefix_views():
    """Bounded workspaces: capacity buckets waste at most 1/3, a workspace of N <= cap units is a prefix
    view of the pool's tensor (same per-unit geometry, shared memory), and growing is the only way memory increases."""
    from disprcnn_amd import engine as E
    assert [E.bucket_units(n) for n in (0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 16, 17, 25, 33, 100, 256, 257)] == \
        [1, 1, 2, 3, 4, 6, 6, 8, 12, 12, 16, 16, 24, 32, 48, 128, 256, 384]
    assert all(n <= E.bucket_units(n) <= max(1.5 * n, 1) for n in range(1, 2000))
    dev = torch.device("cpu")
    pool = E.WorkspacePool(6, dev)
    a = pool.blocked("x", 4, 32, 3, 4, 5, 1, 1, 1)
    b = pool.blocked("x", 6, 32, 3, 4, 5, 1, 1, 1)
    assert a.storage.data_ptr() == b.storage.data_ptr() and a.n_stride == b.n_stride and a.numel < b.numel
    a.view6()[3, 1, 1, 1, 1, 7] = 5.0
    assert b.view6()[3, 1, 1, 1, 1, 7].item() == 5.0           # same memory
    n0 = pool.nbytes()
    pool.blocked("x", 1, 32, 3, 4, 5, 1, 1, 1); pool.dense("c", 5, 3, 4)
    assert pool.nbytes() == n0 + 4 * 6 * 12                      # a re-request allocates nothing; the dense tensor is sized for cap
    with pytest.raises(ValueError):
        pool.blocked("x", 7, 32, 3, 4, 5, 1, 1, 1)               # above capacity: the runtime replaces the pool
    with pytest.raises(ValueError):
        pool.blocked("x", 2, 16, 3, 4, 5, 1, 1, 1)               # same name, other geometry


def test_reference_init_statistics():
    """SURVEY a9 (stackhourglass.py:90-104): Conv2d / Conv3d weights ~ N(0, sqrt(2 / (prod(kernel) * out_channels))), BatchNorm
    gamma = 1 / beta = 0, and ConvTranspose3d -- not an nn.Conv3d instance -- keeps torch's default kaiming-uniform init
    (bound 1 / sqrt(fan_in), fan_in = weight.size(1) * prod(kernel))."""
    import math
    from torch import nn
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    torch.manual_seed(0)
    m = PSMNet(48, -48)
    checked = {"conv": 0, "deconv": 0, "bn": 0}
    for name, mod in m.named_modules():
        if isinstance(mod, nn.ConvTranspose3d):
            w = mod.weight
            bound = 1.0 / math.sqrt(w.shape[1] * math.prod(w.shape[2:]))
            assert w.abs().max().item() <= bound * (1 + 1e-6), name
            assert abs(w.std().item() / (bound / math.sqrt(3)) - 1) < 0.03 and abs(w.mean().item()) < 0.02 * bound, name
            checked["deconv"] += 1
        elif isinstance(mod, (nn.Conv2d, nn.Conv3d)):
            w = mod.weight
            std = math.sqrt(2.0 / (math.prod(mod.kernel_size) * mod.out_channels))
            if w.numel() >= 4096:
                assert abs(w.std().item() / std - 1) < 0.06, (name, w.std().item(), std)
                assert abs(w.mean().item()) < 0.05 * std, name
            assert mod.bias is None, name                                   # every conv on the path is bias-free (SURVEY 8a checklist)
            checked["conv"] += 1
        elif isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm3d)):
            assert torch.equal(mod.weight, torch.ones_like(mod.weight)) and torch.equal(mod.bias, torch.zeros_like(mod.bias)), name
            checked["bn"] += 1
    assert checked["deconv"] == 6 and checked["conv"] > 60 and checked["bn"] > 60, checked


def test_graphed_step_needs_the_gpu():
    """utils/graph.GraphedStep replays HIP graphs: without a GPU it raises instead of silently running the step eagerly."""
    import pytest
    import torch
    from disprcnn_amd.utils.graph import GraphedStep
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        GraphedStep(lambda: torch.zeros(1))


def test_stem_space_to_depth_weight_is_the_same_convolution():
    """backbone/runtime.py: the 7x7 stride-2 pad-3 stem as a 4x4 stride-1 convolution over the 2x2 pixel-unshuffled image (round 3) --
    exact re-indexing of the weights (float64, even and odd image sizes)."""
    from disprcnn_amd.modeling.backbone.runtime import _stem_s2d_weight
    g = torch.Generator().manual_seed(3)
    for H, W in ((10, 14), (11, 13), (7, 8)):
        x = torch.randn(2, 3, H, W, dtype=torch.float64, generator=g)
        w = torch.randn(5, 3, 7, 7, dtype=torch.float64, generator=g)
        ref = torch.nn.functional.conv2d(x, w, stride=2, padding=3)
        xs = torch.nn.functional.pixel_unshuffle(torch.nn.functional.pad(x, (0, W & 1, 0, H & 1)), 2)
        got = torch.nn.functional.conv2d(torch.nn.functional.pad(xs, (2, 1, 2, 1)), _stem_s2d_weight(w))
        assert got.shape[2] >= ref.shape[2] and got.shape[3] >= ref.shape[3]
        assert (got[:, :, :ref.shape[2], :ref.shape[3]] - ref).abs().max().item() < 1e-12


def test_fp16_kernel_selection():
    """Which fp16 kernel a layer takes (engine.ConvPlan16): the depth-sliding walk for one input block / <= 32 couts / depth >= 4; the
    cout-split, double-buffered kernels of conv16x.hip for the other 3x3x3 layers (stride 1, stride 2, transposed; round 4); conv16t.hip
    for the 3x3 layers; conv16.hip's generic tap walk for dilated and 1x1."""
    from disprcnn_amd import engine as E
    dev = torch.device("cpu")

    def geo(n, c, d, h, w, pad=1, pd=1):
        b = E.Blocked16.__new__(E.Blocked16)
        b.N, b.C, b.D, b.H, b.W, b.pd, b.ph, b.pw = n, c, d, h, w, pd, pad, pad
        b.cb = (c + 31) // 32
        b.Dp, b.Hp, b.Wp = d + 2 * pd, h + 2 * pad, w + 2 * pad
        b.h_stride = b.Wp * 32; b.d_stride = b.Hp * b.h_stride; b.cb_stride = b.Dp * b.d_stride; b.n_stride = b.cb * b.cb_stride
        b.device = dev
        return b

    assert E.plan_conv3d16(geo(4, 32, 24, 56, 56), geo(4, 32, 24, 56, 56), 1, 32, True).kname == "conv16sp_kernel<7,2>"         # 56 rows = two full 28-row tiles: the counted-wait walk
    assert E.plan_conv3d16(geo(4, 32, 24, 20, 56), geo(4, 32, 24, 20, 56), 1, 32, True).kname == "conv16s_kernel<2,2>"          # ragged rows: conv16s
    assert E.plan_conv3d16(geo(4, 32, 24, 32, 56), geo(4, 32, 24, 32, 56), 1, 32, True).kname == "conv16sp_kernel<4,2>"
    assert E.plan_conv3d16(geo(4, 64, 24, 56, 56), geo(4, 32, 24, 56, 56), 1, 32, True).kname == "conv16sw_kernel<7,2>"       # two input blocks, 56 = 4 x 14 rows: the two-block depth walk
    assert E.plan_conv3d16(geo(4, 64, 24, 20, 56), geo(4, 32, 24, 20, 56), 1, 32, True).kname == "conv16d_kernel<2,2,1>"      # ragged rows: staged kernel
    assert E.plan_conv3d16(geo(4, 64, 12, 28, 28), geo(4, 64, 12, 28, 28), 1, 64, True).kname == "conv16sw_kernel<7,4>"
    assert E.plan_conv3d16(geo(4, 96, 12, 28, 28), geo(4, 64, 12, 28, 28), 1, 64, True).kname == "conv16d_kernel<7,4,1>"       # three input blocks
    assert E.plan_conv3d16(geo(4, 32, 3, 28, 28), geo(4, 32, 3, 28, 28), 1, 32, True).kname == "conv16d_kernel<7,2,1>"        # depth < 4
    assert E.plan_conv3d16(geo(4, 32, 24, 56, 56), geo(4, 64, 12, 28, 28), 2, 64, True).kname == "conv16d_kernel<7,4,2>"
    assert E.plan_conv3d16(geo(4, 64, 12, 28, 28), geo(4, 64, 6, 14, 14), 2, 64, True).kname == "conv16d_kernel<7,4,2>"
    assert E.plan_conv3d16(geo(4, 32, 12, 12, 12), geo(4, 32, 6, 6, 6), 2, 32, True).kname == "conv16d_kernel<4,2,2>"        # 6 rows: one 8-row tile
    assert E.plan_deconv3d16(geo(4, 64, 12, 28, 28), geo(4, 32, 24, 56, 56), 32, False).kname == "conv16u_kernel<7,2,2>"
    assert E.plan_deconv3d16(geo(4, 64, 6, 14, 14), geo(4, 64, 12, 28, 28), 64, False).kname == "conv16u_kernel<7,4,2>"
    assert E.x16_rows(4, 4, 1) == 2 and E.x16_rows(6, 4, 1) == 7 and E.x16_rows(28, 4, 1) == 7 and E.x16_rows(8, 1, 2) == 2 and E.x16_rows(28, 1, 0, cb=4) == 2
    assert E.plan_conv3d16_cout1(geo(4, 32, 24, 56, 56)).kname == "conv16sp_kernel<7,1>"
    x2 = geo(8, 64, 1, 56, 56, pad=1, pd=0)
    assert E.plan_conv2d16(x2, geo(8, 64, 1, 56, 56, pd=0), 3, 1, 1, 1, 64, True).kname == "conv16t_kernel<4,4,1>"
    assert E.plan_conv2d16(geo(8, 128, 1, 56, 56, pad=2, pd=0), geo(8, 128, 1, 56, 56, pd=0), 3, 1, 2, 2, 128, True).kname.startswith("conv16_kernel")
    assert E.plan_conv2d16(x2, geo(8, 64, 1, 56, 56, pd=0), 1, 1, 0, 1, 64, False).kname.startswith("conv16_kernel")
    saved = E.C16_TILE["enabled"]
    E.C16_TILE["enabled"] = False
    try:
        assert E.plan_conv3d16(geo(4, 32, 24, 56, 56), geo(4, 32, 24, 56, 56), 1, 32, True).kname.startswith("conv16_kernel")
    finally:
        E.C16_TILE["enabled"] = saved


def test_packed_weight_cache_is_tied_to_the_live_parameter():
    """the GEMM operand cache must not be keyed by id() alone (ids and allocator addresses are reused once a model is freed)
    and must not keep packed copies of dead models alive."""
    import gc

    import torch

    from disprcnn_amd.modeling import head_ops as H
    lin = torch.nn.Linear(8, 4)
    cache = H._packed_cache(lin.weight)
    cache["tag"] = ("key", "packed", (4, 8))
    assert H._packed_cache(lin.weight) is cache
    H.clear_packed_weights(lin)
    assert H._packed_cache(lin.weight) == {}
    k = id(lin.weight)
    H._packed_cache(lin.weight)["tag"] = 1
    # an impostor entry under a recycled id is not trusted: the weak reference has to point at THIS parameter
    other = torch.nn.Parameter(torch.zeros(4, 8))
    H._PACKED_W[id(other)] = H._PACKED_W[k]
    assert H._packed_cache(other) == {}
    del lin, cache, other
    gc.collect()
    assert k not in H._PACKED_W


def test_fused_head_weight_packing_puts_the_depth_taps_of_a_tap_in_one_lane():
    """s16.pack_head_weight_s16 (the 32 -> 1 layer as the MFMA A operand of the fused head, convs16.hip HEAD form): emulating the MFMA's
    row / k-group layout, product row m of a voxel is tap head_rows()[m] applied to the voxel's 32 channels; every tap appears once; lane
    half g' holds (kd, j) in register 3 (j - 5 g') + kd."""
    from disprcnn_amd import s16
    rows = s16.head_rows()
    taps = [r for r in rows if r is not None]
    assert len(taps) == 27 and len(set(taps)) == 27
    for g in range(2):
        for r in range(16):
            m = (r & 3) + 8 * (r >> 2) + 4 * g
            jj, kd = divmod(r, 3)
            assert rows[m] == ((kd, 5 * g + jj) if jj < (5 if g == 0 else 4) else None)
    w1 = torch.randn(1, 32, 3, 3, 3, generator=torch.Generator().manual_seed(5))
    hp, wexp = s16.pack_head_weight_s16(w1)
    assert tuple(hp.shape) == (2, 2, 64, 8) and hp.dtype == torch.float16
    a = torch.randn(32, generator=torch.Generator().manual_seed(6)).double()
    A = (hp[:, 0].double() + hp[:, 1].double()) * 2.0 ** -wexp                     # [slice][lane][e]
    P = torch.zeros(32, dtype=torch.float64)
    for s_ in range(2):
        for g in range(2):
            for e in range(8):
                c = 4 * g + 8 * (2 * s_ + (e >> 2)) + (e & 3)                      # the channel a finishing wave holds in (slice, g, e)
                P += A[s_, 32 * g: 32 * g + 32, e] * a[c]
    for m, r in enumerate(rows):
        want = 0.0 if r is None else float((w1[0, :, r[0], r[1] // 3, r[1] % 3].double() * a).sum())
        assert abs(float(P[m]) - want) < 1e-5 * max(1.0, abs(want))               # (hi + lo carries 22 bits of the scaled weight)


def test_split_f16_shape_rules_through_the_c_abi():
    """The *_supported entry points of the split-f16 kernels are host code: their shape rules hold without a GPU (the launch entries repeat
    them and return -4)."""
    from disprcnn_amd import _lib
    lib = _lib.lib()
    s1, s2, up, c2 = lib.drc_conv3d_k3_s16_supported, lib.drc_conv3d_k3s2_s16_supported, lib.drc_deconv3d_k3s2_s16_supported, lib.drc_conv2d_k3_s16_supported
    # 3D stride 1 (round 6): cin / cout in {32, 64}, ANY D, H, W > 0 -- D that is not a multiple of 3 walks phantom zero planes, widths that are
    # not 7 | 14 | a multiple of 28 mask their last tile (the reference's contract: D, H, W = 0 mod 4 at 1/4 resolution, stackhourglass.py:115-128)
    assert s1(32, 32, 12, 28, 28) and s1(64, 32, 24, 56, 56) and s1(64, 64, 6, 14, 14) and s1(64, 64, 3, 7, 7) and s1(32, 64, 6, 5, 7)
    assert all(s1(32, 32, d, 32, w) for d in range(4, 49, 4) for w in (16, 28, 32, 40, 56, 64))
    assert s1(32, 32, 10, 28, 28) and s1(32, 32, 12, 27, 28) and s1(32, 32, 12, 28, 30) and s1(64, 64, 1, 1, 1)
    assert not s1(48, 32, 12, 28, 28) and not s1(32, 96, 12, 28, 28) and not s1(32, 32, 0, 28, 28) and not s1(32, 32, 12, 28, 0)
    # stride 2: even input dims; transposed: cin 64; any size (masked last tiles)
    assert s2(32, 64, 12, 28, 28) and s2(64, 64, 6, 14, 14) and s2(32, 64, 24, 56, 56) and s2(32, 64, 12, 28, 30) and s2(64, 64, 16, 32, 32)
    assert not s2(32, 64, 11, 28, 28) and not s2(32, 64, 12, 27, 28) and not s2(48, 64, 12, 28, 28)
    assert up(64, 32, 6, 14, 14) and up(64, 64, 3, 7, 7) and up(64, 32, 12, 28, 28) and up(64, 32, 6, 14, 15) and up(64, 64, 4, 8, 8)
    assert not up(32, 32, 6, 14, 14) and not up(64, 48, 6, 14, 14)
    # 2D: any map size at dilation 1, cin in {32, 64, 128} (wider layers: chained launches), cout a power of two in 32..512; dilation 2: cin 128, 56-row blocks
    assert c2(64, 64, 56, 56, 1) and c2(32, 32, 29, 57, 1) and c2(128, 512, 12, 39, 1) and c2(128, 256, 94, 310, 1)
    assert not c2(256, 64, 28, 28, 1) and not c2(128, 96, 28, 28, 1) and not c2(128, 1024, 28, 28, 1) and not c2(64, 64, 0, 28, 1)
    assert c2(128, 128, 56, 56, 2) and not c2(128, 128, 28, 56, 2) and not c2(64, 64, 56, 56, 2) and not c2(128, 128, 56, 56, 3)


def test_bridged_conv_slices_and_threshold():
    """engine.BridgedConv2dS16: input-channel slices of the chained launches and the large-maps-only rule (TRUNK_S16)."""
    from disprcnn_amd import engine as E
    B = E.BridgedConv2dS16
    assert B.slices(64) == ((0, 64),) and B.slices(256) == ((0, 128), (128, 256)) and B.slices(512)[-1] == (384, 512) and B.slices(96) is None
    saved = dict(E.TRUNK_S16)
    try:
        E.TRUNK_S16.update(enabled=True, min_tiles=192, min_rows=24)
        assert B.worth(2, 256, 256, 94, 310) and B.worth(2, 256, 512, 47, 155) and B.worth(2, 64, 64, 94, 310)       # FPN P2, RPN on P3, layer1
        assert not B.worth(2, 256, 256, 24, 78) and not B.worth(2, 512, 512, 12, 39) and not B.worth(2, 256, 96, 94, 310)
        E.TRUNK_S16["enabled"] = False
        assert not B.worth(2, 256, 256, 94, 310)
    finally:
        E.TRUNK_S16.update(saved)


def test_overflow_policy_backs_off_exponentially_and_recovers():
    """engine.OverflowPolicy: after an overflow the next 2^level passes skip the split-f16 kernels (capped), a clean fast pass resets."""
    from disprcnn_amd import engine as E
    pol = E.OverflowPolicy(cap=8)
    assert pol.want_fast()
    pol.report(True)                                   # first overflow: skip 1
    assert [pol.want_fast() for _ in range(3)] == [False, True, True]
    pol.report(True)                                   # second in a row: skip 2
    assert [pol.want_fast() for _ in range(3)] == [False, False, True]
    for _ in range(6):
        pol.report(True)
    assert pol.skip == 8 and pol.overflows == 8        # capped
    while not pol.want_fast():
        pass
    pol.report(False)                                  # a clean split-f16 pass: back to the start
    pol.report(True)
    assert pol.skip == 1


def test_guarded_control_flow_without_a_gpu():
    """engine.guarded with a stand-in guard: clean pass -> one call; overflow -> the pass is repeated with s16_allowed() False ("auto") or
    raises (strict); nested scopes leave the check to the outermost; the back-off passes run with s16_allowed() False from the start."""
    import warnings
    from disprcnn_amd import engine as E

    class FakeGuard:
        def __init__(self, trips):
            self.policy, self.used, self.warned, self.trips, self.reads = E.OverflowPolicy(), False, False, list(trips), 0

        def tripped(self):
            self.reads += 1
            return self.trips.pop(0) if self.trips else False

    calls = []

    def fn(g):
        def run():
            calls.append(E.s16_allowed())
            if E.s16_allowed():
                g.used = True                           # (a split-f16 launch picked the word up)
            return len(calls)
        return run

    g = FakeGuard([False])
    assert E.guarded(g, fn(g)) == 1 and calls == [True] and g.reads == 1 and E.guard_in_scope() is None
    calls.clear()
    g = FakeGuard([True])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert E.guarded(g, fn(g), what="unit") == 2
    assert calls == [True, False] and len(w) == 1 and "unit" in str(w[0].message) and E.s16_allowed()
    calls.clear()
    assert E.guarded(g, fn(g)) == 1 and calls == [False] and g.reads == 1          # back-off pass: fp32 from the start, no read
    assert E.guarded(g, fn(g)) == 2 and calls == [False, True]                       # then split-f16 again
    g = FakeGuard([True])
    with pytest.raises(RuntimeError, match="f16x2"):
        E.guarded(g, fn(g), strict=True)
    assert E.guard_in_scope() is None and E.s16_allowed()
    # nested: the inner component reports to the outer guard and does not check
    outer, inner = FakeGuard([False]), FakeGuard([True])
    seen = []

    def outer_fn():
        seen.append(E.guard_in_scope() is outer)
        outer.used = True
        return E.guarded(inner, lambda: seen.append(E.guard_in_scope() is outer) or 7)
    assert E.guarded(outer, outer_fn) == 7 and seen == [True, True] and inner.reads == 0 and outer.reads == 1
    # disabled / no guard: plain call, no scope
    assert E.guarded(None, lambda: E.guard_in_scope()) is None and E.guarded(FakeGuard([]), lambda: E.guard_in_scope(), enabled=False) is None
    # a pass without split-f16 launches is not read back
    g = FakeGuard([True])
    assert E.guarded(g, lambda: 3) == 3 and g.reads == 0
    # an exception inside the pass: the scope closes, and the guard remembers that its word may hold a flag nobody read
    g = FakeGuard([])

    def boom():
        g.used = True
        raise KeyError("x")
    with pytest.raises(KeyError):
        E.guarded(g, boom)
    assert E.guard_in_scope() is None and E.s16_allowed() and g.dirty


def test_newest_profile_set_is_complete_and_from_one_commit():
    """no empty PMC tables, every summary of a round's set from the same commit (profiles/collect_all.sh writes it)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_profiles", os.path.join(root, "profiles", "check_profiles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check("r4") == []
    assert mod.check("r5c") == []                # the final set of round 5 (seven workloads, one commit)
    assert mod.check("r6") == []                 # round 6 (the same seven workloads; its traffic files carry the digest of the kernel sources)


def test_roofline_traffic_is_tied_to_the_kernel_sources():
    """bench.py reads roofline.traffic from a committed PMC pass; the pass records a digest of the kernel sources
    (profiles/summarize.py) and bench.py nulls the figure when the sources it runs differ.  Held here: the newest headline traffic file
    carries a digest, and it is the digest of the sources in this tree (a kernel edit without a re-collected profile fails this test --
    or must lower the claim by deleting the stale file)."""
    import json
    import os
    from disprcnn_amd.csrc.build import source_digest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(root, "profiles", "r6_traffic.json"))).get("__meta__") or {}
    assert meta.get("csrc_sha") and len(meta.get("commit", "")) >= 7
    assert meta["csrc_sha"] == source_digest(), "kernel sources changed since profiles/r6_* were collected: re-run profiles/collect_all.sh r6"

