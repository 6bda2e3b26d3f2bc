"""GPU parity of the kernels the BENCH runs.

At the golden fixtures' own batch size (2 ROIs) the launch heuristics send the stride-1 3x3x3 layers to the generic kernel, so
the reference-recorded whole-path goldens never touched wino3d / the LDS-free kernels.  Here the recorded inputs are
replicated to the bench's batch (bench.DEFAULT_ROIS ROI pairs per step; 16 crops for Config B) -- ROIs are independent units, so every replica
must reproduce the reference's output -- and the kernel names of every launch plan are asserted to be exactly the bench's.
Tolerances as in test_hip_parity.py: mean <= 1e-3 px, max <= 2e-2 px vs the reference's fp32 outputs.
"""
import pytest
import torch

from disprcnn_amd import engine as E

from oracle import psmnet_oracle as O
from disprcnn_amd.utils import synth
from tests.helpers import golden_npz, state_for

pytestmark = pytest.mark.gpu

WINO_LAYERS = ["dres0.0", "dres0.2", "dres1.0", "dres1.2", "hg1.conv2", "hg2.conv2", "hg3.conv2", "classif1.0", "classif2.0", "classif3.0"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _model(dev, case, mx, mn, math="f32"):
    """math = "f32": every 3D layer on the fp32 MFMA kernels (rounds 1-4); "auto": the round-5 default, the full-resolution stride-1
    layers in split-f16 arithmetic (convs16.hip)."""
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(mx, mn)
    m.load_state_dict(state_for(case), strict=True)
    m.regressor_math = math
    m.feature_math = math           # the same switch for the 2D CNN's stride-1 undilated 3x3 layers (convs16r.hip)
    return m.to(dev).eval()


# (plan names: the last two template flags -- residual, blocked fp32 output -- follow the call's arguments and read false,false here)
S16_LAYERS = {"dres0.0": "convs16w_kernel<4,true>",          # (round 6: the cost-volume layer runs two tiles per wave at the bench's batch, convs16w.hip) "dres0.2": "convs16_kernel<2,false,1,28,false,false,false>",
              "dres1.0": "convs16_kernel<2,false,1,28,false,false,false>", "dres1.2": "convs16_kernel<2,false,1,28,false,false,false>",
              "classif1.0": "convs16_kernel<2,false,1,28,false,false,false>", "classif2.0": "convs16_kernel<2,false,1,28,false,false,false>",
              "classif3.0": "convs16_kernel<2,false,1,28,false,false,false>"}
S16_HG = {"conv1": "convs16d_kernel<2,2,14,3,true,true>", "conv2": "convs16_kernel<4,false,2,14,false,false,false>", "conv3": "convs16d_kernel<4,4,7,2,true,false>",
          "conv4": "convs16_kernel<4,false,4,7,false,false,false>", "conv5": "convs16u_kernel<4,7>", "conv6": "convs16u_kernel<2,14>"}


def _assert_bench_kernels_s16(ws):
    p = ws["p"]
    for name, kname in S16_LAYERS.items():
        assert p[name].kname == kname, (name, p[name].kname)
    for k in (1, 2, 3):
        for c, kname in S16_HG.items():
            assert p[f"hg{k}.{c}"].kname == kname, (k, c, p[f"hg{k}.{c}"].kname)
    return {n: pl.kname for n, pl in p.items()}


def _assert_bench_kernels_s16_b(ws):
    """Config B's volume (24 x 56 x 56): every layer of the regressor on the split-f16 kernels (1 x 28 tiles at full resolution and in the
    hourglass' half-resolution maps, 2 x 14 in its quarter-resolution ones)."""
    for name, pl in ws["p"].items():
        assert pl.kname.startswith(("convs16_kernel", "convs16w_kernel", "convs16d_kernel", "convs16u_kernel")), (name, pl.kname)


def _assert_bench_kernels(ws):
    p = ws["p"]
    for name in WINO_LAYERS:
        assert p[name].wino and p[name].kname.startswith(("wino3d_rb_kernel", "wino3d_kernel")), (name, p[name].kname)
    for k in (1, 2, 3):
        assert p[f"hg{k}.conv4"].direct and p[f"hg{k}.conv4"].slide and not p[f"hg{k}.conv4"].wino      # 3x7x7: odd dims -> tapdirect
        assert p[f"hg{k}.conv1"].kname.startswith("downdirect_kernel") and p[f"hg{k}.conv3"].kname.startswith("downdirect_kernel")
        assert p[f"hg{k}.conv5"].deconv_direct and p[f"hg{k}.conv6"].deconv_direct
    return {n: pl.kname for n, pl in p.items()}


def test_bench_batch_executes_the_fused_head_and_residual_instantiations_by_name(dev):
    """a plan's `kname` carries the template flags of a plain call; the flags of the launch that actually ran (residual,
    blocked-fp32 output, fused cout-1 head) follow the call's arguments.  At the bench's batch, the names the engine records per executed
    launch (engine.TIMING, what bench.py's roofline reads) must be exactly the bench's: three fused-head launches -- the dominant kernel of
    BENCH_r05 -- one residual form (dres1[2] + cost0a), the cost-volume form, and the hourglass set."""
    import bench
    from collections import Counter
    m = _model(dev, "A", 48, 0, "auto")
    N = bench.DEFAULT_ROIS
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="caseA")
    fl, fr = fl.repeat(N // 2, 1, 1, 1).to(dev), fr.repeat(N // 2, 1, 1, 1).to(dev)
    with torch.no_grad():
        m.forward_from_features(fl, fr, (112, 112))
        E.TIMING = []
        try:
            m.forward_from_features(fl, fr, (112, 112))
        finally:
            tim, E.TIMING = E.TIMING, None
    torch.cuda.synchronize()
    c = Counter(t[0] for t in tim)
    want = {"convs16_kernel<2,false,1,28,false,false,true>": 3,        # classif1..3[0] + the 32 -> 1 layer behind it
            "convs16_kernel<2,false,1,28,true,false,false>": 1,        # dres1[2] + cost0a
            "convs16_kernel<2,false,1,28,false,false,false>": 2,       # dres0[2], dres1[0]
            "convs16w_kernel<4,true>": 1,                              # dres0[0] on the virtual cost volume (two tiles per wave, convs16w.hip)
            "convs16d_kernel<2,2,14,3,true,true>": 3, "convs16d_kernel<4,4,7,2,true,false>": 3,
            "convs16_kernel<4,false,2,14,false,false,false>": 1, "convs16_kernel<4,false,2,14,true,false,false>": 2,     # conv2 (+ postsqu in dres3 / dres4)
            "convs16_kernel<4,false,4,7,false,false,false>": 3, "convs16u_kernel<4,7>": 3, "convs16u_kernel<2,14>": 3}
    got = {k: v for k, v in c.items() if k.startswith("convs16")}
    assert got == want, got


@pytest.mark.parametrize("math", ["f32", "auto"])
@pytest.mark.parametrize("case,mx,mn", [("A", 48, 0), ("At", 48, 0), ("A2", 24, -24)])
def test_config_a_golden_replicated_to_bench_batch(dev, case, mx, mn, math):
    """The bench's batch (bench.DEFAULT_ROIS ROI pairs per step) = the recorded 2-ROI input replicated: the bench's plans (Winograd
    on the ten even stride-1 layers, LDS-free direct / stride-2 kernels, LDS-free fused transposed conv) reproduce the reference's
    recorded disparities and sampled intermediates in EVERY replica."""
    import bench
    z = golden_npz()
    m = _model(dev, "At" if case == "At" else "A", mx, mn, math)
    fl, fr = synth.synth_features(2, 32, 28, 28, tag="caseA")
    N = bench.DEFAULT_ROIS
    fl, fr = fl.repeat(N // 2, 1, 1, 1).to(dev), fr.repeat(N // 2, 1, 1, 1).to(dev)
    with torch.no_grad():
        pred = m.forward_from_features(fl, fr, (112, 112)).cpu()
    Dp = (mx - mn) // 4
    if math == "f32":
        ws = m._rt._ws[("3d", N, Dp, 28, 28)]
        names = _assert_bench_kernels(ws)
        assert names["hg1.conv1"] == "downdirect_kernel<7,4>" and names["dres1.0"] == "wino3d_rb_kernel<14>" and names["hg1.conv2"] == "wino3d_rb_kernel<7>", names
    else:
        ws = m._rt._ws[("3ds16", N, Dp, 28, 28)]
        names = _assert_bench_kernels_s16(ws)
    ref = torch.from_numpy(z[f"{case}_pred"])
    err = (pred.view(N // 2, 2, 112, 112) - ref[None]).abs()
    print(case, f"N={N} mean/max err px", err.mean().item(), err.max().item())
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (err.mean().item(), err.max().item())
    # replicas agree with each other bitwise (no cross-ROI state, FMA order independent of the position in the batch)
    assert torch.equal(pred[0:2], pred[N - 2:N]) and torch.equal(pred[0:2], pred[100:102])
    vox = Dp * 28 * 28
    cost3 = ws["t"]["costk3"].cpu().reshape(N // 2, 2 * vox)
    idx = torch.from_numpy(z[f"{case}_cost3_idx"]); val = torch.from_numpy(z[f"{case}_cost3_val"])
    for rep in (0, 63, N // 2 - 1):
        assert (cost3[rep][idx] - val).abs().max().item() < 1e-3 * max(1.0, val.abs().max().item())
    out3 = ws["t"]["out3"].to_dense().cpu().reshape(N // 2, -1)
    idx = torch.from_numpy(z[f"{case}_out3_idx"]); val = torch.from_numpy(z[f"{case}_out3_val"])
    for rep in (0, 63, N // 2 - 1):
        assert (out3[rep][idx] - val).abs().max().item() < 1e-4 * max(1.0, val.abs().max().item()) + 1e-4


@pytest.mark.parametrize("math", ["f32", "auto"])
def test_config_a_bench_batch_distinct_rois_vs_oracle_subset(dev, math):
    """One bench-sized step on DISTINCT synthetic ROI pairs (the bench's own inputs, bench.DEFAULT_ROIS of them); 64 ROIs spread over the
    batch are checked against the CPU oracle (pinned to the reference by tests/test_oracle_golden.py).  The round-1 batch of 256
    is checked the same way: the launch heuristics depend on the batch."""
    import bench
    sd = state_for("A")
    m = _model(dev, "A", 48, 0, math)
    for N in (bench.DEFAULT_ROIS, 256):
        fl, fr = synth.synth_features(N, 32, 28, 28, tag="bench0")
        with torch.no_grad():
            pred = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        if math == "f32":
            _assert_bench_kernels(m._rt._ws[("3d", N, 12, 28, 28)])
        else:
            _assert_bench_kernels_s16(m._rt._ws[("3ds16", N, 12, 28, 28)])
        pick = sorted({int(i) for i in torch.linspace(0, N - 1, 64).round().tolist()} | {0, 37, N // 2, N - 1})
        with torch.no_grad():
            ref = O.psmnet_from_features(sd, fl[pick], fr[pick], 48, 0, 112, 112)
        err = (pred[pick] - ref).abs()
        print(f"N={N}: {len(pick)} distinct ROIs, mean/max err px", err.mean().item(), err.max().item())
        assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (N, err.mean().item(), err.max().item())
        assert err.reshape(len(pick), -1).mean(1).max().item() < 1e-3          # ... for every single ROI, not only on average
        assert torch.isfinite(pred).all() and pred.min() >= 0 and pred.max() <= 47


# the plans of one image's ROIs (BASELINE configs[1]: 16 ROIs) and of four images': what bench.py's batch_sensitivity entries run
SMALL_BATCH_PLANS = {
    16: {"dres1.0": "wino3d_rb_kernel<14>", "hg1.conv1": "downdirect_kernel<7,1>", "hg1.conv2": "tapdirect_kernel<7,1>", "hg1.conv4": "tapdirect_kernel<4,1>"},
    64: {"dres1.0": "wino3d_rb_kernel<14>", "hg1.conv1": "downdirect_kernel<7,4>", "hg1.conv2": "wino3d_rb_kernel<7>", "hg1.conv4": "tapdirect_kernel<4,1>"},
}


@pytest.mark.parametrize("N", [16, 64])
def test_config_a_small_batches_vs_oracle_and_plan_names(dev, N):
    """Config A at the batch of ONE image (16 ROIs, BASELINE configs[1]) and of four: every ROI against the CPU oracle, and the kernels the
    launch heuristics pick there (they differ from the bench batch's: fewer tile groups than CUs on the half-resolution maps)."""
    sd = state_for("A")
    m = _model(dev, "A", 48, 0)
    fl, fr = synth.synth_features(N, 32, 28, 28, tag=f"small{N}")
    with torch.no_grad():
        pred = m.forward_from_features(fl.to(dev), fr.to(dev), (112, 112)).cpu()
        ref = O.psmnet_from_features(sd, fl, fr, 48, 0, 112, 112)
    err = (pred - ref).abs()
    print(f"N={N}: all ROIs, mean/max err px", err.mean().item(), err.max().item())
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2 and err.reshape(N, -1).mean(1).max().item() < 1e-3
    plans = m._rt._ws[("3d", N, 12, 28, 28)]["p"]
    got = {k: plans[k].kname for k in SMALL_BATCH_PLANS[N]}
    print(f"N={N} plans:", {k: pl.kname for k, pl in plans.items()})
    assert got == SMALL_BATCH_PLANS[N], got


@pytest.mark.parametrize("math", ["f32", "auto"])
def test_config_b_golden_replicated_to_16_crops(dev, math):
    """Config B (full PSMNet on 224x224 crops, D=96) at the bench extra's batch of 16 ROI pairs = the recorded 2 crops x 8; "auto" (the
    default): the regressor and the 45 stride-1 undilated 3x3 layers of the 2D CNN in split-f16 arithmetic."""
    z = golden_npz()
    m = _model(dev, "B", 48, -48, math)
    left, right = synth.synth_images(2, 224, 224, tag="caseB")
    with torch.no_grad():
        pred = m((left.repeat(8, 1, 1, 1).to(dev), right.repeat(8, 1, 1, 1).to(dev))).cpu()
    if math == "f32":
        ws3 = m._rt._ws[("3d", 16, 24, 56, 56)]
        for name in WINO_LAYERS:
            assert ws3["p"][name].wino, name
        ws2 = m._rt._ws[("2d", 32, 224, 224)]
        assert ws2["p"]["fe.firstconv.2"].wino and ws2["p"]["fe.layer4.0.conv1"].wino
    else:
        _assert_bench_kernels_s16_b(m._rt._ws[("3ds16", 16, 24, 56, 56)])
        ws2 = m._rt._ws[("2ds16", 32, 224, 224)]
        names = [e[1].kname for e in ws2["sched"] if e[0] == "s16"]
        assert len(names) == 51, len(names)
        assert names.count("convs16r_kernel<2,1,false,1>") == 8 and names.count("convs16r_kernel<4,1,false,1>") == 31, names      # firstconv + layer1; layer2
        assert names.count("convs16r_kernel<2,2,false,1>") == 1 and names.count("convs16r_kernel<4,2,false,1>") == 5, names       # layer3
        assert names.count("convs16r_kernel<4,2,false,2>") == 6, names                                                            # layer4 (dilation 2)
    assert ws2["p"]["fe.lastconv.0"].wino and ws2["p"]["fe.layer2.0.conv1"].kname.startswith("conv2ddirect")
    ref = torch.from_numpy(z["B_pred"])
    err = (pred.view(8, 2, 224, 224) - ref[None]).abs()
    print("B x8 mean/max err px", err.mean().item(), err.max().item())
    assert err.mean().item() < 1e-3 and err.max().item() < 2e-2, (err.mean().item(), err.max().item())
    assert torch.equal(pred[0:2], pred[14:16])


# ------------------------------------------------------------------------------------------------ bounded workspaces
def test_workspace_memory_flat_over_roi_counts(dev):
    """Detection output has a different ROI count on every image (BASELINE configs[4]).  Twenty different counts through one
    model must not grow HBM beyond the pool sized for the largest (VERDICT r1 weak #12 / ADVICE: unbounded per-N caches), and
    a pooled prefix-view workspace must give bitwise the results of a fresh model that only ever saw that count."""
    counts = [7, 30, 1, 12, 19, 3, 25, 9, 14, 2, 28, 5, 17, 22, 11, 6, 27, 4, 21, 16]
    fl, fr = synth.synth_features(30, 32, 28, 28, tag="wsflat")
    fl, fr = fl.to(dev), fr.to(dev)
    m = _model(dev, "A", 48, 0)
    m.graph_eval = False              # (captured graphs keep their outputs in a private pool: this test measures the workspaces alone)
    outs = {}
    torch.cuda.synchronize()
    with torch.no_grad():
        m.forward_from_features(fl[:30], fr[:30], (112, 112))             # largest count first: sizes the pool (bucket 32)
        torch.cuda.synchronize()
        base_bytes, base_alloc = m._rt.workspace_bytes(), torch.cuda.memory_allocated()
        for n in counts:
            outs[n] = m.forward_from_features(fl[:n], fr[:n], (112, 112)).cpu()
        torch.cuda.synchronize()
        # the one tensor the pool builds late: the materialised 64-channel volume (round 3: allocated on first use -- the fused eval path
        # of the big counts never touches it, the generic kernel of the smallest counts does), once, at the pool's capacity
        cost_bytes = 4 * E.Blocked.geometry(32, 64, 12, 28, 28, 1, 1, 1, dev).numel
        first_bytes, first_alloc = m._rt.workspace_bytes(), torch.cuda.memory_allocated()
        assert first_bytes - base_bytes in (0, cost_bytes + 4 * E.SLACK_FLOATS), (first_bytes, base_bytes, cost_bytes)
        for n in counts:
            m.forward_from_features(fl[:n], fr[:n], (112, 112))
        torch.cuda.synchronize()
    assert m._rt.workspace_bytes() == first_bytes, "a ROI count seen before allocated workspace memory"
    assert torch.cuda.memory_allocated() - first_alloc < 8 << 20, (torch.cuda.memory_allocated(), first_alloc)   # only small per-call outputs
    assert len(m._rt._pools) == 1 and len(m._rt._ws) == len(set(counts))
    for n in (3, 12, 30):
        fresh = _model(dev, "A", 48, 0)
        with torch.no_grad():
            ref = fresh.forward_from_features(fl[:n], fr[:n], (112, 112)).cpu()
        assert torch.equal(outs[n], ref), n
    # growing: a count above the capacity replaces the pool (old views dropped), memory = the bigger pool only
    fl2, fr2 = synth.synth_features(40, 32, 28, 28, tag="wsflat2")
    with torch.no_grad():
        m.forward_from_features(fl2.to(dev), fr2.to(dev), (112, 112))
    assert len(m._rt._pools) == 1 and list(m._rt._pools.values())[0].cap == 48
    assert all(w["pool"] is list(m._rt._pools.values())[0] for w in m._rt._ws.values())
    assert m._rt.workspace_bytes() <= base_bytes * 48 // 32 + (1 << 20)


def test_forwards_in_flight_and_gradient_accumulation(dev):
    """Round 3 (VERDICT r2 missing #6): a train-mode forward keeps its workspace slot until its backward ran, so a second forward
    (another batch of the same geometry, an eval pass) no longer clobbers the saved activations -- plain autograd semantics
    (reference engine/trainer.py:101-115).  Two batches forwarded back to back and one backward over the summed loss give, bit for bit,
    the gradients of forward/backward/forward/backward accumulation; slots are released by backward or when the outputs are dropped;
    more than MAX_SLOTS pending forwards raise."""
    import gc
    from disprcnn_amd.modeling.psmnet import runtime as R
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, 0)
    m.load_state_dict(state_for("A"), strict=True)
    m = m.to(dev).train()
    fl, fr = synth.synth_features(4, 32, 28, 28, tag="gen")
    fl, fr = fl.to(dev), fr.to(dev)
    a, b = (fl[:2], fr[:2]), (fl[2:], fr[2:])
    loss = lambda p: p[0].sum() + 0.7 * p[1].sum() + 0.5 * p[2].sum()
    params = [p for p in m.parameters() if p.requires_grad and p.grad is None]

    def grads():
        g = [None if p.grad is None else p.grad.clone() for p in m.parameters()]
        for p in m.parameters():
            p.grad = None
        return g

    # sequential accumulation
    loss(m.forward_from_features(*a, (112, 112))).backward()
    loss(m.forward_from_features(*b, (112, 112))).backward()
    seq = grads()
    assert not m._rt._held
    # both in flight, with an eval pass in between
    pa = m.forward_from_features(*a, (112, 112))
    assert list(m._rt._held) == [0]
    with torch.no_grad():
        m.forward_from_features(fl[:3], fr[:3], (112, 112))               # takes slot 1, holds nothing
    pb = m.forward_from_features(*b, (112, 112))
    assert sorted(m._rt._held) == [0, 1]
    (loss(pa) + loss(pb)).backward()
    assert not m._rt._held
    fly = grads()
    n = 0
    for g0, g1 in zip(seq, fly):
        assert (g0 is None) == (g1 is None)
        if g0 is not None:
            assert torch.equal(g0, g1)
            n += 1
    assert n > 50
    # dropped outputs free their slot; too many pending forwards raise
    pend = [m.forward_from_features(*a, (112, 112)) for _ in range(R.MAX_SLOTS)]
    with pytest.raises(RuntimeError, match="waiting for their backward"):
        m.forward_from_features(*a, (112, 112))
    del pend
    gc.collect()
    assert not m._rt._held
    p2 = m.forward_from_features(fl, fr, (112, 112))
    (p2[0].sum() + p2[1].sum() + p2[2].sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.dres0.parameters())


def test_eval_shortcuts_of_the_2d_cnn_match_the_generic_paths(dev):
    """Round 4: in eval the first 2D layer reads the dense image (stemconv.hip) and the coarser SPP pools are pooled from the finest pool's
    cells.  Both only change summation order: the full PSMNet on Config B's crops with the shortcuts on (default; held to the reference
    goldens by the test above) and off (layout conversion + generic stride-2 kernel, one pooling pass per branch) agrees inside the
    golden tolerance (the synthetic network's sharp soft-argmin amplifies fp32 rounding of the features)."""
    from disprcnn_amd import engine as E
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    m = PSMNet(48, -48)
    m.load_state_dict(state_for("B"), strict=True)
    m = m.to(dev).eval()
    left, right = synth.synth_images(4, 224, 224, tag="shortcuts")
    assert E.STEM_DIRECT["enabled"] and E.SPP_NESTED["enabled"]
    with torch.no_grad():
        on = m((left.to(dev), right.to(dev))).cpu()
        E.STEM_DIRECT["enabled"] = E.SPP_NESTED["enabled"] = False
        try:
            off = m((left.to(dev), right.to(dev))).cpu()
        finally:
            E.STEM_DIRECT["enabled"] = E.SPP_NESTED["enabled"] = True
    err = (on - off).abs()
    print("2D-CNN shortcuts on vs off: mean/max |diff| px", err.mean().item(), err.max().item())
    assert err.mean().item() < 5e-4 and err.max().item() < 1e-2 and torch.isfinite(on).all()      # measured 2.3e-4 / 4.6e-3 (goldens: 1e-3 / 2e-2)
