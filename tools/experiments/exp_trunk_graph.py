"""Round 6 experiment that lost: the ResNet-50-FPN trunk replayed from a HIP graph (BackboneRuntime._forward_graphed, removed again) against the eager
launches, same box, interleaved rounds of ten calls on the KITTI pair: replay 3.08-3.10 ms, eager 3.02-3.03 ms -- the trunk is paced by the GPU
(2.8 ms of kernels), not by the host; the static input copy and the clones of the five output maps cost more than the launches saved.  This
script needs that code path (git history: "trunk graph") to show the `auto` column."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace as NS
from disprcnn_amd.modeling.backbone import build_backbone
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
def mk(mode):
    bb = build_backbone(NS(MODEL=NS(BACKBONE=NS(CONV_BODY="R-50-FPN"), RESNETS=NS(BACKBONE_OUT_CHANNELS=256, RES2_OUT_CHANNELS=256))))
    bsd = synth.synth_backbone_state(bb.state_dict())
    synth.load_bn_stats(bsd, "/root/repo/tests/golden/bn_stats_backbone.npz")
    bb.load_state_dict(bsd); bb = bb.to(dev).eval(); bb.graph_eval = mode
    return bb
pair = synth.hash_uniform("benchpair", (2, 3, 375, 1242), 0.0, 1.0).to(dev)
ms = {m: mk(m) for m in ("auto", False)}
res = {m: [] for m in ms}
with torch.no_grad():
    for m, bb in ms.items():
        for _ in range(3): bb(pair)
    for rnd in range(4):
        for m, bb in ms.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): bb(pair)
            torch.cuda.synchronize(); res[m].append((time.perf_counter() - t0) / 10 * 1e3)
for m in ms: print("graph_eval", m, " ".join(f"{t:.3f}" for t in res[m]))
