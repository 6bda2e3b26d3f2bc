#!/usr/bin/env python3
"""Generate golden fixtures from the IMPORTED REFERENCE (run once, in the authoring container).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Needs /root/reference (read-only).  Nothing here runs on the GPU box: the outputs
(*.npz in this directory) are committed, the reference code is not.  Inputs and
weights come from the closed-form generator ``disprcnn_amd.utils.synth`` so the
tests can rebuild them bit-identically anywhere; only BN running statistics
(calibrated by train-mode passes of the reference) and outputs are stored.

Reference entry points exercised:
  PSMNet.forward            disprcnn/modeling/psmnet/stackhourglass.py:106-174
  feature_extraction        disprcnn/modeling/psmnet/submodule.py:60-139
  PSMLoss                   disprcnn/utils/loss_utils.py:4-32
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from disprcnn.modeling.psmnet.stackhourglass import PSMNet  # noqa: E402  (the reference)
from disprcnn.utils.loss_utils import PSMLoss  # noqa: E402

from disprcnn_amd.utils import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


class FeatStub(torch.nn.Module):
    """Replaces feature_extraction for Config A: returns preset features, L then R."""

    def __init__(self, fl, fr):
        super().__init__()
        self.q = [fl, fr]
        self.i = 0

    def forward(self, x):
        out = self.q[self.i % 2]
        self.i += 1
        return out


def sample_idx(numel, k=256, key="s"):
    u = synth.hash_uniform(f"sample:{key}:{numel}", (k,), 0.0, 1.0).double()
    return (u * numel).long().clamp(max=numel - 1).numpy()


def sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def capture(model):
    """Hooks that record the intermediates of stackhourglass.py:115-144."""
    store = {}
    hs = []
    hs.append(model.dres0.register_forward_pre_hook(lambda m, i: store.__setitem__("cost", i[0].detach().clone())))
    def hg_hook(name):
        def f(m, i, o):
            store[name] = [t.detach().clone() for t in o]
        return f

    def cl_hook(name):
        def f(m, i, o):
            store[name + "_in"] = i[0].detach().clone()
            store[name] = o.detach().clone()
        return f

    for name in ("dres2", "dres3", "dres4"):
        hs.append(getattr(model, name).register_forward_hook(hg_hook(name)))
    for name in ("classif1", "classif2", "classif3"):
        hs.append(getattr(model, name).register_forward_hook(cl_hook(name)))
    return store, hs


def calibrate(model, batches):
    """BN running stats := cumulative average of batch stats over train-mode passes."""
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.reset_running_stats()
            m.momentum = None
    model.train()
    with torch.no_grad():
        for b in batches:
            model(b)
    model.eval()
    return {k: v.numpy().copy() for k, v in model.state_dict().items()
            if k.endswith("running_mean") or k.endswith("running_var")}


def record_intermediates(out, store, tag):
    cost = store["cost"]
    out[f"{tag}_cost_sha"] = np.array(sha(cost))
    inter = {
        "cost": cost,
        "out1": store["classif1_in"], "out2": store["classif2_in"], "out3": store["classif3_in"],
        "pre1": store["dres2"][1], "post1": store["dres2"][2], "post2": store["dres3"][2],
        "hg3_raw": store["dres4"][0],
        "cost1": store["classif1"],
        "cost2": store["classif2"] + store["classif1"],
        "cost3": store["classif3"] + store["classif2"] + store["classif1"],
    }
    for k, t in inter.items():
        flat = t.reshape(-1)
        idx = sample_idx(flat.numel(), key=k)
        out[f"{tag}_{k}_idx"] = idx
        out[f"{tag}_{k}_val"] = flat[idx].numpy()
        out[f"{tag}_{k}_sum"] = np.array(flat.double().sum().item())
        out[f"{tag}_{k}_abssum"] = np.array(flat.double().abs().sum().item())


def main():
    template = PSMNet(48, -48).state_dict()
    out = {}

    # ------------------------------------------------------------ cost-volume-only cases (bit exact)
    cv = {}
    for (mx, mn, shp) in [(48, -48, (2, 4, 6, 20)), (8, 0, (1, 3, 5, 12)), (8, -8, (1, 3, 5, 12)),
                          (0, -8, (2, 2, 3, 9)), (12, -4, (1, 2, 4, 10)), (48, 0, (1, 32, 28, 28)),
                          (24, -24, (1, 32, 28, 28)), (48, -48, (1, 32, 56, 56))]:
        fl, fr = synth.synth_features(*shp, tag=f"cv{mx}_{mn}")
        m = PSMNet(mx, mn).eval()
        m.feature_extraction = FeatStub(fl, fr)
        store, hs = capture(m)
        # run only as far as dres0's pre-hook: abort there to avoid needing valid 3D shapes
        class Stop(Exception):
            pass
        def stop(mod, i):
            store["cost"] = i[0].detach().clone()
            raise Stop()
        m.dres0.register_forward_pre_hook(stop)
        try:
            with torch.no_grad():
                m((torch.zeros(shp[0], 3, 4 * shp[2], 4 * shp[3]),) * 2)
        except Stop:
            pass
        c = store["cost"]
        key = f"cv_{mx}_{mn}_{'x'.join(map(str, shp))}"
        cv[key + "_sha"] = np.array(sha(c))
        cv[key + "_shape"] = np.array(c.shape)
        if c.numel() <= 20000:
            cv[key + "_full"] = c.numpy()
    np.savez_compressed(os.path.join(HERE, "cost_volume.npz"), **cv)
    print("cost volume cases:", len(cv))

    # ------------------------------------------------------------ Config A: features -> disparity
    for tempered in (False, True):
        sd = synth.synth_state_dict(template, tempered=tempered)
        tag = "At" if tempered else "A"
        model = PSMNet(48, 0)
        model.load_state_dict(sd, strict=True)
        batches = []
        for p in range(4):
            fl, fr = synth.synth_features(2, 32, 28, 28, tag=f"calA{p}")
            batches.append((fl, fr))

        class CalWrap(torch.nn.Module):
            def __init__(s, m):
                super().__init__()
                s.m = m
            def forward(s, b):
                s.m.feature_extraction = FeatStub(*b)
                return s.m((torch.zeros(2, 3, 112, 112),) * 2)
        # calibrate 3D BN layers through the reference's own forward
        wrap = CalWrap(model)
        for m_ in model.modules():
            if isinstance(m_, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m_.reset_running_stats(); m_.momentum = None
        model.train()
        with torch.no_grad():
            for b in batches:
                wrap(b)
        model.eval()
        bn = {k: v.numpy().copy() for k, v in model.state_dict().items()
              if (k.endswith("running_mean") or k.endswith("running_var")) and not k.startswith("feature_extraction")}
        if not tempered:
            np.savez_compressed(os.path.join(HERE, "bn_stats_A.npz"), **bn)
        else:
            np.savez_compressed(os.path.join(HERE, "bn_stats_At.npz"), **bn)

        fl, fr = synth.synth_features(2, 32, 28, 28, tag="caseA")
        model.feature_extraction = FeatStub(fl, fr)
        store, hs = capture(model)
        with torch.no_grad():
            pred = model((torch.zeros(2, 3, 112, 112),) * 2)
        out[f"{tag}_pred"] = pred.numpy()
        record_intermediates(out, store, tag)
        print(tag, "pred range", pred.min().item(), pred.max().item(), pred.std().item())
        for h in hs:
            h.remove()

        if not tempered:
            # A2: same weights/stats, maxdisp 24 / mindisp -24 (negative, zero and positive shifts)
            m2 = PSMNet(24, -24)
            m2.load_state_dict(model.state_dict(), strict=False)  # feature_extraction is a stub here
            m2.eval()
            m2.feature_extraction = FeatStub(fl, fr)
            store, hs = capture(m2)
            with torch.no_grad():
                pred = m2((torch.zeros(2, 3, 112, 112),) * 2)
            out["A2_pred"] = pred.numpy()
            record_intermediates(out, store, "A2")
            print("A2 pred range", pred.min().item(), pred.max().item())
            # (the reference cannot run in fp64: stackhourglass.py:117 forces .float(); the fp64
            #  noise floor is measured with the oracle restatement in tests/test_oracle_golden.py)

    # ------------------------------------------------------------ Config B: images -> disparity (full PSMNet)
    sd = synth.synth_state_dict(template)
    model = PSMNet(48, -48)
    model.load_state_dict(sd, strict=True)
    batches = [synth.synth_images(2, 224, 224, tag=f"calB{p}") for p in range(3)]
    bn = calibrate(model, batches)
    np.savez_compressed(os.path.join(HERE, "bn_stats_B.npz"), **bn)
    left, right = synth.synth_images(2, 224, 224, tag="caseB")
    store, hs = capture(model)
    feats = {}
    fh = model.feature_extraction.register_forward_hook(lambda m, i, o: feats.setdefault(len(feats), o.detach().clone()))
    with torch.no_grad():
        pred = model((left, right))
    fh.remove()
    out["B_pred"] = pred.numpy()
    record_intermediates(out, store, "B")
    for i, nm in ((0, "featL"), (1, "featR")):
        flat = feats[i].reshape(-1)
        idx = sample_idx(flat.numel(), key=nm)
        out[f"B_{nm}_idx"], out[f"B_{nm}_val"] = idx, flat[idx].numpy()
        out[f"B_{nm}_abssum"] = np.array(flat.double().abs().sum().item())
    print("B pred range", pred.min().item(), pred.max().item(), pred.std().item())
    for h in hs:
        h.remove()

    # ------------------------------------------------------------ B-train: 3 heads + PSMLoss + grads
    model.train()
    for m_ in model.modules():
        if isinstance(m_, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m_.momentum = 0.1
    left, right = synth.synth_images(2, 224, 224, tag="caseBtrain")
    left.requires_grad_(True)
    target = synth.hash_uniform("tgt", (2, 224, 224), -48.0, 48.0)
    mask = (synth.hash_uniform("mask", (2, 224, 224), 0.0, 1.0) > 0.5).to(torch.uint8)
    preds = model({"left": left, "right": right})
    loss = PSMLoss()(preds, {"mask": mask, "disparity": target})
    loss.backward()
    out["Bt_loss"] = np.array(loss.item())
    for i, p in enumerate(preds):
        out[f"Bt_pred{i + 1}_s4"] = p.detach()[:, ::4, ::4].numpy()
    out["Bt_gleft_s4"] = left.grad[:, :, ::4, ::4].numpy()
    for name in ("dres0.0.0.weight", "dres2.conv5.0.weight", "classif3.2.weight", "dres4.conv6.1.weight",
                 "feature_extraction.lastconv.2.weight"):
        g = dict(model.named_parameters())[name].grad.reshape(-1)
        idx = sample_idx(g.numel(), key="g" + name)
        out[f"Bt_g:{name}_idx"], out[f"Bt_g:{name}_val"] = idx, g[idx].numpy()
        out[f"Bt_g:{name}_abssum"] = np.array(g.double().abs().sum().item())
    # eval-mode loss (EPE) on the eval prediction
    model.eval()
    with torch.no_grad():
        pe = model((left.detach(), right))
        out["Bt_epe_eval"] = np.array(float(PSMLoss()(pe, {"mask": mask, "disparity": target})))
    print("Btrain loss", loss.item(), "eval epe", out["Bt_epe_eval"])

    np.savez_compressed(os.path.join(HERE, "psmnet_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "psmnet_golden.npz"))


if __name__ == "__main__":
    main()
