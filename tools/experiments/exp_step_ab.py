"""Same-box A/B of the headline step (Config A, 1024 ROI pairs): engine.CV_WIDE on / off, interleaved rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from disprcnn_amd import engine as E
from disprcnn_amd.utils import synth
dev = torch.device("cuda:0")
fl, fr = synth.synth_features(1024, 32, 28, 28, tag="bench0")
fl, fr = fl.to(dev), fr.to(dev)
models = {}
for on in (True, False):
    E.CV_WIDE["enabled"] = on
    m, _ = bench.build_model(dev, 48, 0, "A")
    with torch.no_grad():
        for _ in range(3):
            m.forward_from_features(fl, fr, (112, 112))
    models[on] = m
    print("wide" if on else "one tile", m._rt._ws[("3ds16", 1024, 12, 28, 28)]["p"]["dres0.0"].kname)
res = {True: [], False: []}
with torch.no_grad():
    for rnd in range(4):
        for on in (True, False):
            m = models[on]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                m.forward_from_features(fl, fr, (112, 112))
            torch.cuda.synchronize()
            res[on].append((time.perf_counter() - t0) / 10 * 1e3)
for on in (True, False):
    print("CV_WIDE", on, "ms per step:", " ".join(f"{t:.3f}" for t in res[on]), f"-> median {sorted(res[on])[len(res[on]) // 2]:.3f}")
a, b = torch.no_grad(), None
with torch.no_grad():
    print("identical outputs:", torch.equal(models[True].forward_from_features(fl, fr, (112, 112)), models[False].forward_from_features(fl, fr, (112, 112))))
