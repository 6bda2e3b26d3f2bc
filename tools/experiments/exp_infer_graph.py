"""Development tool: eager vs HIP-graph replay of the Config-A inference step from ROI features at small batches (is it launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from disprcnn_amd.utils import synth
from disprcnn_amd.utils.graph import GraphedStep
dev = torch.device("cuda:0")
m, _ = bench.build_model(dev, 48, 0, "A")
for n in (4, 16, 64, 256):
    fl, fr = synth.synth_features(n, 32, 28, 28, tag="g")
    fl, fr = fl.to(dev), fr.to(dev)
    f = lambda: m.forward_from_features(fl, fr, (112, 112))
    def t(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    with torch.no_grad():
        te = t(f)
        gs = GraphedStep(f, warmup=2)
        ref = f().clone(); out = gs().clone()
        tg = t(gs)
    print(f"N={n}: eager {te:.3f} ms ({n / te * 1e3:.0f} ROI/s), graph {tg:.3f} ms ({n / tg * 1e3:.0f} ROI/s), equal {torch.equal(ref, out)}")
