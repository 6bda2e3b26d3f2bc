"""Development tool: torch.profiler view of the 2D stage: which python lines issue memcpys / syncing ops per pair."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import importlib.util
spec = importlib.util.spec_from_file_location("p2d", os.path.join(os.path.dirname(__file__), "prof_2d.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        mod.run()
    torch.cuda.synchronize()
c = collections.Counter(); t = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense", "aten::nonzero", "aten::fill_", "aten::zero_", "aten::cat",
                  "aten::index", "aten::index_select", "aten::sort", "aten::zeros", "aten::empty", "aten::full", "aten::arange", "aten::clone", "aten::contiguous"):
        st = [s for s in (e.stack or []) if "disprcnn_amd" in s]
        key = (e.name, st[0].split("disprcnn_amd/")[-1] if st else "?")
        c[key] += 1; t[key] += e.cuda_time_total if hasattr(e, "cuda_time_total") else 0
for k, v in c.most_common(45):
    print(round(v / 3, 1), round(t[k] / 3, 1), k)
