"""Development tool: torch.profiler view of the 2D stage (which aten ops / memcpys the host side issues per pair)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("p2d", os.path.join(os.path.dirname(__file__), "prof_2d.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        mod.run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
evs = [e for e in prof.events() if "Memcpy" in e.name or "copy_" == e.name or e.name == "aten::copy_"]
import collections
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense", "aten::nonzero", "aten::fill_", "aten::zero_"):
        st = [s for s in (e.stack or []) if "disprcnn_amd" in s]
        c[(e.name, st[0] if st else "?")] += 1
for k, v in c.most_common(40):
    print(v / 3, k)
