#!/bin/bash
# Print the headline line's value, step time and per-kernel table (tools helper for gpurun).
python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu --no-extra "$@" 2>&1 | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["extra"]["kernels"].items(): print("  ", k, v)
'
