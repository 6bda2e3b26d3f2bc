#!/bin/bash
# time_kernels.sh PATTERN CMD...: per-kernel average (us) of the kernels matching PATTERN under rocprofv3 --kernel-trace (development tool; GPU box)
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o t -- "$@" > /tmp/tk.log 2>&1
grep -h " ms " /tmp/tk.log | tail -1
python - "$pat" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/tk/**/t_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(p in r["Name"] for p in sys.argv[1].split("|")):
        print("   %-46s calls %4s avg %9.1f us  %5s %%" % (r["Name"].replace("void (anonymous namespace)::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
