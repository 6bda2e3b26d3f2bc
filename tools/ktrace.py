"""Development tool: per-launch durations (us) of kernels whose name contains a substring, from a rocprofv3 kernel_trace.csv."""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
per = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
    per.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    print(k, len(v), "launches, median us", sorted(v)[len(v) // 2], "total", round(sum(v), 1))
