"""Development tool: print a rocprofv3 kernel_stats.csv (first match under a directory) as us per step."""
import csv, glob, sys
d, div, top = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", tot / div / 1e6, "launches per step", sum(int(r["Calls"]) for r in rows) / div)
for r in rows[:top]:
    print(r["Name"][:110], int(r["Calls"]) / div, round(float(r["TotalDurationNs"]) / div / 1e3, 1), "us/step", r["Percentage"])
