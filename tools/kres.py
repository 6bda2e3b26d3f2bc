"""Per-kernel register / spill / scratch summary of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
    python tools/kres.py disprcnn_amd/csrc/convs16d.hip [more.hip ...]
"""
import os
import re
import subprocess
import sys

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-Wno-pass-failed",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def main():
    for src in sys.argv[1:]:
        out = subprocess.run([HIPCC] + FLAGS + [src], capture_output=True, text=True).stderr
        cur = None
        rows = []
        for line in out.splitlines():
            m = re.search(r"remark: +(?:Function )?Name: (\S+)", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = {"name": re.sub(r"\(anonymous namespace\)::|\(drc_\w+\)", "", name)}
                rows.append(cur)
                continue
            m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        print(src)
        for r in rows:
            print("  %-62s VGPR %3d AGPR %3d  spill V %d S %d  scratch %d B  occ %d" % (
                r["name"][:62], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                r.get("ScratchSize", -1), r.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
