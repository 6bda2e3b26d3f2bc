#!/usr/bin/env python3
"""Fail if a round's profile set has holes: every profiles/<tag>*_pmc.md must have counter rows, every <tag>*_traffic.json entries, all
summaries of the set must quote the same commit (profiles/collect_all.sh).  Also run by tests/test_host_logic.py for the newest set."""
import glob
import json
import os
import re
import sys


def check(tag, here=None):
    here = here or os.path.dirname(os.path.abspath(__file__))
    problems, commits = [], set()
    pmcs = sorted(glob.glob(os.path.join(here, f"{tag}_pmc.md")) + glob.glob(os.path.join(here, f"{tag}_*_pmc.md")))
    if not pmcs:
        problems.append(f"no {tag}*_pmc.md at all")
    for p in pmcs:
        txt = open(p).read()
        rows = [l for l in txt.splitlines() if l.startswith("| `")]
        if not rows:
            problems.append(f"{os.path.basename(p)}: no counter rows")
        m = re.search(r"^Commit: `?([0-9a-f]{7,40}|unknown)", txt, re.M)
        commits.add(m.group(1) if m else "missing")
        tj = p.replace("_pmc.md", "_traffic.json")
        try:
            if not json.load(open(tj)):
                problems.append(f"{os.path.basename(tj)}: empty")
        except (OSError, ValueError):
            problems.append(f"{os.path.basename(tj)}: missing or unreadable")
        ks = p.replace("_pmc.md", "_kernel_stats.md")
        if not os.path.exists(ks) or not [l for l in open(ks).read().splitlines() if l.startswith("| `")]:
            problems.append(f"{os.path.basename(ks)}: missing or without kernel rows")
    if len(commits) > 1 or commits & {"missing", "unknown"}:
        problems.append(f"the set does not quote ONE commit: {sorted(commits)}")
    return problems


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
    bad = check(tag)
    for b in bad:
        print("PROFILE CHECK:", b)
    print(f"profiles/{tag}*: {'OK' if not bad else 'INCOMPLETE'}")
    sys.exit(1 if bad else 0)
