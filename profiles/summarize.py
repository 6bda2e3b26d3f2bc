#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (profiles/collect.sh) into small committed files:
   profiles/<tag>_kernel_stats.md   per-kernel calls / avg / total (the --kernel-trace --stats view)
   profiles/<tag>_pmc.md            HBM traffic (FETCH_SIZE / WRITE_SIZE) and SQ counters per kernel
   profiles/<tag>_traffic.json      per-launch HBM bytes of the dominant kernel (read by bench.py for roofline.traffic)
FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced streams (MI355X_MICROARCH.md, HBM section), so the read side is given raw and doubled ("corrected")."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def load_counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            agg[k]["__dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg


def main(out, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    # ---- kernel stats from the trace
    tr = glob.glob(os.path.join(out, "trace_kernel_trace.csv"))
    rows = collections.defaultdict(list)
    if tr:
        for r in csv.DictReader(open(tr[0])):
            rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in rows.values()) or 1
    cmd = os.environ.get("PROF_CMD", "python bench.py --steps 10 --warmup 3 --no-cpu --no-extra` (plus the instrumented repeat of the same 10 steps)")
    commit = os.environ.get("PROF_COMMIT", "unknown")
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "", f"Commit: `{commit}`", "", f"Command: `{cmd}`" if not cmd.endswith(")") else f"Command: `{cmd}", ""]
    bl = os.path.join(out, "bench_line.json")
    if os.path.exists(bl):
        lines += ["bench.py line of this run:", "", "```", open(bl).read().strip(), "```", ""]
    lines += ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{k}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {sum(v) / 1e6:.2f} | {100 * sum(v) / tot:.1f} |")
    open(os.path.join(here, f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

    # ---- PMC
    fetch = load_counters(os.path.join(out, "fetch_counter_collection.csv"))
    write = load_counters(os.path.join(out, "write_counter_collection.csv"))
    sq = load_counters(os.path.join(out, "sq_counter_collection.csv"))
    lds = load_counters(os.path.join(out, "lds_counter_collection.csv"))
    lines = [f"# rocprofv3 PMC summary ({tag})", "", f"Commit: `{commit}`", "", "Averages per launch.  FETCH/WRITE in MB (raw KiB counters x 1024 / 1e6); "
             "`fetch x2` applies the gfx950 128-B-request correction.", "",
             "| kernel | launches | fetch MB | fetch x2 MB | write MB | avg us | MFMA busy % | clock GHz | waves/SIMD | wait_any % | wait_inst % | active % | LDS conflict % |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    for k in sorted(set(fetch) | set(sq), key=lambda k: -sum(sq.get(k, {}).get("__dur_ns", [0]))):
        f = mean(fetch.get(k, {}).get("FETCH_SIZE", [])) * 1024 / 1e6
        w = mean(write.get(k, {}).get("WRITE_SIZE", [])) * 1024 / 1e6
        s = sq.get(k, {})
        dur = mean(s.get("__dur_ns", []))
        gui = mean(s.get("GRBM_GUI_ACTIVE", [])) / 8.0            # summed over 8 XCDs
        clock = gui / dur if dur == dur and dur else float("nan")
        simd_cycles = 1024 * gui
        busy = 100 * mean(s.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / simd_cycles if simd_cycles else float("nan")
        wc = mean(s.get("SQ_WAVE_CYCLES", []))
        occ = 4 * wc / simd_cycles if simd_cycles else float("nan")
        pct = lambda c: 100 * mean(s.get(c, [])) / wc if wc else float("nan")
        l = lds.get(k, {})
        lconf = 100 * mean(l.get("SQ_LDS_BANK_CONFLICT", [])) / mean(l.get("SQ_LDS_IDX_ACTIVE", [])) if l.get("SQ_LDS_IDX_ACTIVE") and mean(l["SQ_LDS_IDX_ACTIVE"]) else float("nan")
        n = len(s.get("__dur_ns", [])) or len(fetch.get(k, {}).get("__dur_ns", []))
        lines.append(f"| `{k}` | {n} | {f:.1f} | {2 * f:.1f} | {w:.1f} | {dur / 1e3:.1f} | {busy:.1f} | {clock:.2f} | {occ:.2f} | "
                     f"{pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_ANY'):.1f} | {lconf:.1f} |")
        traffic[k] = {"fetch_bytes_raw": f * 1e6, "fetch_bytes_corrected": 2 * f * 1e6, "write_bytes": w * 1e6, "launches": n}
    cache = load_counters(os.path.join(out, "cache_counter_collection.csv"))
    if cache:
        lines += ["", "Vector L1 (TCP) and L2 (TCC) per launch: wave-level L1 accesses, L1->L2 read requests, L2 hit rate.", "",
                  "| kernel | L1 accesses (M) | L1->L2 read requests (M) | L2 hits (M) | L2 misses (M) | L2 hit % |", "|---|---|---|---|---|---|"]
        for k in sorted(cache, key=lambda k: -sum(cache[k].get("__dur_ns", [0])))[:8]:
            c = cache[k]
            a, rq, h, m = (mean(c.get(n, [])) / 1e6 for n in ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"))
            lines.append(f"| `{k}` | {a:.2f} | {rq:.2f} | {h:.2f} | {m:.2f} | {100 * h / (h + m) if h + m else float('nan'):.1f} |")
    open(os.path.join(here, f"{tag}_pmc.md"), "w").write("\n".join(lines) + "\n")
    try:        # what the numbers are valid for: bench.py compares the digest with the sources it runs and nulls roofline.traffic when they differ
        sys.path.insert(0, os.path.dirname(here))
        from disprcnn_amd.csrc.build import source_digest
        traffic["__meta__"] = {"commit": commit, "csrc_sha": source_digest(), "command": cmd}
    except Exception as ex:  # noqa: BLE001
        traffic["__meta__"] = {"commit": commit, "csrc_sha": None, "error": repr(ex)}
    json.dump(traffic, open(os.path.join(here, f"{tag}_traffic.json"), "w"), indent=1)
    print("wrote", f"profiles/{tag}_kernel_stats.md", f"profiles/{tag}_pmc.md", f"profiles/{tag}_traffic.json")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r1")
