#!/bin/bash
# pmc_sq.sh "<command>" <kernel-name-substring>: ONE counter pass (SQ busy / wait counters + GRBM_GUI_ACTIVE) and a per-kernel summary:
# MFMA busy %, effective clock, waves/SIMD.  (Development tool; run via gpurun.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
export PMC_FILTER="$2"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o sq -- $1 > $OUT/sq.log 2>&1
python - <<'PY'
import csv, glob, collections, os
flt = os.environ.get("PMC_FILTER", "")
for f in sorted(glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k.split("::")[-1][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["__dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        gui = m["GRBM_GUI_ACTIVE"] / 8.0
        simd = 1024 * gui
        print(f"{k}: launches {len(d['GRBM_GUI_ACTIVE'])} dur {m['__dur'] / 1e3:.1f} us clock {gui / m['__dur']:.2f} GHz MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:.1f}% "
              f"waves/SIMD {4 * m['SQ_WAVE_CYCLES'] / simd:.2f} wait_any {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.1f}% wait_inst {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.1f}% active {100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.1f}%")
PY
