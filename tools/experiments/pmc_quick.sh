#!/bin/bash
# pmc_quick.sh "<command>" <kernel-substring>: fetch/write bytes, MFMA busy, wait mix, LDS conflicts of matching kernels (dev tool; gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
CMD="$1"; export PMC_FILTER="$2"
i=0
for set in "FETCH_SIZE WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT -o s$i -- $CMD > $OUT/s$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os, re
flt = os.environ.get("PMC_FILTER", "")
for f in sorted(glob.glob("gpurun_out/pmc_quick/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        m = re.search(r"(\w+<[^>]*>|\w+)\(", k.replace("(anonymous namespace)::", ""))
        k = m.group(1) if m else k[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, d in sorted(acc.items()):
        print(k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in d.items()})
PY
