#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r1'
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2-4: PMC counters, each in its own run with
# --kernel-trace only (never combined with sys/runtime traces).  Raw CSVs land in gpurun_out/prof_<tag>/ (scratch);
# profiles/summarize.py turns them into the small summaries committed under profiles/.
set -u
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT" -o fetch -- $BENCH > "$OUT/fetch.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT" -o write -- $BENCH > "$OUT/write.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT" -o sq -- $BENCH > "$OUT/sq.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_LDS -d "$OUT" -o lds -- $BENCH > "$OUT/lds.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o cache -- $BENCH > "$OUT/cache.log" 2>&1
grep -h '"metric"' "$OUT"/trace.log | tail -1 > "$OUT/bench_line.json"
python profiles/summarize.py "$OUT" "$TAG"
ls -la "$OUT" | head -30
