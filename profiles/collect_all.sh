#!/bin/bash
# ALL rocprofv3 evidence of a round from ONE commit: for each workload a --kernel-trace --stats pass and the PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ busy + waits, LDS and cache counters for the headline), each pass in its own run, never combined with
# sys/runtime traces.  Run from the repo root of the authoring container:
#     git rev-parse HEAD > profiles/.head && gpurun --timeout 2400 -- 'bash profiles/collect_all.sh r4'
# (the GPU box has no .git: profiles/.head carries the commit the snapshot was taken from; every summary quotes it).  Raw CSVs stay on the
# box (/tmp/prof_<tag>_<workload>/: tens of MB per PMC pass, gpurun returns at most 64 MiB); profiles/summarize.py writes the summaries, which
# are copied to gpurun_out/profiles_<tag>/ -- copy them from there into profiles/ and commit; profiles/check_profiles.py fails if any
# <tag>_*_pmc.md came out without rows.
set -u
TAG=${1:-r5}
ONLY=${2:-}      # optional: space-separated list of workloads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PROF_COMMIT=$(cat profiles/.head 2>/dev/null || echo unknown)
PMC_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PMC_LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_LDS"
PMC_CACHE="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
LIGHT=${LIGHT:-}     # workloads that get the kernel-trace pass only (no counters): a partial re-collection under a tight GPU budget
NOFULL=${NOFULL:-}   # workloads whose LDS / cache passes are skipped
run() {   # name, suffix of the summary files, description, full (1: also LDS + cache passes), command...
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $1 "* ]]; then return; fi
  local name=$1 sfx=$2 desc=$3 full=$4; shift 4
  if [[ " $NOFULL " == *" $name "* ]]; then full=0; fi
  local OUT=/tmp/prof_${TAG}_$name
  rm -rf "$OUT"; mkdir -p "$OUT"
  local t0=$SECONDS
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- "$@" > "$OUT/trace.log" 2>&1
  if [[ " $LIGHT " != *" $name "* ]]; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT" -o fetch -- "$@" > "$OUT/fetch.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT" -o write -- "$@" > "$OUT/write.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $PMC_SQ -d "$OUT" -o sq -- "$@" > "$OUT/sq.log" 2>&1
  fi
  if [ "$full" = 1 ]; then
    timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $PMC_LDS -d "$OUT" -o lds -- "$@" > "$OUT/lds.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $PMC_CACHE -d "$OUT" -o cache -- "$@" > "$OUT/cache.log" 2>&1
  fi
  grep -h '"metric"' "$OUT"/trace.log | tail -1 > "$OUT/bench_line.json"; [ -s "$OUT/bench_line.json" ] || rm -f "$OUT/bench_line.json"
  PROF_CMD="$desc" python profiles/summarize.py "$OUT" "${TAG}${sfx}"
  echo "[$name] $((SECONDS - t0)) s"
}
run headline "" "python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --sustain-seconds 0  (Config A headline, 1024 ROI pairs per step; plus the instrumented repeat of the same 10 steps)" 1 python bench.py --steps 10 --warmup 3 --no-cpu --no-extra --sustain-seconds 0
run configB _configB "WHAT=psm python tools/prof_pair.py  (Config B: full PSMNet on 16 ROI crops 224x224, D=96; 2 warm-up + 5 timed passes)" 1 env WHAT=psm python tools/prof_pair.py
run pair_backbone _pair_backbone "WHAT=bb python tools/prof_pair.py  (R-50-FPN trunk on one stereo pair 2x3x375x1242 = 250.3 GFLOP; 2 warm-up + 5 timed passes)" 0 env WHAT=bb python tools/prof_pair.py
run stage2d _stage2d "python tools/prof_2d.py  (2D stage: DispRCNN = R-50-FPN trunk + Stereo RPN + stereo box head + mask head on one 2x3x375x1242 pair, synthetic weights; 2 warm-up + 5 timed passes)" 0 python tools/prof_2d.py
run stress16 _stress16 "WHAT=psm16 python tools/prof_pair.py  (configs[3]: 64 ROI crops 224x224, D=96, fp16-storage regressor, fp32 2D CNN)" 0 env WHAT=psm16 python tools/prof_pair.py
run train _train "N=64 python tools/prof_train.py  (Config A train step from the feature boundary, 64 ROI pairs: fwd + PSMLoss + bwd; 2 + 3 steps, then 3 forward-only passes)" 0 env N=64 python tools/prof_train.py
run trainB _trainB "N=8 CFG_B=1 python tools/prof_train.py  (Config B train step, full PSMNet on 8 crops 224x224, D=96: fwd + PSMLoss + bwd; 2 + 3 steps)" 0 env N=8 CFG_B=1 python tools/prof_train.py
python profiles/check_profiles.py "$TAG"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_*.md profiles/${TAG}_*.json profiles/${TAG}*.md gpurun_out/profiles_$TAG/ 2>/dev/null
for w in headline configB pair_backbone stage2d stress16 train trainB; do tail -3 /tmp/prof_${TAG}_$w/trace.log > gpurun_out/profiles_$TAG/$w.trace_tail.log 2>/dev/null; done
ls gpurun_out/profiles_$TAG | head -40
