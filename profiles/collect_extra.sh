#!/bin/bash
# rocprofv3 evidence for the secondary workloads: Config B (full PSMNet, 16 ROI crops), the R-50-FPN trunk on a
# KITTI-sized stereo pair, one Config-A train step, the fp16-storage stress shape.  Run on the GPU box from the repo root:
#   gpurun --timeout 1500 -- 'bash profiles/collect_extra.sh r2'
# Per workload: --kernel-trace --stats, then PMC passes (FETCH_SIZE / WRITE_SIZE / SQ busy + waits), each in its own run.
set -u
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ONLY=${2:-}      # optional: collect just this workload (or a space-separated list)
TRACE_ONLY=${3:-} # optional: workloads (space-separated) for which only the --kernel-trace --stats pass is collected (PMC passes serialise kernels)
run() {   # name, description, command...
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $1 "* ]]; then return; fi
  local name=$1 desc=$2; shift 2
  local OUT=gpurun_out/prof_${TAG}_$name
  mkdir -p "$OUT"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- "$@" > "$OUT/trace.log" 2>&1
  if [[ " $TRACE_ONLY " != *" $name "* ]]; then
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT" -o fetch -- "$@" > "$OUT/fetch.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT" -o write -- "$@" > "$OUT/write.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT" -o sq -- "$@" > "$OUT/sq.log" 2>&1
  fi
  tail -2 "$OUT/trace.log"
  PROF_CMD="$desc" python profiles/summarize.py "$OUT" "${TAG}_$name"
}
run configB "WHAT=psm python tools/prof_pair.py  (Config B: full PSMNet on 16 ROI crops 224x224, D=96; 2 warm-up + 5 timed passes)" env WHAT=psm python tools/prof_pair.py
run pair_backbone "WHAT=bb python tools/prof_pair.py  (R-50-FPN trunk on one stereo pair 2x3x375x1242 = 250.3 GFLOP; 2 warm-up + 5 timed passes)" env WHAT=bb python tools/prof_pair.py
run train "N=64 python tools/prof_train.py  (Config A train step from the feature boundary, 64 ROI pairs: fwd + PSMLoss + bwd; 2 + 3 steps, then 3 forward-only passes)" env N=64 python tools/prof_train.py
run trainB "N=8 CFG_B=1 python tools/prof_train.py  (Config B train step, full PSMNet on 8 crops 224x224, D=96: fwd + PSMLoss + bwd; 2 + 3 steps)" env N=8 CFG_B=1 python tools/prof_train.py
run stage2d "python tools/prof_2d.py  (2D stage: DispRCNN = R-50-FPN trunk + Stereo RPN + stereo box head + mask head on one 2x3x375x1242 pair, synthetic weights; 2 warm-up + 5 timed passes)" python tools/prof_2d.py
run stress16 "WHAT=psm16 python tools/prof_pair.py  (configs[3]: 64 ROI crops 224x224, D=96, fp16-storage regressor)" env WHAT=psm16 python tools/prof_pair.py
run stress16f "WHAT=psm16f python tools/prof_pair.py  (configs[3] shape with the fp16-storage 2D feature CNN as well, PSMNet.feature_storage = f16: opt-in mode)" env WHAT=psm16f python tools/prof_pair.py
ls profiles/ | grep "$TAG"
