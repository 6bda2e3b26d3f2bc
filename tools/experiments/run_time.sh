#!/bin/bash
# development: timing only of Winograd kernel variants (ablations give wrong results by design)
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L=$PWD/tools/exp_libs/lib_$v.so; fi
  DRC_LIB=$L N=256 ${EXTRA_ENV} timeout 120 python tools/experiments/exp_conv.py 2>&1 | grep -E "wino3d|rror" | head -3
done
