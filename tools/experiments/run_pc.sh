#!/bin/bash
# development: correctness + timing of Winograd kernel variants (tools/exp_libs/lib_NAME.so) on the GPU box
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = default ]; then L=""; else L=$PWD/tools/exp_libs/lib_$v.so; fi
  DRC_LIB=$L ONLY3D=1 timeout 120 python tools/check_wino.py 2>&1 | tail -4
  DRC_LIB=$L N=256 ALL=1 timeout 120 python tools/experiments/exp_conv.py 2>&1 | grep -E "wino3d|Error|error" 
done
