#!/bin/bash
for v in "$@"; do
  if [ "$v" = default ]; then L=""; else L=$PWD/tools/exp_libs/lib_$v.so; fi
  DRC_LIB=$L N=256 ALL=1 timeout 120 python tools/experiments/exp_conv.py 2>&1 | grep -E "dc=True|rror" | head -3
done
