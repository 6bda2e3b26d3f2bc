#!/bin/bash
# run_rb_variants.sh [names...]: time the default lib and each tools/exp_libs/lib_NAME.so (development tool; run on the GPU box)
RB=0 timeout 100 python tools/experiments/exp_rb_time.py 2>&1 | tail -1 | sed 's/^default/wino3d (old)/'
timeout 100 python tools/experiments/exp_rb_time.py 2>&1 | tail -1
for v in "$@"; do DRC_LIB=$PWD/tools/exp_libs/lib_$v.so timeout 100 python tools/experiments/exp_rb_time.py 2>&1 | tail -1; done
