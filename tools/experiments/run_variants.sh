#!/bin/bash
# run_variants.sh [names...]: time the 32->32 layer at N=256 with the default lib and each tools/exp_libs/lib_NAME.so
N=256 timeout 60 python tools/experiments/exp_conv.py 2>&1 | tail -1 | awk '{print "default", $(NF-3), $(NF-2), $(NF-1), $NF}'
for v in "$@"; do DRC_LIB=$PWD/tools/exp_libs/lib_$v.so N=256 timeout 60 python tools/experiments/exp_conv.py 2>&1 | tail -1 | awk -v n=$v '{print n, $(NF-3), $(NF-2), $(NF-1), $NF}'; done
