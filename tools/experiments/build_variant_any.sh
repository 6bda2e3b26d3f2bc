#!/bin/bash
# build_variant_any.sh NAME SRC.hip [flags...] : link tools/exp_libs/lib_NAME.so with csrc/SRC.hip rebuilt with the given flags (development tool)
set -e
name=$1; src=$2; shift 2
csrc=/root/repo/disprcnn_amd/csrc
mkdir -p /root/repo/tools/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed -Wno-uninitialized -I$csrc "$@" -c $csrc/$src -o /tmp/variant_$name.o
objs=$(ls $csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp_libs/lib_$name.so $objs /tmp/variant_$name.o
