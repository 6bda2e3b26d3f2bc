#!/bin/bash
# build_variant_any.sh NAME FILE.hip [flags...] : tools/exp_libs/lib_NAME.so with csrc/FILE.hip rebuilt with the given flags (development tool)
set -e
name=$1; file=$2; shift 2
csrc=/root/repo/disprcnn_amd/csrc
mkdir -p /root/repo/tools/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$csrc "$@" -c $csrc/$file -o /tmp/variant_$name.o
objs=$(ls $csrc/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp_libs/lib_$name.so $objs /tmp/variant_$name.o
