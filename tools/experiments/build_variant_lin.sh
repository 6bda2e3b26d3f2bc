#!/bin/bash
# build_variant_lin.sh NAME [flags...] : tools/exp_libs/lib_NAME.so with csrc/linear.hip rebuilt with the given flags (development tool)
set -e
name=$1; shift 1
csrc=/root/repo/disprcnn_amd/csrc
mkdir -p /root/repo/tools/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$csrc "$@" -c $csrc/linear.hip -o /tmp/variant_$name.o
objs=$(ls $csrc/*.o | grep -v "/linear.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp_libs/lib_$name.so $objs /tmp/variant_$name.o
