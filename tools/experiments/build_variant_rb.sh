#!/bin/bash
# build_variant_rb.sh NAME [flags...] : link tools/exp_libs/lib_NAME.so with csrc/wino3d_rb.hip rebuilt with the given flags (development tool)
set -e
name=$1; shift 1
csrc=/root/repo/disprcnn_amd/csrc
src=${RB_SRC:-$csrc/wino3d_rb.hip}
mkdir -p /root/repo/tools/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed -Wno-uninitialized -I$csrc "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls $csrc/*.o | grep -v "/wino3d_rb.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp_libs/lib_$name.so $objs /tmp/variant_$name.o
