#!/bin/bash
# build_variant2.sh NAME OBJ SRC.hip [flags...] : link tools/exp_libs/lib_NAME.so with SRC replacing csrc/OBJ.o (development tool)
set -e
name=$1; obj=$2; src=$3; shift 3
csrc=/root/repo/disprcnn_amd/csrc
mkdir -p /root/repo/tools/exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed -Wno-uninitialized "$@" -c $src -o /tmp/variant_$name.o
objs=$(ls $csrc/*.o | grep -v "/$obj.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/exp_libs/lib_$name.so $objs /tmp/variant_$name.o
