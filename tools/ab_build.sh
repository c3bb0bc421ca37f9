#!/bin/bash
# A/B builds: compile ONE source with extra flags and link it with the other objects of the regular build into
# c3_amd/libc3prop_<name>.so (the regular library is left alone); select it with C3P_LIB=<path> in tools/*.py.
#   tools/ab_build.sh timing c3p_regr.hip -DC3P_REGR_TIMING
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
obj=c3_amd/csrc/build/${src%.hip}.o
mkdir -p c3_amd/csrc/build_ab
alt=c3_amd/csrc/build_ab/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c c3_amd/csrc/$src -o $alt
objs=$(ls c3_amd/csrc/build/*.o | grep -v "^$obj$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o c3_amd/libc3prop_$name.so $objs $alt
echo c3_amd/libc3prop_$name.so
