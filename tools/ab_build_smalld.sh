#!/bin/bash
# A/B build of part 1 of the small-D file: tools/ab_build_smalld.sh <name> <flags...> -> c3_amd/libc3prop_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p c3_amd/csrc/build_ab
alt=c3_amd/csrc/build_ab/c3p_smalld_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DC3P_SMALLD_PART=1 "$@" -c c3_amd/csrc/c3p_smalld.hip -o $alt
objs=$(ls c3_amd/csrc/build/*.o | grep -v "c3p_smalld.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o c3_amd/libc3prop_$name.so $objs $alt
echo c3_amd/libc3prop_$name.so
