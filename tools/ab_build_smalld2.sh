#!/bin/bash
# A/B build of part 2 of the small-D file (the real-Hamiltonian backward sweep): tools/ab_build_smalld2.sh <name> <flags...> -> c3_amd/libc3prop_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p c3_amd/csrc/build_ab
alt=c3_amd/csrc/build_ab/c3p_smalld_gradreal_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DC3P_SMALLD_PART=2 -mllvm -amdgpu-mfma-vgpr-form "$@" -c c3_amd/csrc/c3p_smalld.hip -o $alt
objs=$(ls c3_amd/csrc/build/*.o | grep -v "c3p_smalld_gradreal.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o c3_amd/libc3prop_$name.so $objs $alt
echo c3_amd/libc3prop_$name.so
