#!/bin/bash
# A/B build of the mid-D file (three translation units): tools/ab_build_midd.sh <name> [parts] <flags...>
#   tools/ab_build_midd.sh nopw "2 3" -DC3P_MDR_NO_PINWHEEL   -> c3_amd/libc3prop_nopw.so
set -e
name=$1; parts=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p c3_amd/csrc/build_ab
declare -A sfx=([1]=chain [2]=real [3]=grad)
objs=$(ls c3_amd/csrc/build/*.o)
pids=()
for p in $parts; do
  alt=c3_amd/csrc/build_ab/c3p_midd_${sfx[$p]}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DC3P_MIDD_PART=$p "$@" -c c3_amd/csrc/c3p_midd.hip -o $alt &
  pids+=($!)
  objs=$(echo "$objs" | grep -v "c3p_midd_${sfx[$p]}.o")
  objs="$objs $alt"
done
for q in "${pids[@]}"; do wait $q; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o c3_amd/libc3prop_$name.so $objs
echo c3_amd/libc3prop_$name.so
