#!/bin/bash
# rebuild only c3p_regd.o (+ optional extra flags) and relink libc3prop.so
set -e
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c c3_amd/csrc/c3p_regd.hip -o c3_amd/csrc/build/c3p_regd.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o c3_amd/libc3prop.so c3_amd/csrc/build/*.o
