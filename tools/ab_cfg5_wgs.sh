#!/bin/bash
# cfg5 (D = 36, 48-row real class): two workgroups per CU (256 registers per wave, ~60 doubles per lane spilled) against one
# (512 registers, no spills) -- the default was chosen in round 3, before the 32 + 4 row split changed the tile deal.
#   build here:  hipcc ... -DC3P_MIDD_PART=2 -DC3P_MDR_BIG_WGS=1 -c c3p_midd.hip, linked into c3_amd/libc3prop_wg1.so
#   run on the GPU box: bash tools/ab_cfg5_wgs.sh
cd "$(dirname "$0")/.."
O=gpurun_out/ab_cfg5_wgs
mkdir -p $O
for v in "" wg1; do
  lib=""; [ -n "$v" ] && lib=$PWD/c3_amd/libc3prop_$v.so
  for b in 256 1024; do
    C3P_LIB=$lib python -c "
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from c3_amd import _lib
if os.environ.get('C3P_LIB'): _lib.LIB_PATH = os.environ['C3P_LIB']
sys.argv = ['bench.py', '--config', '5', '--batch', '$b', '--steps', '6', '--warmup', '2', '--check', '--no-cpu-baseline', '--no-e2e']
runpy.run_path('bench.py', run_name='__main__')" > $O/bench_${v:-regular}_B$b.json 2> $O/bench_${v:-regular}_B$b.err
    python -c "
import json
d = json.loads(open('$O/bench_${v:-regular}_B$b.json').read().strip().splitlines()[-1])
print('${v:-regular}', $b, d['value'], d['ms_per_step'], d.get('max_fro_err_vs_oracle'))"
  done
done
