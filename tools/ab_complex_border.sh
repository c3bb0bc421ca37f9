#!/bin/bash
# Complex-Hamiltonian small-D path (bench.py --complex, cfg2 operators with one complex control operator): what could a core +
# border form of the complex tile products buy?  Timing-only builds of the chain kernel (results are WRONG, no --check):
#   skipj : no matrix instructions for the last column block (the one that holds column 8 alone: 25 of 75 per product)
#   core  : 8 x 8 core alone (32 of 75 per product), i.e. every border product for free
# Build here (tools/ab_complex_border.sh build), run on the GPU box (tools/ab_complex_border.sh run).
cd "$(dirname "$0")/.."
case "$1" in
build)
  bash tools/ab_build.sh skipj c3p_smalld.hip -DC3P_SMALLD_PART=1 -DC3P_SD_ABSKIP=1
  bash tools/ab_build.sh core c3p_smalld.hip -DC3P_SMALLD_PART=1 -DC3P_SD_ABSKIP=2
  ;;
run)
  O=gpurun_out/ab_complex_border
  mkdir -p $O
  for v in "" skipj core; do
    lib=""; [ -n "$v" ] && lib=$PWD/c3_amd/libc3prop_$v.so
    for b in 256 1024; do
      C3P_LIB=$lib python -c "
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from c3_amd import _lib
if os.environ.get('C3P_LIB'): _lib.LIB_PATH = os.environ['C3P_LIB']
sys.argv = ['bench.py', '--complex', '--batch', '$b', '--steps', '100', '--warmup', '10', '--no-cpu-baseline', '--no-e2e']
runpy.run_path('bench.py', run_name='__main__')" > $O/bench_${v:-regular}_B$b.json 2> $O/bench_${v:-regular}_B$b.err
    done
  done
  python - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob("gpurun_out/ab_complex_border/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        out[os.path.basename(f)[6:-5]] = {"propagators_per_s": d["value"], "ms_per_step": d["ms_per_step"]}
    except Exception as e:
        out[os.path.basename(f)] = str(e)
json.dump({"what": "bench.py --complex (cfg2 operators, one complex control operator), regular build against TIMING-ONLY builds that skip the "
                   "matrix instructions of the border blocks (upper bounds of a core + border form; their results are wrong)", "rows": out},
          open("gpurun_out/ab_complex_border/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  ;;
esac
