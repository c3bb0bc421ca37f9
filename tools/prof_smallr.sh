cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sr -o sr -- python $R/tools/bench_lindblad_small_real.py > $R/gpurun_out/prof_sr.log 2>&1
python - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_sr/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:95].ljust(95), r["Calls"].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
