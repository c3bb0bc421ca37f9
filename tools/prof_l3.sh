cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_l3 -o l3 -- python $R/tests/perf/bench_goal_run.py --cases L3:64:1000,L4:256:1000 > $R/gpurun_out/prof_l3.log 2>&1
tail -3 $R/gpurun_out/prof_l3.log | cut -c1-400
python - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_l3/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
