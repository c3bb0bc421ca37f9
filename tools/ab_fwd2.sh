cd $GRAFT_REPO_ROOT
python tests/perf/bench_goal_run.py --cases 2:256:1000,2:1024:1000,1:256:200 > gpurun_out/goal_fwd2_on.log 2>&1
C3P_GRAD_FWD_SEG2=0 python tests/perf/bench_goal_run.py --cases 2:256:1000,2:1024:1000,1:256:200 > gpurun_out/goal_fwd2_off.log 2>&1
python -m pytest tests/test_gradient.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_r04.py --seconds 60 --seed 7 2>&1 | tail -2
grep -h fused_ms gpurun_out/goal_fwd2_on.log gpurun_out/goal_fwd2_off.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['case'], r['B'], r['N'], 'three', round(r['three_call_ms'],4), 'fused', round(r.get('fused_ms',0),4), 'graph', r.get('fused_graph_ms'))
"
