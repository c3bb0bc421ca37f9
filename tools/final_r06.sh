# Round-6 measurements on one MI355X; results under gpurun_out/final/ (copy what is to be judged into profiles/r06/).
#   bash tools/final_r06.sh pmc "4 3" "2 30"      rocprofv3 kernel stats + PMC passes of the bench command (cfg, steps), -> pmc_cfgN.json
#   bash tools/final_r06.sh bench                  bench lines of every configuration (+ --complex, one-rank RCCL schedules)
#   bash tools/final_r06.sh grad                   gradient timings
#   bash tools/final_r06.sh goal                   optimiser evaluation, Lindblad gradient, 8 + 1 A/B, ODE trajectories
# PMC entries are merged into profiles/r06/pmc.json per configuration (bench.py quotes an entry only while the digest of
# that configuration's kernel sources still matches), so one configuration can be re-profiled without the others.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O $R/gpurun_out/r06
cd $R
stage=$1
shift
case "$stage" in
pmc)
  for spec in "$@"; do
    set -- $spec
    C=$1; K=$2; shift; shift
    EXTRA="$*"
    TAG=cfg$C
    case "$EXTRA" in *--complex*) TAG=cfg${C}_complex;; esac
    bash tools/profile_r03.sh $C $K $EXTRA
    cp gpurun_out/pmc_$TAG.json gpurun_out/${TAG}_pmc_summary.txt gpurun_out/${TAG}_kernel_stats.csv gpurun_out/${TAG}_bench_under_rocprof.json $O/ 2>/dev/null
  done
  python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
dst = os.path.join(R, "profiles", "r06", "pmc.json")
try:
    out = json.load(open(dst))
except Exception:
    out = {}
out["_comment"] = ("per-launch PMC figures of the dominant kernel of each bench configuration (tools/final_r06.sh pmc -> tools/profile_r03.sh on the "
                   "bench command WITH its clock ramp, tools/pmc_to_json.py); bench.py quotes an entry only when its kernel_sources_digest, batch "
                   "and slice count match the running build")
for tag in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg2_complex"):
    f = os.path.join(R, "gpurun_out", f"pmc_{tag}.json")
    try:
        ent = json.load(open(f))
        if ent.get("issued_flop_per_launch"):
            out[tag] = ent
    except Exception as e:
        print("no new pmc for", tag, e)
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
json.dump(out, open(os.path.join(R, "gpurun_out", "final", "pmc.json"), "w"), indent=1)
PY
  ;;
bench)
  for c in 2 1 3 4 5; do
    python bench.py --config $c --check > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
    tail -c 400 $O/bench_cfg$c.json | head -c 200; echo
  done
  python bench.py > $O/bench_default.json 2> $O/bench_default.err
  python bench.py --complex --check --no-cpu-baseline > $O/bench_cfg2_complex.json 2> $O/bench_cfg2_complex.err
  TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
  $TR --master-port 29611 bench.py --gpus 1 --steps 300 --warmup 10 --no-cpu-baseline --no-e2e > $O/bench_cfg2_rccl_default.json 2> $O/bench_rccl_default.err
  $TR --master-port 29613 bench.py --gpus 1 --steps 300 --warmup 10 --exchange goal --no-cpu-baseline --no-e2e > $O/bench_cfg2_rccl_goal.json 2> $O/bench_rccl_goal.err
  ;;
grad)
  python tools/bench_grad.py --config 2 --batch 256 --reps 20 > $O/grad_cfg2.json
  python tools/bench_grad.py --config 3 --batch 256 --reps 3 > $O/grad_cfg3.json
  python tools/bench_grad.py --config 5 --batch 256 --reps 3 > $O/grad_cfg5.json
  python tools/bench_grad_tiled.py --out $O/grad_tiled.json > /dev/null 2>&1
  python tools/bench_grad_lindblad.py --cases 2:256:1000,3:256:1000,3:16:1000,3:1024:1000,4:256:1000,4:64:1000,5:64:500,6:64:500 --out $O/grad_lindblad_small_mfma.json > /dev/null 2>&1
  ;;
goal)
  # one optimiser evaluation (closed and open systems), the Hermitian-basis Lindblad gradient, the 8 + 1 A/B, ODE trajectories
  python tests/perf/bench_goal_run.py --out $O/goal_run.json > $O/goal_run.log 2>&1
  python tools/bench_grad_lindblad_hb.py --out $O/grad_lindblad_hb.json > $O/grad_lindblad_hb.log 2>&1
  python tools/ab_split81.py --out $O/ab_split81.json > /dev/null 2>&1
  ./tools/ubench_sym9 > $O/ubench_sym9.txt 2>&1
  python tests/perf/bench_ode.py --config 2 --steps schrodinger --solvers rk4,tsit5 --batches 16,64,256,1024 --trajectory --out $O/ode_trajectory_small_batches.json > /dev/null 2>&1
  C3P_ODE_NO_SEG=1 python tests/perf/bench_ode.py --config 2 --steps schrodinger --solvers rk4,tsit5 --batches 16,64,256,1024 --trajectory --out $O/ode_trajectory_small_batches_direct.json > /dev/null 2>&1
  python tests/perf/bench_ode.py --config 3 --steps schrodinger --solvers rk4 --batches 4,16,64,256 --trajectory --out $O/ode_trajectory_cfg3_small_batches.json > /dev/null 2>&1
  C3P_ODE_NO_SEG=1 python tests/perf/bench_ode.py --config 3 --steps schrodinger --solvers rk4 --batches 4,64 --trajectory --out $O/ode_trajectory_cfg3_small_batches_direct.json > /dev/null 2>&1
  ;;
esac
ls $O | wc -l
