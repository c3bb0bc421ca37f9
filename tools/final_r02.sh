# Round-2 closing measurements on one MI355X: PMC passes of the bench command for cfg2..5 (-> profiles/r02/pmc.json, so that
# the bench lines that follow quote traffic / issued flops for THIS build), the five bench lines, gradient timings and
# counters, the side benches.  bash tools/final_r02.sh   (everything lands under gpurun_out/final/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
for spec in "2 30" "3 4" "5 3" "4 3"; do
  set -- $spec
  bash tools/profile_r02.sh $1 $2
  cp gpurun_out/pmc_cfg$1.json gpurun_out/cfg$1_pmc_summary.txt gpurun_out/cfg$1_kernel_stats.csv $O/ 2>/dev/null
done
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
out = {"_comment": "per-launch PMC figures of the dominant kernel of each bench configuration (tools/profile_r02.sh, tools/pmc_to_json.py); bench.py quotes them only when kernel_sources_digest and batch match"}
for c in (2, 3, 4, 5):
    f = os.path.join(R, "gpurun_out", f"pmc_cfg{c}.json")
    try:
        out[f"cfg{c}"] = json.load(open(f))
    except Exception as e:
        print("no pmc for cfg", c, e)
json.dump(out, open(os.path.join(R, "profiles", "r02", "pmc.json"), "w"), indent=1)
json.dump(out, open(os.path.join(R, "gpurun_out", "final", "pmc.json"), "w"), indent=1)
PY
for c in 2 1 3 4 5; do
  python bench.py --config $c --check > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
  tail -c 600 $O/bench_cfg$c.json | head -c 300; echo
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/bench_grad.py --config 2 --batch 256 --reps 20 > $O/grad_cfg2.json
python tools/bench_grad.py --config 2 --batch 4096 --reps 5 > $O/grad_cfg2_B4096.json
python tools/bench_grad.py --config 3 --batch 256 --reps 3 > $O/grad_cfg3.json
python tools/bench_grad.py --config 5 --batch 256 --reps 3 > $O/grad_cfg5.json
for c in 2 3 5; do C3P_NO_REAL_GRAD=1 python tools/bench_grad.py --config $c --batch 256 --reps 3 > $O/grad_cfg${c}_general_sweep.json; done
bash tools/pmc_grad.sh 2 256 10; bash tools/pmc_grad.sh 3 256 2; bash tools/pmc_grad.sh 5 256 2
cp gpurun_out/grad_cfg*_pmc_summary.txt gpurun_out/grad_cfg*_kernel_stats.csv $O/
python tests/checks/check_tiled.py --time > $O/tiled_check.txt 2>&1
python tests/perf/bench_complex_path.py > $O/complex_path_cfg2.json 2> /dev/null
python tools/bench_midd_real_vs_complex.py > $O/midd_real_vs_complex.json 2> /dev/null
cat $O/grad_cfg2.json $O/grad_cfg3.json $O/grad_cfg5.json
