# cfg4 timing + parity: default and with the Taylor T18 parameters forced
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
out=$O/cfg4_$(date +%H%M%S).txt
for opt in "" "C3P_NO_T18N=1"; do
  echo -n "cfg4 $opt: " >> $out
  env $opt python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g /s %.3f ms err %.2e (%d samples)'%(d['value'],d['ms_per_step'],d['max_fro_err_vs_oracle'],d['oracle_samples_checked']))" >> $out 2>&1
done
cat $out
