# cfg3 / cfg5 forward timing + parity for the main library (and optional A/B library names)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
out=$O/mid_$(date +%H%M%S).txt
for n in "" "$@"; do
  lib=c3_amd/libc3prop${n:+_$n}.so
  for c in 3 5; do
    echo -n "$lib cfg$c " >> $out
    C3P_LIB=$lib python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g /s %.3f ms err %.2e (%d samples) kernel %s'%(d['value'],d['ms_per_step'],d['max_fro_err_vs_oracle'],d['oracle_samples_checked'],d['config']['kernel']))" >> $out 2>&1
  done
done
cat $out
