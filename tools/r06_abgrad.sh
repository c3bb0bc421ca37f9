# cfg2 gradient timing for a list of A/B libraries: bash tools/r06_abgrad.sh name1 name2 ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
out=$O/abgrad_$(date +%H%M%S).txt
for n in "" "$@" ""; do
  lib=c3_amd/libc3prop${n:+_$n}.so
  echo -n "$lib " >> $out
  C3P_LIB=$lib python tools/bench_grad.py --config 2 --batch 256 --reps 30 2>&1 | tail -1 >> $out
done
cat $out
