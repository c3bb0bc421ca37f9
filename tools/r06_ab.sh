# time_cfg2 for a list of A/B libraries: bash tools/r06_ab.sh name1 name2 ... -> gpurun_out/r06/ab_<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
out=$O/ab_$(date +%H%M%S).txt
python tools/time_cfg2.py c3_amd/libc3prop.so >> $out 2>&1
for n in "$@"; do python tools/time_cfg2.py c3_amd/libc3prop_$n.so 2>&1 | tail -1 >> $out; done
python tools/time_cfg2.py c3_amd/libc3prop.so 2>&1 | tail -1 >> $out
cat $out
