# What the cfg2 step costs besides its slices: kernel time against the slice count and with the table build / the segment
# combine as separate launches (C3P_PREP_KERNEL, C3P_NO_FUSE).  bash tools/fixed_cost_cfg2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # label, env, slices
  rm -rf /tmp/fx; env $2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fx -o s -- python $R/bench.py --config 2 --slices $3 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  python - "$1" "$3" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/fx/**/s_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'smalld' in r['Name']:
            print(sys.argv[1], 'N=' + sys.argv[2], r['Name'][27:75], 'calls', r['Calls'], 'avg_us %.2f' % (float(r['AverageNs']) / 1e3), 'min_us %.2f' % (float(r['MinNs']) / 1e3))
PY
}
run default A=1 1000
run default A=1 128
run default A=1 64
run default A=1 32
run prepkernel C3P_PREP_KERNEL=1 1000
run prepkernel C3P_PREP_KERNEL=1 32
run nofuse C3P_NO_FUSE=1 1000
run nofuse C3P_NO_FUSE=1 32
