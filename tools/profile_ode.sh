# rocprofv3 passes for the ODE bench (lane-row kernels): bash tools/profile_ode.sh [batch] [extra bench_ode args]
# Writes gpurun_out/r03/ode_kernel_stats.csv, ode_pmc_summary.txt (copy into profiles/r03/).
set -x
B=${1:-16384}
shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tests/perf/bench_ode.py --config 2 --batches $B --rho-batches $B --solvers rk4 --steps schrodinger,von_neumann,lindblad --reps 2 $*"
rm -rf $R/gpurun_out/o_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/o_stats -o s -- $CMD > $R/gpurun_out/o_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d $R/gpurun_out/o_pmc1 -o p1 -- $CMD > $R/gpurun_out/o_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $R/gpurun_out/o_pmc2 -o p2 -- $CMD > $R/gpurun_out/o_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/o_pmc3 -o p3 -- $CMD > $R/gpurun_out/o_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/o_pmc4 -o p4 -- $CMD > $R/gpurun_out/o_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/o_stats gpurun_out/o_pmc1 gpurun_out/o_pmc2 gpurun_out/o_pmc3 gpurun_out/o_pmc4 > gpurun_out/r03/ode_pmc_summary.txt 2>&1
cp $(ls gpurun_out/o_stats/*/*kernel_stats.csv gpurun_out/o_stats/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r03/ode_kernel_stats.csv
tail -3 gpurun_out/o_stats.log
