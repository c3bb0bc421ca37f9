# rocprofv3 passes for the matrix-core rho ODE kernel (c3p_ode_rhoq.hip): bash tools/profile_ode_rho.sh [config] [batch]
# Writes gpurun_out/r03/ode_rho_kernel_stats.csv, ode_rho_pmc_summary.txt (copy into profiles/r03/).
set -x
C=${1:-3}
B=${2:-1536}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tests/perf/bench_ode.py --config $C --rho-batches $B --solvers rk4 --steps von_neumann,lindblad --synth-col --reps 2"
rm -rf $R/gpurun_out/q_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_stats -o s -- $CMD > $R/gpurun_out/q_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/q_pmc1 -o p1 -- $CMD > $R/gpurun_out/q_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/q_pmc2 -o p2 -- $CMD > $R/gpurun_out/q_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/q_pmc5 -o p5 -- $CMD > $R/gpurun_out/q_pmc5.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/q_pmc3 -o p3 -- $CMD > $R/gpurun_out/q_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/q_pmc4 -o p4 -- $CMD > $R/gpurun_out/q_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/q_stats gpurun_out/q_pmc1 gpurun_out/q_pmc2 gpurun_out/q_pmc5 gpurun_out/q_pmc3 gpurun_out/q_pmc4 > gpurun_out/r03/ode_rho_pmc_summary.txt 2>&1
cp $(ls gpurun_out/q_stats/*/*kernel_stats.csv gpurun_out/q_stats/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/r03/ode_rho_kernel_stats.csv
tail -3 gpurun_out/q_stats.log
