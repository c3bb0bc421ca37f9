# rocprofv3 passes for one bench configuration (round 3): bash tools/profile_r03.sh <cfg> [steps] [extra bench flags, e.g. --complex]
# The bench command is profiled AS THE DRIVER RUNS IT: with its untimed clock ramp, so that the average launch duration of
# the trace agrees with the line's ms_per_step (round 2 profiled a cold run).  kernel trace + stats, three SQ counter
# passes, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: TCC slots), then tools/pmc_to_json.py folds
# them into gpurun_out/pmc_<tag>.json (-> profiles/r03/pmc.json).
C=${1:-2}
K=${2:-30}
shift; shift
EXTRA="$*"
TAG=cfg${C}
case "$EXTRA" in *--complex*) TAG=cfg${C}_complex;; esac
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $C --steps $K --warmup 2 --no-cpu-baseline --no-e2e $EXTRA"
rm -rf $R/gpurun_out/q_${TAG}_*
P=$R/gpurun_out/q_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats -o s -- $CMD > ${P}_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d ${P}_pmc1 -o p1 -- $CMD > ${P}_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d ${P}_pmc2 -o p2 -- $CMD > ${P}_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d ${P}_pmc5 -o p5 -- $CMD > ${P}_pmc5.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d ${P}_pmc6 -o p6 -- $CMD > ${P}_pmc6.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d ${P}_pmc3 -o p3 -- $CMD > ${P}_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d ${P}_pmc4 -o p4 -- $CMD > ${P}_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py ${P}_stats ${P}_pmc1 ${P}_pmc2 ${P}_pmc5 ${P}_pmc6 ${P}_pmc3 ${P}_pmc4 > gpurun_out/${TAG}_pmc_summary.txt 2>&1
find ${P}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
grep '^{' ${P}_stats.log | tail -1 > gpurun_out/${TAG}_bench_under_rocprof.json
case "$EXTRA" in *--complex*) export C3P_PMC_COMPLEX=1;; *) unset C3P_PMC_COMPLEX;; esac  # (complex Hamiltonians run the padded-tile loop of the same kernel instance)
python tools/pmc_to_json.py $C ${P}_stats ${P}_pmc1 ${P}_pmc2 ${P}_pmc5 ${P}_pmc3 ${P}_pmc4 ${P}_pmc6 > gpurun_out/pmc_${TAG}.json
