# rocprofv3 passes for one bench configuration: bash tools/profile_cfg.sh <cfg>
set -x
C=${1:-3}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --ramp-ms 0"
rm -rf $R/gpurun_out/q_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_stats -o s -- $CMD > $R/gpurun_out/q_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/q_pmc1 -o p1 -- $CMD > $R/gpurun_out/q_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/q_pmc2 -o p2 -- $CMD > $R/gpurun_out/q_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/q_pmc3 -o p3 -- $CMD > $R/gpurun_out/q_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/q_pmc4 -o p4 -- $CMD > $R/gpurun_out/q_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/q_stats gpurun_out/q_pmc1 gpurun_out/q_pmc2 gpurun_out/q_pmc3 gpurun_out/q_pmc4 > gpurun_out/cfg${C}_pmc_summary.txt 2>&1
