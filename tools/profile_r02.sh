# rocprofv3 passes for one bench configuration (round 2): bash tools/profile_r02.sh <cfg> [steps]
# kernel trace + stats, two SQ counter passes, FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md:
# TCC slots), then tools/pmc_to_json.py folds them into gpurun_out/pmc_cfg<cfg>.json (-> profiles/r02/pmc.json).
C=${1:-2}
K=${2:-4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $C --steps $K --warmup 1 --no-cpu-baseline --no-e2e --ramp-ms 0"
rm -rf $R/gpurun_out/q${C}_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q${C}_stats -o s -- $CMD > $R/gpurun_out/q${C}_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/q${C}_pmc1 -o p1 -- $CMD > $R/gpurun_out/q${C}_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/q${C}_pmc2 -o p2 -- $CMD > $R/gpurun_out/q${C}_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $R/gpurun_out/q${C}_pmc5 -o p5 -- $CMD > $R/gpurun_out/q${C}_pmc5.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/q${C}_pmc3 -o p3 -- $CMD > $R/gpurun_out/q${C}_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/q${C}_pmc4 -o p4 -- $CMD > $R/gpurun_out/q${C}_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/q${C}_stats gpurun_out/q${C}_pmc1 gpurun_out/q${C}_pmc2 gpurun_out/q${C}_pmc5 gpurun_out/q${C}_pmc3 gpurun_out/q${C}_pmc4 > gpurun_out/cfg${C}_pmc_summary.txt 2>&1
find gpurun_out/q${C}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/cfg${C}_kernel_stats.csv \;
python tools/pmc_to_json.py $C gpurun_out/q${C}_stats gpurun_out/q${C}_pmc1 gpurun_out/q${C}_pmc2 gpurun_out/q${C}_pmc5 gpurun_out/q${C}_pmc3 gpurun_out/q${C}_pmc4 > gpurun_out/pmc_cfg${C}.json
