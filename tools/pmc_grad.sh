# PMC passes over tools/bench_grad.py for one configuration: bash tools/pmc_grad.sh <cfg> <batch> [reps]
C=${1:-3}; B=${2:-256}; K=${3:-2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_grad.py --config $C --batch $B --reps $K"
rm -rf $R/gpurun_out/g${C}_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g${C}_stats -o s -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/g${C}_pmc1 -o p1 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/g${C}_pmc2 -o p2 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d $R/gpurun_out/g${C}_pmc5 -o p5 -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/g${C}_stats gpurun_out/g${C}_pmc1 gpurun_out/g${C}_pmc2 gpurun_out/g${C}_pmc5 > gpurun_out/grad_cfg${C}_pmc_summary.txt 2>&1
find gpurun_out/g${C}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/grad_cfg${C}_kernel_stats.csv \;
