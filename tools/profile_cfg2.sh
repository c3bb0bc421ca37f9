set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats -o s -- $CMD > $R/gpurun_out/p_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $R/gpurun_out/p_pmc1 -o p1 -- $CMD > $R/gpurun_out/p_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/p_pmc2 -o p2 -- $CMD > $R/gpurun_out/p_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/p_pmc3 -o p3 -- $CMD > $R/gpurun_out/p_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/p_pmc4 -o p4 -- $CMD > $R/gpurun_out/p_pmc4.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/p_stats gpurun_out/p_pmc1 gpurun_out/p_pmc2 gpurun_out/p_pmc3 gpurun_out/p_pmc4 > gpurun_out/cfg2_pmc_summary.txt 2>&1
find gpurun_out/p_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/cfg2_kernel_stats.csv
tail -3 gpurun_out/p_stats.log
