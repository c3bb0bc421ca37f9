# PMC passes over the Hermitian-basis Lindblad gradient at cfg4's operators: bash tools/pmc_grad_lindblad_hb.sh [batch]
# (kernel trace + stats, SQ counter passes, FETCH_SIZE / WRITE_SIZE in their own passes) -> gpurun_out/final/grad_lindblad_hb_*
B=${1:-64}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_grad_lindblad_hb.py --batches $B --degrees= --tiled-up-to 0 --reps 2"
P=$R/gpurun_out/gl
rm -rf ${P}_*
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats -o s -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d ${P}_pmc1 -o p1 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d ${P}_pmc2 -o p2 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d ${P}_pmc5 -o p5 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d ${P}_pmc3 -o p3 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d ${P}_pmc4 -o p4 -- $CMD > /dev/null 2>&1
cd $R
python tools/pmc_summary.py ${P}_stats ${P}_pmc1 ${P}_pmc2 ${P}_pmc5 ${P}_pmc3 ${P}_pmc4 > $O/grad_lindblad_hb_B${B}_pmc_summary.txt 2>&1
find ${P}_stats -name "*kernel_stats.csv" -exec cp {} $O/grad_lindblad_hb_B${B}_kernel_stats.csv \;
python - <<PY
import csv, glob, json, collections
P="$P"; B=$B
def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}
res = {}
for kern in ("regr_grad_kernel", "regr_chain_kernel<5, 4, false, true, false, true>"):
    c = {}
    for d in ("pmc1", "pmc2", "pmc5", "pmc3", "pmc4"):
        c.update(counters(P + "_" + d, kern))
    dur = []
    for f in glob.glob(P + "_stats/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not dur:
        continue
    us = sum(dur) / len(dur) / 1e3
    e = {"avg_launch_us": us, "launches": len(dur), "counters_per_launch": c}
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
        valu = 64.0 * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0))
        e["issued_flop_per_launch"] = mfma + valu
        e["issued_frac_of_fp64_peak"] = (mfma + valu) / (cyc * 1024.0 * 32.0)
        e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac_of_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    res[kern.split("<")[0]] = e
json.dump({"what": "Hermitian-basis Lindblad gradient at cfg4's operators, B = %d, N = 1000: per-launch PMC figures of the backward sweep and of the forward chain kernel with Q^T (tools/pmc_grad_lindblad_hb.sh)" % B, "kernels": res}, open("$O/grad_lindblad_hb_B%d_pmc.json" % B, "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x != "counters_per_launch"} for k, v in res.items()}, indent=1))
PY
