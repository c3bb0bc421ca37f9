# PMC passes over the Lindblad gradient of one qutrit / two qubits on the real Hermitian-basis kernels (c3p_smallr.hip):
#   bash tools/pmc_smallr.sh        -> gpurun_out/final/smallr_pmc.json (+ kernel stats)
C=${1:-2}
B=${2:-256}
REPS=${3:-5}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_grad_lindblad.py --cases 3:256:1000,4:256:1000 --no-tiled"
P=$R/gpurun_out/gsr
rm -rf ${P}_*
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_stats -o s -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d ${P}_pmc1 -o p1 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d ${P}_pmc2 -o p2 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d ${P}_pmc5 -o p5 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d ${P}_pmc3 -o p3 -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d ${P}_pmc4 -o p4 -- $CMD > /dev/null 2>&1
cd $R
find ${P}_stats -name "*kernel_stats.csv" -exec cp {} $O/smallr_kernel_stats.csv \;
python - <<PY
import csv, glob, json, collections
P="$P"; B=$B; C=$C
names = set()
for f in glob.glob(P + "_stats/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        names.add(r["Kernel_Name"])
want = [n for n in names if "smallr" in n]
def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == name:
                out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}
res = {}
for kern in sorted(want):
    c = {}
    for d in ("pmc1", "pmc2", "pmc5", "pmc3", "pmc4"):
        c.update(counters(P + "_" + d, kern))
    dur = []
    for f in glob.glob(P + "_stats/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == kern:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not dur :
        continue
    us = sum(dur) / len(dur) / 1e3
    e = {"avg_launch_us": us, "launches": len(dur)}
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
        valu = 64.0 * (2.0 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0))
        e["issued_mfma_flop_per_launch"] = mfma
        e["issued_valu_f64_flop_per_launch"] = valu
        e["issued_frac_of_fp64_peak"] = (mfma + valu) / (cyc * 1024.0 * 32.0)
        e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac_of_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_WAVE_CYCLES"):
            e["issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    e["counters_per_launch"] = c
    res[kern.replace("(anonymous namespace)::", "").replace("void ", "")] = e
json.dump({"what": "Lindblad gradient (c3p_pwc_lindblad_vjp) of one qutrit and of two qubits on the real Hermitian-basis kernels, B = 256, N = 1000: per-launch PMC figures (tools/pmc_smallr.sh; issued = MFMA MOPS x 512 + fp64 vector flops, peak 78.6 TFLOP/s)", "kernels": res},
          open("$O/smallr_pmc.json", "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x != "counters_per_launch"} for k, v in res.items()}, indent=1))
PY
