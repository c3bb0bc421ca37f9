"""Fold the rocprofv3 passes of tools/profile_r03.sh into one JSON entry for profiles/r03/pmc.json: per-launch
averages of the DOMINANT kernel (largest total time in the kernel trace), HBM bytes with the gfx950 FETCH_SIZE correction
of MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide reads: x2), issued flops from the instruction counters.

    python tools/pmc_to_json.py <cfg> <dirs...>
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def tile_utilisation(Dm, kernel):
    """Fraction of every issued MFMA tile that multiplies matrix data (the rest is zero padding), from the tile geometry
    of the kernel class that ran:
      smalld (Dm <= 12): 4 x 4 blocks, rows and columns padded to 4 ceil(Dm / 4); the K dimension is exact when
                         Dm = 1 (mod 4) (rank-1 tail on the vector unit), otherwise padded as well;
      midd (13..40):     16-row units and 4-column blocks, K in steps of 4 (real instance, Dm = 25..28: pinwheel deal, 28^3);
      regd (49/65/81):   the 16 n x 16 n core tiles exactly, the border runs on the vector unit: 1.0;
      others:            1.0 (not corrected)."""
    if Dm in (5, 9) and not os.environ.get("C3P_PMC_COMPLEX") and "smalld_chain_kernel" in kernel and kernel.split("smalld_chain_kernel")[-1].split(">")[0].rstrip().endswith("true"):
        # core + border form of the real path (round 4): the (Dm-1)^2 core tiles exactly; of the matrix instructions of a slice
        # (degree-16 variant: 7 symmetric products of NC^2 (NC+1)/2, the chain step's 3 NC^3 core and 3 NC^2 row-border ones)
        # only the row-border instructions of the chain step carry padding (one of four A rows)
        # round 6: the border sums run on the matrix cores as well, and the degree-6 economised pair needs 6 symmetric products
        # (cfg1 / cfg2 scale below theta_6 = 0.83): per slice 6 products of NC^2 (NC+1)/2 exact core instructions + (NC + 1)
        # reductions against a tile of ones each, the chain step's 3 NC^3 core, 3 NC^2 row-border (the border row in all four
        # rows of the A tile: 1/4 useful) and 2 (NC + 1) reduction instructions.  A reduction instruction adds 4 x 4 partial
        # products to 4 sums per block: 12 useful additions of its 128 flops.
        nc = (Dm - 1) // 4
        sym, core, row = 6 * nc * nc * (nc + 1) // 2, 3 * nc**3, 3 * nc * nc
        red = 6 * (nc + 1) + 2 * (nc + 1)
        return (sym + core + 0.25 * row + (12.0 / 128.0) * red) / (sym + core + row + red), (
            f"small-D kernel, core + border form: {sym} + {core} exact core MFMAs, {row} row-border MFMAs at 1/4 and {red} "
            f"reduction MFMAs at 12/128 per slice (degree-6 economised variant)")
    if Dm <= 12:
        p = 4 * ((Dm + 3) // 4)
        kk = 1.0 if Dm % 4 == 1 else Dm / p
        return (Dm / p) ** 2 * kk, f"small-D kernel: ({Dm}/{p})^2 rows x columns" + ("" if Dm % 4 == 1 else f" x {Dm}/{p} in K")
    if 33 <= Dm <= 36 and "midd_chain_kernel" in kernel and "true" in kernel.split("midd_chain_kernel")[-1][:40]:
        # real instance with the 32 + 4 row split (round 4): four full 16 x 16 units, two wide units (rows 32..35 exactly), three
        # 16 x 4 units of column block 8 -- 4 x 64 + 2 x 16 + 3 x 16 matrix-pipe cycles per K-step
        rv = (Dm - 32) / 4.0
        useful = (Dm / 36.0) * (256.0 + 32.0 * rv + 32.0 * rv + 16.0 * ((Dm - 32) / 16.0) * rv)
        return useful / 336.0, f"mid-D real instance, 32 + 4 row split: exact rows in the wide units, column block 8 and its last unit carry {Dm - 32} of 4 / 16"
    if 25 <= Dm <= 28 and "midd_chain_kernel" in kernel and "true" in kernel.split("midd_chain_kernel")[-1][:40]:
        # real instance, pinwheel deal (round 5): per product 4 x 21 v_mfma_f64_4x4x4_4b of the four 3 x 4 / 4 x 3 block
        # rectangles and 2 K-packed ones of the centre block, 256 multiply-adds each
        return Dm**3 / (86 * 256.0), f"mid-D real instance, pinwheel deal: {Dm}^3 of 86 x 256 multiply-adds per product (28 x 28 x 28 tiles + the K-packed centre block)"
    if Dm <= 40:
        rows = 16 * ((Dm + 15) // 16)
        cols = 4 * ((Dm + 3) // 4)
        kp = 4 * ((Dm + 3) // 4)
        return (Dm / rows) * (Dm / cols) * (Dm / kp), f"mid-D kernel: {Dm}/{rows} rows x {Dm}/{cols} columns x {Dm}/{kp} in K"
    return 1.0, "no padded MFMA work (register-resident core / not corrected)"


def main():
    cfg = int(sys.argv[1])
    dirs = sys.argv[2:]
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            break
        if dur:
            break
    dom = max(dur, key=lambda k: sum(dur[k]))
    counters = collections.defaultdict(list)
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"] == dom:
                    counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                            "scratch_bytes_per_lane": int(r["Scratch_Size"]), "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    import bench
    from c3_amd import workloads

    c = workloads.CONFIGS[cfg]
    _, lo, hi, _ = bench.plan_batch(c, "weak", None, 1, 0)
    D = 1
    for d in c["dims"]:
        D *= d
    Dm = D * D if c["lindblad"] else D
    out = {
        "kernel": dom[:160],
        "launches_in_trace": len(dur[dom]),
        "avg_launch_us": sum(dur[dom]) / len(dur[dom]) / 1e3,
        # (the mean takes in the launches of the clock ramp; the median is the steady launch)
        "median_launch_us": sorted(dur[dom])[len(dur[dom]) // 2] / 1e3,
        "min_launch_us": min(dur[dom]) / 1e3,
        "batch": hi - lo,
        "slices": c["N"],
        "kernel_sources_digest": bench.kernel_sources_digest(cfg),
        "resources": meta,
        "counters_per_launch": avg,
    }
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        out["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
        out["hbm_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts half the bytes of wide (16 B / lane) reads (MI355X_MICROARCH.md); WRITE_SIZE uncalibrated"
    mfma = avg.get("SQ_INSTS_MFMA")
    if mfma is not None:
        # wave-level v_mfma_f64_4x4x4_4b (4 blocks x 4x4x4 MAC) and v_mfma_f64_16x16x4 differ 4x in flops; MOPS_F64 counts
        # 512-flop units when available
        mops = avg.get("SQ_INSTS_VALU_MFMA_MOPS_F64")
        mfma_flop = (mops * 512.0) if mops else mfma * 512.0
        valu_flop = 64.0 * (2.0 * avg.get("SQ_INSTS_VALU_FMA_F64", 0.0) + avg.get("SQ_INSTS_VALU_ADD_F64", 0.0) + avg.get("SQ_INSTS_VALU_MUL_F64", 0.0))
        out["issued_mfma_flop_per_launch"] = mfma_flop
        out["issued_valu_f64_flop_per_launch"] = valu_flop
        out["issued_flop_per_launch"] = mfma_flop + valu_flop
        util, why = tile_utilisation(Dm, dom)
        out["mfma_tile_utilisation"] = util
        out["mfma_tile_utilisation_note"] = why
        # Active lanes of the vector instructions: SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), both from
        # the same kind of pass (SQ_ACTIVE_INST_VALU is averaged over passes 1 and 6), is the average
        # fraction of the 64 lanes that execute over ALL vector instructions, matrix instructions included (those run with
        # all lanes) -- so it bounds the fraction for the fp64 vector arithmetic from ABOVE only when the exec-masked
        # instructions are the arithmetic ones; it is applied to the vector flops as measured, no finer split exists in the counters.
        lane_util = None
        if avg.get("SQ_THREAD_CYCLES_VALU") and avg.get("SQ_ACTIVE_INST_VALU"):
            lane_util = min(1.0, avg["SQ_THREAD_CYCLES_VALU"] / (64.0 * avg["SQ_ACTIVE_INST_VALU"]))
            out["valu_active_lane_fraction"] = lane_util
        out["useful_flop_per_launch"] = mfma_flop * util + valu_flop * (lane_util if lane_util is not None else 1.0)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg:
        out["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles); launch cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs: it reproduces launch time x 2.4 GHz)"
        if avg.get("GRBM_GUI_ACTIVE"):
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            out["launch_cycles"] = cyc
            out["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            out["issued_frac_of_fp64_peak"] = out.get("issued_flop_per_launch", 0.0) / (cyc * 1024.0 * 32.0)
            # VERDICT r5 item 1 asked for a shared-pipe busy figure.  No counter gives vector-unit busy CYCLES on gfx950
            # (SQ_ACTIVE_INST_VALU counts instructions: it equals SQ_INSTS_VALU, which INCLUDES the matrix instructions --
            # checked on tools/ubench_coissue.hip), so the figure is instruction counts x MEASURED issue costs
            # (profiles/r06/ubench_coissue.txt, ubench_valu64.txt: a SIMD issues matrix and vector instructions one after the
            # other, nothing overlaps; v_mfma_f64_4x4x4_4b 16.7 cycles, 16x16x4 65.1; an fp64 vector instruction 5.8 with two waves
            # per SIMD (4.8 - 6.4 by operand form; 8.2 with ONE wave, 5.3 with four); other vector instructions 2.6 - 4.5, priced 3.5)
            n_mfma = avg.get("SQ_INSTS_MFMA", 0.0)
            n_f64 = avg.get("SQ_INSTS_VALU_FMA_F64", 0.0) + avg.get("SQ_INSTS_VALU_ADD_F64", 0.0) + avg.get("SQ_INSTS_VALU_MUL_F64", 0.0)
            n_other = max(0.0, avg.get("SQ_INSTS_VALU", 0.0) - n_mfma - n_f64)
            waves_per_simd = max(1.0, meta.get("workgroup", 64) / 64.0 / 4.0) if meta else 2.0
            mops = avg.get("SQ_INSTS_VALU_MFMA_MOPS_F64") or n_mfma
            c_mfma = 16.7 * (mops / n_mfma) if n_mfma else 16.7  # (a 16x16x4 instruction is four 512-flop units: 65 cycles)
            c_f64 = 8.2 if waves_per_simd < 1.5 and meta.get("vgpr", 0) + meta.get("agpr", 0) > 256 else 5.8
            busy = n_mfma * c_mfma + n_f64 * c_f64 + n_other * 3.5
            out["issue_busy_model"] = {
                "mfma": n_mfma * c_mfma / (cyc * 1024.0), "valu_f64": n_f64 * c_f64 / (cyc * 1024.0), "valu_other": n_other * 3.5 / (cyc * 1024.0),
                "total": busy / (cyc * 1024.0), "cycles_per_mfma": c_mfma, "cycles_per_f64_valu": c_f64, "cycles_per_other_valu": 3.5,
                "instructions_per_launch": {"mfma": n_mfma, "valu_f64": n_f64, "valu_other": n_other},
                "note": "fraction of the SIMD cycles of the launch taken by the serialised issue of matrix + vector instructions (counts x measured costs)",
            }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
