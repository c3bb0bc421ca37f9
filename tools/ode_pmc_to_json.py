#!/usr/bin/env python3
"""Folds the rocprofv3 passes of tools/profile_ode.sh (gpurun_out/o_stats, o_pmc1..4) into profiles/r03/ode_roofline.json:
per ODE kernel instance -- average launch time, issued fp64 VALU flops (all lanes), RK steps per second, fraction of the
78.6 TFLOP/s fp64 vector peak, HBM-side bytes.      python tools/ode_pmc_to_json.py <B> <N> > profiles/r03/ode_roofline.json"""
import collections, csv, glob, json, os, sys

B, N = int(sys.argv[1]), int(sys.argv[2])
root = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "o_stats", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("o_pmc1", "o_pmc2", "o_pmc3", "o_pmc4"):
    for f in glob.glob(os.path.join(root, d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"peak_fp64_tflops": 78.6, "batch": B, "steps_per_sample": N, "kernels": {}}
for k, ds in dur.items():
    if "ode_" not in k or "colprep" in k:
        continue
    # both instances (real / complex operators) are launched and one exits at once: keep the launches that did the work
    work = [t for t in ds if t > 50e-6]
    if not work:
        continue
    c = {n: v for n, v in cnt[k].items()}
    # counters of the early-exit launches are ~0: average over the working launches = sum / number of working launches
    nw = max(1, round(len(work) * len(next(iter(c.values()))) / len(ds))) if c else 1
    tot = lambda n: sum(c.get(n, [0.0])) / nw
    fl = 64.0 * (2.0 * tot("SQ_INSTS_VALU_FMA_F64") + tot("SQ_INSTS_VALU_ADD_F64") + tot("SQ_INSTS_VALU_MUL_F64"))
    t = sum(work) / len(work)
    name = k.replace("void (anonymous namespace)::", "").split("(")[0]
    out["kernels"][name] = {
        "avg_launch_ms": t * 1e3, "working_launches_in_trace": len(work), "rk_steps_per_s": B * N / t,
        "issued_fp64_valu_flop_per_launch": fl, "issued_tflops": fl / t * 1e-12, "issued_frac_of_peak": fl / t * 1e-12 / 78.6,
        "valu_instructions_per_step_and_wave": tot("SQ_INSTS_VALU") / max(1.0, tot("SQ_WAVES") or (B / 4.0)) / N if c else None,
        "hbm_bytes_per_launch": (2.0 * tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024.0,
    }
print(json.dumps(out, indent=1))
