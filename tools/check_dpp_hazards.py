#!/usr/bin/env python3
"""Static check of the lane-row ODE kernels' ISA: every `v_fmac_f64_dpp` must sit in a run of DPP instructions that opens
with `s_nop` (a VALU write of a VGPR needs wait states before a DPP read of it and the compiler's hazard recogniser does
not see into inline asm -- tools/ubench_dpp.hip shows wrong results without them), and no VALU instruction may write EXEC
(v_cmpx: 5 wait states before a DPP instruction, which nothing would insert).
    python tools/check_dpp_hazards.py [source.hip ...]      (default: both ODE lane-row sources; exit code 1 on a finding)"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "c3_amd", "csrc")


def check(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        os.path.join(CSRC, src), "-o", out], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", "."))]
    ndpp = bad = cmpx = 0
    for i, l in enumerate(lines):
        if l.startswith("v_cmpx"):
            cmpx += 1
        if "v_fmac_f64_dpp" in l:
            ndpp += 1
            prev = lines[i - 1]
            if not ("v_fmac_f64_dpp" in prev or prev.startswith("s_nop")):
                bad += 1
                print(f"{src}: unguarded DPP read: `{prev}` -> `{l[:70]}`")
    print(f"{src}: {ndpp} v_fmac_f64_dpp, {bad} unguarded, {cmpx} v_cmpx")
    return ndpp > 0 and bad == 0 and cmpx == 0


if __name__ == "__main__":
    srcs = sys.argv[1:] or ["c3p_ode_rowq.hip", "c3p_ode_row.hip"]
    sys.exit(0 if all([check(s) for s in srcs]) else 1)
