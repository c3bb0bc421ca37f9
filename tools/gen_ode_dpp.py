#!/usr/bin/env python3
"""Generates c3_amd/csrc/c3p_ode_dpp.inc: the broadcast-fused fp64 FMA primitives of the lane-row ODE kernels.

gfx950 has `v_fmac_f64_dpp ... row_newbcast:J` (DP-ALU DPP): D += bcast_row(S0, lane J) * S1, one instruction, full fp64
rate, no LDS and no separate lane move.  The compiler does not select it from `__builtin_amdgcn_update_dpp` + fma (it emits
v_mov_b64_dpp + v_fma_f64, +50 % instructions), so the primitives are inline-asm blocks; an asm statement takes at most 30
operands, hence one specialisation per padded dimension DP with the operand lists spelled out -- which is what this script
writes.  Hazard note: a VALU write of a VGPR needs two wait states before a DPP read of it and the hazard recogniser does
not look into inline asm, so every block opens with `s_nop 1`.

    python tools/gen_ode_dpp.py > c3_amd/csrc/c3p_ode_dpp.inc
"""
import sys

DPS = [2, 3, 4, 6, 8, 9, 12, 16]
TAIL = "row_mask:0xf bank_mask:0xf"


def block(lines, outs, ins, imm=None):
    body = "\\n\\t".join(["s_nop 1"] + lines)
    o = ", ".join(f'[{n}] "+v"({e})' for n, e in outs)
    i = ", ".join(f'[{n}] "v"({e})' for n, e in ins)
    if imm:
        i += f', [J] "n"({imm})'
    return f'    asm("{body}"\n        : {o}\n        : {i});\n'


def chunks(n, size):
    out, c0 = [], 0
    while c0 < n:
        nc = min(size, n - c0)
        # balance the last two chunks
        if n - c0 > size and n - c0 < 2 * size:
            nc = (n - c0 + 1) // 2
        out.append((c0, nc))
        c0 += nc
    return out


def gen_matvec_c(dp):
    s = f"  // w[0] += sum_j hr[j] yr(j), w[1] += sum_j hr[j] yi(j), w[2] -= sum_j hi[j] yi(j), w[3] += sum_j hi[j] yr(j);  y(j) = lane j of the row\n"
    s += f"  static __device__ __forceinline__ void matvec_c(double (&w)[4], double yr, double yi, const double (&hr)[{dp}], const double (&hi)[{dp}]) {{\n"
    for c0, nc in chunks(dp, 12):
        lines, ins = [], [("yr", "yr"), ("yi", "yi")]
        for j in range(c0, c0 + nc):
            lines += [
                f"v_fmac_f64_dpp %[w0], %[yr], %[hr{j}] row_newbcast:{j} {TAIL}",
                f"v_fmac_f64_dpp %[w1], %[yi], %[hr{j}] row_newbcast:{j} {TAIL}",
                f"v_fmac_f64_dpp %[w2], %[yi], -%[hi{j}] row_newbcast:{j} {TAIL}",
                f"v_fmac_f64_dpp %[w3], %[yr], %[hi{j}] row_newbcast:{j} {TAIL}",
            ]
            ins += [(f"hr{j}", f"hr[{j}]"), (f"hi{j}", f"hi[{j}]")]
        s += block(lines, [(f"w{q}", f"w[{q}]") for q in range(4)], ins)
    s += "  }\n"
    return s


def gen_matvec_r(dp):
    s = f"  // real H row: w[0] / w[2] += hr[j] yr(j) (even / odd j), w[1] / w[3] += hr[j] yi(j)\n"
    s += f"  static __device__ __forceinline__ void matvec_r(double (&w)[4], double yr, double yi, const double (&hr)[{dp}]) {{\n"
    lines, ins = [], [("yr", "yr"), ("yi", "yi")]
    for j in range(dp):
        a, b = (0, 1) if j % 2 == 0 else (2, 3)
        lines += [
            f"v_fmac_f64_dpp %[w{a}], %[yr], %[hr{j}] row_newbcast:{j} {TAIL}",
            f"v_fmac_f64_dpp %[w{b}], %[yi], %[hr{j}] row_newbcast:{j} {TAIL}",
        ]
        ins.append((f"hr{j}", f"hr[{j}]"))
    s += block(lines, [(f"w{q}", f"w[{q}]") for q in range(4)], ins)
    s += "  }\n"
    return s


def gen_bmac(dp, kind):
    """W[c] (+/-)= s * bcast_J(V[c]) for all c.  kind: cc (complex own scalar, complex V), rs (real scalar, complex V),
    rv (complex scalar, real V)."""
    per = {"cc": 7, "rs": 7, "rv": 9}[kind]
    if kind == "cc":
        sig = f"double (&wr)[{dp}], double (&wi)[{dp}], double sr, double si, const double (&vr)[{dp}], const double (&vi)[{dp}]"
        doc = "wr[c] += sr vr[c](J) - si vi[c](J); wi[c] += sr vi[c](J) + si vr[c](J)"
    elif kind == "rs":
        sig = f"double (&wr)[{dp}], double (&wi)[{dp}], double sr, const double (&vr)[{dp}], const double (&vi)[{dp}]"
        doc = "real own scalar: wr[c] += sr vr[c](J); wi[c] += sr vi[c](J)"
    else:
        sig = f"double (&wr)[{dp}], double (&wi)[{dp}], double sr, double si, const double (&vr)[{dp}]"
        doc = "real broadcast row: wr[c] += sr vr[c](J); wi[c] += si vr[c](J)"
    s = f"  // {doc}   (NEG: -=);  x(J) = lane J of the 16-lane row\n"
    s += f"  template <int J, bool NEG>\n  static __device__ __forceinline__ void bmac_{kind}({sig}) {{\n"
    for neg in (False, True):
        s += f"    if constexpr (NEG == {'true' if neg else 'false'}) {{\n"
        p, m = ("-", "") if neg else ("", "-")
        for c0, nc in chunks(dp, per):
            lines, late, outs = [], [], []  # `late`: the second term of every element, issued after all first terms
            ins = [("sr", "sr")] + ([("si", "si")] if kind != "rs" else [])
            for c in range(c0, c0 + nc):
                if kind == "cc":
                    lines += [
                        f"v_fmac_f64_dpp %[wr{c}], %[vr{c}], {p}%[sr] row_newbcast:%[J] {TAIL}",
                        f"v_fmac_f64_dpp %[wi{c}], %[vi{c}], {p}%[sr] row_newbcast:%[J] {TAIL}",
                    ]
                    late += [
                        f"v_fmac_f64_dpp %[wr{c}], %[vi{c}], {m}%[si] row_newbcast:%[J] {TAIL}",
                        f"v_fmac_f64_dpp %[wi{c}], %[vr{c}], {p}%[si] row_newbcast:%[J] {TAIL}",
                    ]
                    ins += [(f"vr{c}", f"vr[{c}]"), (f"vi{c}", f"vi[{c}]")]
                elif kind == "rs":
                    lines += [
                        f"v_fmac_f64_dpp %[wr{c}], %[vr{c}], {p}%[sr] row_newbcast:%[J] {TAIL}",
                        f"v_fmac_f64_dpp %[wi{c}], %[vi{c}], {p}%[sr] row_newbcast:%[J] {TAIL}",
                    ]
                    ins += [(f"vr{c}", f"vr[{c}]"), (f"vi{c}", f"vi[{c}]")]
                else:
                    lines += [
                        f"v_fmac_f64_dpp %[wr{c}], %[vr{c}], {p}%[sr] row_newbcast:%[J] {TAIL}",
                        f"v_fmac_f64_dpp %[wi{c}], %[vr{c}], {p}%[si] row_newbcast:%[J] {TAIL}",
                    ]
                    ins += [(f"vr{c}", f"vr[{c}]")]
                outs += [(f"wr{c}", f"wr[{c}]"), (f"wi{c}", f"wi[{c}]")]
            lines += late
            s += "  " + block(lines, outs, ins, imm="J").replace("\n        ", "\n          ")
        s += "    }\n"
    s += "  }\n"
    return s


def main():
    out = sys.stdout
    out.write("// GENERATED by tools/gen_ode_dpp.py -- do not edit; see that script for the why.\n")
    out.write("// Broadcast-fused fp64 FMA primitives (v_fmac_f64_dpp row_newbcast) for the lane-row ODE kernels, one\n")
    out.write("// specialisation per padded dimension DP (asm operand lists must be literal).\n")
    out.write("#pragma once\n\ntemplate <int DP>\nstruct OdeDpp;\n\n")
    for dp in DPS:
        out.write(f"template <>\nstruct OdeDpp<{dp}> {{\n")
        out.write(gen_matvec_c(dp))
        out.write(gen_matvec_r(dp))
        for kind in ("cc", "rs", "rv"):
            out.write(gen_bmac(dp, kind))
        out.write("};\n\n")


if __name__ == "__main__":
    main()
