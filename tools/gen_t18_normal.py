#!/usr/bin/env python3
"""T18 for NORMAL generators with an imaginary spectrum (round 6): the 5-product scheme of Bader, Blanes & Casas
    A2 = A^2, A3 = A2 A, A6 = A3^2, B1 = a11 A + a21 A2 + a31 A3, B5 = b24 A2 + b34 A3 + b64 A6,
    B4 = b03 I + b13 A + b23 A2 + b33 A3 + b63 A6, A9 = B1 B5 + B4, B3 = b02 I + b12 A + b22 A2 + b32 A3 + b62 A6,
    B2 = b11 A + b21 A2 + b31 A3 + b61 A6, T18 = B2 + (B3 + A9) A9
evaluates a degree-18 polynomial whose 19 coefficients are polynomial functions of the 20 parameters.  The published
parameters reproduce 1/k! (backward-error radius 1.13 for ANY matrix).  For X = -i dt (H - tr H / D) with H Hermitian -- and
for Lindblad generators with a Hermitian H up to their (small) dissipator -- X is normal with spectrum on [-i theta, i theta],
so ||p(X) - exp(X)||_2 = max_y |p(iy) - e^{iy}| and the Taylor coefficients may be replaced by the Chebyshev-ECONOMISED ones:
even part = the degree-9 economisation of cos(sqrt w), odd part = y x the degree-8 economisation of sin(sqrt w) / sqrt w on
[0, theta^2] (tools/gen_minimax_cossin.py).  This script re-solves the 20 parameters for that target by Gauss-Newton in
60-digit arithmetic (minimum-norm steps from the published values: 19 equations, 20 unknowns) and prints the table for
c3_amd/csrc/c3p_common.h.

    python tools/gen_t18_normal.py [theta]
"""
import importlib.util
import os
import sys
from decimal import Decimal as Dc, getcontext
from fractions import Fraction as F

getcontext().prec = 70
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tools", "gen_minimax_cossin.py"))
g = importlib.util.module_from_spec(spec)
spec.loader.exec_module(g)

NAMES = ["a11", "a21", "a31", "b11", "b21", "b31", "b61", "b02", "b12", "b22", "b32", "b62", "b03", "b13", "b23", "b33", "b63", "b24", "b34", "b64"]
TAYLOR = dict(a11="-0.10036558103014462001", a21="-0.00802924648241156960", a31="-0.00089213849804572995",
              b11="0.39784974949964507614", b21="1.36783778460411719922", b31="0.49828962252538267755", b61="-0.00063789819459472330",
              b02="-10.9676396052962062593", b12="1.68015813878906197182", b22="0.05717798464788655127", b32="-0.00698210122488052084",
              b62="0.00003349750170860705", b03="-0.09043168323908105619", b13="-0.06764045190713819075", b23="0.06759613017704596460",
              b33="0.02955525704293155274", b63="-0.00001391802575160607", b24="-0.09233646193671185927", b34="-0.01693649390020817171",
              b64="-0.00001400867981820361")


def pmul(a, b, n=18):
    out = [Dc(0)] * (n + 1)
    for i, x in enumerate(a):
        if x == 0:
            continue
        for j, y in enumerate(b):
            if i + j <= n:
                out[i + j] += x * y
    return out


def padd(*ps):
    n = max(len(p) for p in ps)
    return [sum((p[i] if i < len(p) else Dc(0)) for p in ps) for i in range(n)]


def t18_coeffs(v):
    p = dict(zip(NAMES, v))
    mono = lambda k, c: [Dc(0)] * k + [c]
    B1 = padd(mono(1, p["a11"]), mono(2, p["a21"]), mono(3, p["a31"]))
    B5 = padd(mono(2, p["b24"]), mono(3, p["b34"]), mono(6, p["b64"]))
    B4 = padd(mono(0, p["b03"]), mono(1, p["b13"]), mono(2, p["b23"]), mono(3, p["b33"]), mono(6, p["b63"]))
    A9 = padd(pmul(B1, B5), B4)
    B3 = padd(mono(0, p["b02"]), mono(1, p["b12"]), mono(2, p["b22"]), mono(3, p["b32"]), mono(6, p["b62"]))
    B2 = padd(mono(1, p["b11"]), mono(2, p["b21"]), mono(3, p["b31"]), mono(6, p["b61"]))
    T = padd(B2, pmul(padd(B3, A9), A9))
    return (T + [Dc(0)] * 19)[:19]


def solve_linear(M, r):
    n = len(M)
    A = [row[:] + [r[i]] for i, row in enumerate(M)]
    for c in range(n):
        piv = max(range(c, n), key=lambda i: abs(A[i][c]))
        A[c], A[piv] = A[piv], A[c]
        for i in range(c + 1, n):
            f = A[i][c] / A[c][c]
            if f != 0:
                for j in range(c, n + 1):
                    A[i][j] -= f * A[c][j]
    x = [Dc(0)] * n
    for i in range(n - 1, -1, -1):
        x[i] = (A[i][n] - sum(A[i][j] * x[j] for j in range(i + 1, n))) / A[i][i]
    return x


def target(theta):
    """c_k of p(z) = sum c_k z^k ~ e^z on z in [-i theta, i theta]: c_2j = (-1)^j cos_j, c_2j+1 = (-1)^j sinc_j"""
    g.M = 24
    pc, _, dc, _ = g.tables(theta, 9)
    _, ps, _, ds = g.tables(theta, 8)
    c = [F(0)] * 19
    for j in range(10):
        c[2 * j] = (-1) ** j * pc[j]
    for j in range(9):
        c[2 * j + 1] = (-1) ** j * ps[j]
    return c, dc, ds * float(theta)


def solve(theta):
    c, dc, ds = target(theta)
    cd = [Dc(x.numerator) / Dc(x.denominator) for x in c]
    v = [Dc(TAYLOR[n]) for n in NAMES]
    for it in range(60):
        T = t18_coeffs(v)
        r = [T[k] - cd[k] for k in range(19)]
        rn = max(abs(x) for x in r)
        if rn < Dc("1e-45"):
            break
        h = Dc("1e-35")
        J = [[Dc(0)] * 20 for _ in range(19)]
        for j in range(20):
            w = v[:]
            w[j] += h
            Tw = t18_coeffs(w)
            for k in range(19):
                J[k][j] = (Tw[k] - T[k]) / h
        JJt = [[sum(J[a][m] * J[b][m] for m in range(20)) for b in range(19)] for a in range(19)]
        y = solve_linear(JJt, r)
        step = [sum(J[k][j] * y[k] for k in range(19)) for j in range(20)]
        v = [v[j] - step[j] for j in range(20)]
    return v, rn, dc, ds


if __name__ == "__main__":
    theta = F(sys.argv[1]).limit_denominator(1000) if len(sys.argv) > 1 else F(24, 10)
    # sanity: the published parameters reproduce 1/k!
    from math import factorial
    T = t18_coeffs([Dc(TAYLOR[n]) for n in NAMES])
    print("// published parameters vs 1/k!: max relative deviation %.1e" % max(float(abs(T[k] * factorial(k) - 1)) for k in range(19)))
    v, rn, dc, ds = solve(theta)
    print(f"// T18 for normal generators, |spectrum| <= {float(theta)}: economisation error bound even part {dc:.2e}, odd part {ds:.2e}; residual of the parameter solve {float(rn):.1e}")
    print(f"#define C3P_T18N_THETA {float(theta)!r}")
    for n, x in zip(NAMES, v):
        print(f"#define C3P_T18N_{n.upper()} ({float(x).hex()})  /* {x:.25f} */")
    # double-precision check: coefficients of the scheme with the ROUNDED parameters against the target
    vd = [Dc(float(x)) for x in v]
    Tr = t18_coeffs(vd)
    c, _, _ = target(theta)
    th = float(theta)
    print("// with the parameters rounded to double: sum_k |coeff error| theta^k = %.2e" % sum(float(abs(Tr[k] - Dc(c[k].numerator) / Dc(c[k].denominator))) * th**k for k in range(19)))
