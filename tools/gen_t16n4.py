#!/usr/bin/env python3
"""A FOUR-product evaluation of exp for NORMAL generators with an imaginary spectrum (round 6).

The scheme (the "15+" family of Sastre, Ibanez & Defez: 16 parameters, degree 16, 4 matrix products):
    A2 = A^2,   y0 = A2 (c0 A2 + c1 A),
    y1 = (y0 + c2 A2 + c3 A)(y0 + c4 A2) + c5 y0 + c6 A2,
    y2 = (y1 + c7 A2 + c8 A)(y1 + c9 y0 + c10 A) + c11 y1 + c12 y0 + c13 A2 + c14 A + c15 I.
Its 16 parameters fix the coefficients of A^0 .. A^15; the coefficient of A^16 follows from them (0.5457 / 16! when the others
are 1/k!).  For a normal X with spectrum on [-i theta, i theta] the matrix error is the scalar error on that interval, so the
parameters are solved for the Chebyshev-ECONOMISED target instead of the Taylor one: with c16 the scheme's own A^16 coefficient,
    even part  cos(sqrt w) - c16 w^8  -> degree 7 in w = y^2,        odd part  y sin(sqrt w) / sqrt w -> degree 7 in w,
iterated to the fixed point of c16 (two rounds), Gauss-Newton in 70-digit arithmetic from a double-precision solution of the
Taylor system (found once with scipy.optimize.least_squares in the scaled variable A / 0.5; hard-coded below).  The radius is
1.35 (T18 with its published parameters: 5 products, 1.13 for any matrix; this scheme with Taylor targets: ~0.70 for any matrix).

    python tools/gen_t16n4.py [theta]          # prints the table for c3_amd/csrc/c3p_common.h
"""
import importlib.util
import os
import sys
from decimal import Decimal as Dc, getcontext
from fractions import Fraction as F
from math import factorial

getcontext().prec = 70
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tools", "gen_minimax_cossin.py"))
g = importlib.util.module_from_spec(spec)
spec.loader.exec_module(g)

S = Dc("0.5")  # the solve runs in x = A / S (coefficients of order one)
START = [-2.5117260063756473e-05, -3.6819143003496036e-04, 2.1772666442094169e-03, -2.0087842203367845e-01, -8.0769072203057766e-03,
         2.3373194047117150e-02, 6.5373199432452886e-02, -5.9526759346774621e-02, -2.0651381829648853e-02, -5.7923617070732627e+00,
         1.1121045862481871e+00, 1.0408017352313539e+01, 3.0301234007387539e+00, -5.3243889762410845e-01, 5.0e-01, 1.0]


def pmul(a, b, n=16):
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


def scal(c, p):
    return [c * x for x in p]


def scheme(c):
    x = [Dc(0), Dc(1)]
    A2 = [Dc(0), Dc(0), Dc(1)]
    y0 = pmul(A2, padd(scal(c[0], A2), scal(c[1], x)))
    y1 = padd(pmul(padd(y0, scal(c[2], A2), scal(c[3], x)), padd(y0, scal(c[4], A2))), scal(c[5], y0), scal(c[6], A2))
    y2 = padd(pmul(padd(y1, scal(c[7], A2), scal(c[8], x)), padd(y1, scal(c[9], y0), scal(c[10], x))), scal(c[11], y1), scal(c[12], y0),
              scal(c[13], A2), scal(c[14], x), [c[15]])
    return (y2 + [Dc(0)] * 17)[:17]


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


def newton(v, tgt):
    """scaled parameters v so that coefficient k of the scheme (in x = A / S) equals tgt[k] S^k, k = 0..15"""
    want = [tgt[k] * S**k for k in range(16)]
    rn = None
    for _ in range(40):
        T = scheme(v)
        r = [T[k] / want[k] - 1 for k in range(16)]
        rn = max(abs(x) for x in r)
        if rn < Dc("1e-55"):
            break
        h = Dc("1e-38")
        J = [[Dc(0)] * 16 for _ in range(16)]
        for j in range(16):
            w = v[:]
            w[j] += h
            Tw = scheme(w)
            for k in range(16):
                J[k][j] = (Tw[k] - T[k]) / want[k] / h
        d = solve_linear(J, r)
        v = [v[j] - d[j] for j in range(16)]
    return v, rn


def economised_target(theta, c16):
    """c_k, k = 0..15, of the degree-15 polynomial closest to e^z - c16 z^16 on [-i theta, i theta]; error bounds of both parts"""
    g.M = 24
    L = theta * theta
    fc = [F((-1) ** j, factorial(2 * j)) for j in range(g.M + 1)]
    fc[8] -= c16  # cos(sqrt w) - c16 w^8   ((iy)^16 = y^16 = w^8)
    gs = [F((-1) ** j, factorial(2 * j + 1)) for j in range(g.M + 1)]
    pc, dc = g.economise(fc, L, 7)
    ps, ds = g.economise(gs, L, 7)
    c = [F(0)] * 16
    for j in range(8):
        c[2 * j] = (-1) ** j * pc[j]
        c[2 * j + 1] = (-1) ** j * ps[j]
    return c, float(dc), float(ds) * float(theta)


def solve(theta):
    v = [Dc(x) for x in START]
    taylor = [Dc(1) / Dc(factorial(k)) for k in range(16)]
    v, rn = newton(v, taylor)
    c16 = scheme(v)[16] / S**16
    ratio = c16 * Dc(factorial(16))
    for _ in range(4):
        c16f = F(int(c16 * Dc(10) ** 60), 10**60)
        tgt, dc, ds = economised_target(theta, c16f)
        v, rn = newton(v, [Dc(x.numerator) / Dc(x.denominator) for x in tgt])
        c16 = scheme(v)[16] / S**16
    # unscaled parameters (A instead of A / S)
    pw = [4, 3, 2, 1, 2, 0, 2, 2, 1, 0, 1, 0, 0, 2, 1, 0]
    u = [v[j] / S ** pw[j] for j in range(16)]
    return u, rn, dc, ds, float(ratio), float(c16 * Dc(factorial(16)))


if __name__ == "__main__":
    theta = F(sys.argv[1]).limit_denominator(1000) if len(sys.argv) > 1 else F(135, 100)
    u, rn, dc, ds, r0, r1 = solve(theta)
    print(f"// 4-product scheme for normal generators, |spectrum| <= {float(theta)}: error bound even part {dc:.2e}, odd part {ds:.2e};")
    print(f"// A^16 coefficient x 16! = {r0:.6f} (Taylor targets) -> {r1:.6f}; residual of the parameter solve {float(rn):.1e}")
    print(f"#define C3P_E4N_THETA {float(theta)!r}")
    vals = [float(x) for x in u]
    vals[15] = 1.0  # exp(0) = I exactly (the solved constant is 1 - 1.1e-16)
    print("__constant__ double c3p_e4n[16] = {" + ", ".join(x.hex() for x in vals) + "};")
    print("//   " + ", ".join(f"{float(x):.17g}" for x in u))
