#!/usr/bin/env python3
"""Coefficients of the economised polynomials of the real small-D / mid-D path (round 6), in EXACT rational arithmetic.

For a real symmetric Y with ||Y|| <= theta the spectrum of W = Y^2 lies in [0, L], L = theta^2, and for ANY polynomial p
    || p(W) - f(W) ||_2 = max over the spectrum | p(w) - f(w) |        (Y is normal),
so f(w) = cos(sqrt w) and g(w) = sin(sqrt w) / sqrt w may be replaced by their best polynomials on [0, L] instead of Taylor
polynomials.  Chebyshev economisation of the degree-20 Taylor polynomials (remainder < 1e-25 for theta <= 2): expand in shifted
Chebyshev polynomials on [0, L], drop everything above degree d; the error is bounded by the sum of the dropped coefficients'
moduli (printed).  Degree 6 reaches 1e-16 at theta = 0.83 (the degree-8 Taylor polynomials need theta_16 = 0.816 and ONE MORE
product: W^4), degree 7 at theta = 1.30 (the degree-9/8 Taylor pair, theta = 1.13, needs 8 products), degree 8 at theta = 1.85.

    python tools/gen_minimax_cossin.py            # prints the C tables pasted into c3_amd/csrc/c3p_common.h
"""
from fractions import Fraction as F
from math import comb, factorial

M = 20


def cheb_T(n):
    T = [[F(1)], [F(0), F(1)]]
    for k in range(2, n + 1):
        a = [F(0)] + [2 * c for c in T[k - 1]]
        b = T[k - 2] + [F(0)] * (len(a) - len(T[k - 2]))
        T.append([x - y for x, y in zip(a, b)])
    return T


def economise(taylor, L, deg):
    n = len(taylor) - 1
    px = [F(0)] * (n + 1)  # in x, w = L (x + 1) / 2
    for j, a in enumerate(taylor):
        for i in range(j + 1):
            px[i] += a * (L / 2) ** j * comb(j, i)
    T = cheb_T(n)
    c = [F(0)] * (n + 1)
    rem = px[:]
    for k in range(n, -1, -1):
        c[k] = rem[k] / T[k][k]
        for i, t in enumerate(T[k]):
            rem[i] -= c[k] * t
    dropped = sum(abs(x) for x in c[deg + 1:])
    qx = [F(0)] * (deg + 1)
    for k in range(deg + 1):
        for i, t in enumerate(T[k]):
            qx[i] += c[k] * t
    pw = [F(0)] * (deg + 1)  # x = 2 w / L - 1
    for m, a in enumerate(qx):
        for i in range(m + 1):
            pw[i] += a * comb(m, i) * (F(2) / L) ** i * F(-1) ** (m - i)
    return pw, dropped


def tables(theta, deg):
    L = theta * theta
    fc = [F((-1) ** j, factorial(2 * j)) for j in range(M + 1)]
    gs = [F((-1) ** j, factorial(2 * j + 1)) for j in range(M + 1)]
    pc, dc = economise(fc, L, deg)
    ps, ds = economise(gs, L, deg)
    # exp(0) = I exactly (idle slices of a chain, a zero Hamiltonian): the constant terms are set to 1; the economised constant of
    # the cosine is 1 - 1.1e-16, so its error bound grows by that much (the error curve moves from [-d, d] to [0, 2 d])
    dc += abs(1 - pc[0])
    ds += abs(1 - ps[0])
    pc[0] = F(1)
    ps[0] = F(1)
    return pc, ps, float(dc), float(ds)


if __name__ == "__main__":
    for name, theta, deg in (("C3P_MM6", F(83, 100), 6), ("C3P_MM7", F(130, 100), 7), ("C3P_MM8", F(185, 100), 8)):
        pc, ps, dc, ds = tables(theta, deg)
        print(f"// degree {deg} in W = Y^2 on ||Y|| <= {float(theta)}: dropped Chebyshev mass cos {dc:.2e}, sin/Y {ds:.2e}")
        print(f"#define {name}_THETA {float(theta)!r}")
        print(f"__device__ __constant__ const double c3p_{name[4:].lower()}_cos[{deg + 1}] = {{" + ", ".join(float(x).hex() for x in pc) + "};")
        print(f"__device__ __constant__ const double c3p_{name[4:].lower()}_sinc[{deg + 1}] = {{" + ", ".join(float(x).hex() for x in ps) + "};")
        print("//   cos :", ", ".join(f"{float(x):.17e}" for x in pc))
        print("//   sinc:", ", ".join(f"{float(x):.17e}" for x in ps))
    for th in (0.83, 0.85, 1.2, 1.3, 1.33, 1.8, 1.85, 1.9):
        for deg in (6, 7, 8):
            _, _, dc, ds = tables(F(th).limit_denominator(1000), deg)
            print(f"// theta {th}: degree {deg}: cos {dc:.2e} sinc {ds:.2e}")
