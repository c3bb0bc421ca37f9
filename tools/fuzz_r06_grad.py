"""Randomised A/B of the backward sweeps after round 6 (economised polynomials in the real sweeps, schemes for normal generators,
imaginary-only trace shift): every case computes the control gradient twice through DIFFERENT kernels and the two must agree to
1e-10 relative:
  unitary, real symmetric operators (D = 2..40):   real backward sweep                 vs  general sweep (no_real_grad)
  unitary, complex Hermitian (D = 2..40):          default                             vs  published parameters (no_t18n = 1)
  Lindblad D = 2..4, Hermitian H:                  real Hermitian-basis sweep          vs  complex sweeps (no_smallr)
every eighth case also against central finite differences of the forward path on three random (sample, line, slice) entries.
    python tools/fuzz_r06_grad.py --seconds 120 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3_amd import _lib, propagation as prop

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
n = {"real": 0, "complex": 0, "lindblad": 0, "fd": 0}
worst = {"ab": 0.0, "fd": 0.0}
t_end = time.time() + a.seconds


def operator(D, kind, s):
    m = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    if kind == "real":
        return (s * (m.real + m.real.T) / 2).astype(complex)
    h = s * (m + m.conj().T) / 2
    if kind == "lossy":
        h = h - 0.03j * s * np.diag(rng.uniform(0, 1, D))
    return h


it = 0
while time.time() < t_end:
    it += 1
    mode = ("real", "complex", "lindblad")[it % 3]
    D = int(rng.integers(2, 5)) if mode == "lindblad" else int(rng.choice([2, 3, 5, 8, 9, 9, 12, 13, 16, 20, 27, 33, 36, 40]))
    K = int(rng.integers(1, 4))
    big = D > 12
    B = int(rng.choice([1, 3, 17, 64])) if not big else int(rng.choice([1, 2, 4]))
    N = int(rng.choice([1, 7, 40, 300])) if not big else int(rng.choice([3, 17, 40]))
    kind = "real" if mode == "real" else "herm"  # (the gradient entries require Hermitian Hamiltonians)
    target = float(rng.choice([0.05, 0.4, 0.8, 0.85, 1.3, 1.4, 1.8, 1.9, 2.6, 3.5, 7.0]))
    h0 = operator(D, kind, 1.0)
    hks = np.stack([operator(D, "herm" if kind == "lossy" else kind, 0.4) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    dt = target / (one(h0) + sum(one(h) for h in hks)) / (2.0 if mode == "lindblad" else 1.0)
    Dm = D * D if mode == "lindblad" else D
    Ub = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm)) if rng.integers(0, 2) else None
    if mode == "lindblad":
        col = np.stack([float(rng.choice([0.02, 0.2])) * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
        f = lambda: np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ub, fr_phase=ph))
        fwd = lambda s: np.asarray(prop.propagate_batch(h0, hks, s, dt, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
        other = dict(no_smallr=1)
    else:
        f = lambda: np.asarray(prop.propagate_batch_vjp(h0, hks, sig, dt, Ub, fr_phase=ph))
        fwd = lambda s: np.asarray(prop.propagate_batch(h0, hks, s, dt, fr_phase=ph)["U"])
        other = dict(no_real_grad=1) if mode == "real" else dict(no_t18n=1)
    g = f()
    with _lib.options(**other):
        g2 = f()
    assert np.isfinite(g).all() and np.isfinite(g2).all(), ("not finite", mode, D, K, B, N, kind, target)
    scale = max(1e-30, np.abs(g2).max())
    d = np.abs(g - g2).max() / scale
    worst["ab"] = max(worst["ab"], d)
    n[mode] += 1
    assert d < 1e-10, ("sweeps disagree", mode, D, K, B, N, kind, target, d)
    if it % 8 == 0:
        for _ in range(3):
            b, k, t = int(rng.integers(0, B)), int(rng.integers(0, K)), int(rng.integers(0, N))
            eps = 1e-6
            sp, sm = sig.copy(), sig.copy()
            sp[b, k, t] += eps
            sm[b, k, t] -= eps
            # goal = Re <U_bar, U>: d goal / d c = Re sum conj(U_bar) dU
            fd = np.real(np.vdot(Ub[b], fwd(sp)[b] - fwd(sm)[b])) / (2 * eps)
            e = abs(fd - g[b, k, t]) / max(1.0, abs(fd), np.abs(g[b]).max())
            worst["fd"] = max(worst["fd"], e)
            n["fd"] += 1
            assert e < 2e-6, ("finite differences", mode, D, K, B, N, kind, target, b, k, t, fd, g[b, k, t])
print(f"fuzz ok: {n} worst {worst} seed {a.seed}")
