"""Randomised parity sweep of the large-dimension and odd-shape paths (register-resident 49 / 65 / 81 kernels, arena kernel, tiled
path, generic fallback): D = 41..130 unitary and Lindblad D = 5..10 (Dm = 25..100), B = 1..3, N = 1..12, K = 0..9 control lines
(K > 8 leaves the table kernels), real / Hermitian / lossy operators, norms 0.1..6.  Every case: default dispatch against the generic
vector-unit kernel (`force_generic`) <= 5e-12 relative, one sample against scipy's expm slice by slice; every fourth unitary case with
41 <= D <= 64 also the control gradient on the default sweep against the tiled sweep (`tiled_grad`).
    python tools/fuzz_r06_large.py --seconds 120 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sl
from c3_amd import _lib, propagation as prop
from oracle import c3_oracle as o

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
n = {"unitary": 0, "lindblad": 0, "grad": 0}
worst = {"generic": 0.0, "scipy": 0.0, "grad": 0.0}
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
    lind = it % 3 == 0
    D = int(rng.integers(5, 11)) if lind else int(rng.choice([41, 44, 48, 49, 50, 63, 64, 65, 66, 80, 81, 82, 92, 93, 100, 130]))
    K = int(rng.choice([0, 1, 2, 3, 9]))
    B, N = int(rng.integers(1, 4)), int(rng.choice([1, 2, 5, 12]))
    kind = str(rng.choice(["real", "herm", "lossy"])) if not lind else "herm"
    target = float(rng.choice([0.1, 0.9, 1.5, 2.5, 6.0]))
    h0 = operator(D, kind, 1.0)
    hks = np.stack([operator(D, "real" if kind == "real" else "herm", 0.3) for _ in range(max(K, 1))])[:K]
    sig = rng.uniform(-1, 1, size=(B, K, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    dt = target / (one(h0) + sum(one(h) for h in hks)) / (2.0 if lind else 1.0)
    kw = {}
    if lind:
        kw = dict(col_ops=np.stack([float(rng.choice([0.02, 0.3])) * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))]), lindbladian=True)
    Dm = D * D if lind else D
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm)) if rng.integers(0, 2) else None
    got = np.asarray(prop.propagate_batch(h0, hks, sig, dt, fr_phase=ph, **kw)["U"])
    name = _lib.last_kernel()
    assert np.isfinite(got).all(), ("not finite", D, K, B, N, kind, target, lind, name)
    if Dm <= 256:
        gen = np.asarray(prop.propagate_batch(h0, hks, sig, dt, fr_phase=ph, force_generic=True, **kw)["U"])
        d = np.abs(got - gen).max() / max(1.0, np.abs(gen).max())
        worst["generic"] = max(worst["generic"], d)
        assert d < 5e-12, ("default != generic", D, K, B, N, kind, target, lind, name, d)
    b = int(rng.integers(0, B))
    if lind:
        ref = o.propagate_batch(h0, hks, sig[b : b + 1], dt, **kw)[0] if target < 2.0 else None
    else:
        ref = np.eye(D, dtype=complex)
        for t in range(N):
            ref = sl.expm(-1j * dt * (h0 + np.einsum("k,kij->ij", sig[b, :, t], hks))) @ ref
    if ref is not None:
        if ph is not None:
            ref = np.exp(1j * ph[b])[:, None] * ref
        e = np.linalg.norm(got[b] - ref) / max(1.0, np.linalg.norm(ref))
        worst["scipy"] = max(worst["scipy"], e)
        assert e < 1e-11, ("reference", D, K, B, N, kind, target, lind, name, e)
    n["lindblad" if lind else "unitary"] += 1
    if not lind and it % 4 == 0 and D <= 64 and kind != "lossy" and K > 0:
        Ub = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
        g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, dt, Ub, fr_phase=ph))
        with _lib.options(tiled_grad=1):
            g2 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, dt, Ub, fr_phase=ph))
        d = np.abs(g - g2).max() / max(1e-30, np.abs(g2).max())
        worst["grad"] = max(worst["grad"], d)
        n["grad"] += 1
        assert d < 1e-9, ("gradient sweeps disagree", D, K, B, N, kind, target, d)
print(f"fuzz ok: {n} worst {worst} seed {a.seed}")
