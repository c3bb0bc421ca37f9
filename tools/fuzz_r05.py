"""Randomised parity sweep of the pinwheel class (D = 25..28, real symmetric operators; c3p_midd.hip Sched::PW): forward against
the complex instance of the same kernel (`no_real`) and, every fourth case, against the oracle; the real backward sweep against
the general one; random batch, slices, control lines (1..6), drive strength (all polynomial variants, 0..3 squarings), segments,
slice propagators, frame phases, per-sample operators.
    python tools/fuzz_r05.py --seconds 120 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import _lib, propagation as prop
from oracle import c3_oracle as o

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
sym = lambda D, s: (lambda m: (s * (m + m.T) / 2).astype(complex))(rng.normal(size=(D, D)))
n = {"fwd": 0, "oracle": 0, "grad": 0}
worst = {"fwd": 0.0, "oracle": 0.0, "grad": 0.0}
t_end = time.time() + a.seconds
it = 0
while time.time() < t_end:
    it += 1
    D = int(rng.integers(25, 29))
    B, K, N = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.integers(5, 90))
    amp = float(rng.choice([0.1, 0.3, 0.8, 1.1, 1.4, 2.5, 6.0, 11.0]))
    per_sample = bool(rng.integers(0, 2))
    shp = (B,) if per_sample else ()
    h0 = np.stack([sym(D, amp * 1e10) for _ in range(B)]) if per_sample else sym(D, amp * 1e10)
    hks = np.stack([np.stack([sym(D, 1.0) for _ in range(K)]) for _ in range(B)]) if per_sample else np.stack([sym(D, 1.0) for _ in range(K)])
    sig = rng.normal(size=(B, K, N)) * amp * 4e9 / np.sqrt(K)
    ph = rng.uniform(0, 6, size=(B, D)) if rng.integers(0, 2) else None
    S = int(rng.choice([0, 1, 2, 3, 5]))
    dus = bool(rng.integers(0, 2))
    with _lib.options(segments=S):
        r = prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph, want_dUs=dus)
        with _lib.options(no_real=1):
            r2 = prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph, want_dUs=dus)
    U, U2 = np.asarray(r["U"]), np.asarray(r2["U"])
    tol = 3e-12 * max(1.0, amp) ** 2 * max(1.0, N / 30)
    e = float(np.abs(U - U2).max())
    if dus:
        e = max(e, float(np.abs(np.asarray(r["dUs"]) - np.asarray(r2["dUs"])).max()))
    n["fwd"] += 1; worst["fwd"] = max(worst["fwd"], e / tol)
    assert e < tol, ("forward", D, B, K, N, amp, S, dus, per_sample, e)
    if it % 4 == 0:
        b = int(rng.integers(0, B))
        ref = o.pwc_arrays(h0[b] if per_sample else h0, hks[b] if per_sample else hks, sig[b], 1e-11)["U"]
        if ph is not None:
            ref = np.exp(1j * ph[b])[:, None] * ref
        e = float(np.linalg.norm(U[b] - ref))
        n["oracle"] += 1; worst["oracle"] = max(worst["oracle"], e / (3 * tol))
        assert e < 3 * tol, ("oracle", D, B, K, N, amp, S, e)
    if it % 2 == 0 and K <= 3 and amp <= 1.4:
        Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
        g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
        with _lib.options(no_real_grad=1):
            g2 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
        e = float(np.abs(g - g2).max() / max(np.abs(g2).max(), 1e-300))
        n["grad"] += 1; worst["grad"] = max(worst["grad"], e / 1e-11)
        assert e < 1e-11, ("grad", D, B, K, N, amp, per_sample, e)
print("fuzz_r05", n, "worst / tolerance", {k: round(v, 4) for k, v in worst.items()}, "OK")
