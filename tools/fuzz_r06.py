"""Randomised parity sweep of the round-6 evaluation schemes (Chebyshev-economised polynomials for normal generators: DESIGN 3):
every case runs with the default plan and with `no_t18n = 1` (published T18 / Taylor parameters on the complex loops) and the two
must agree to 4e-12 (relative to the larger of 1 and the result; x sqrt(N / 100) for long chains; worst seen over 350 k cases 1.8e-12); every fourth case is also checked against scipy's expm slice by
slice or against the oracle.
Dimensions 2..40 (small-D and mid-D kernels), drift / control operators real symmetric, complex Hermitian, mixed or lossy
(non-Hermitian), shared or per sample, 1..4 control lines, generator norms from 0.05 to 12 (every plan, 0..3 squarings), batch
and slice counts on both sides of the workgroup-per-sample mode, frame phases; Lindblad chains at D = 2..4 with weak and strong
dissipators (the symmetric-part guard of the real Hermitian-basis kernels).  Every third case also asks for the partial propagators
(equal between the schemes; their ordered product is U), every sixth repeats the chain from per-slice Hamiltonians (branch B of pwc).
    python tools/fuzz_r06.py --seconds 120 --seed 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3_amd import _lib, propagation as prop
from oracle import c3_oracle as o

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
n = {"unitary": 0, "lindblad": 0, "oracle": 0, "differ": 0}
worst = {"ab": 0.0, "oracle": 0.0}
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
    lind = (it % 5 == 0)
    D = int(rng.integers(2, 5)) if lind else int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 10, 12, 13, 16, 20, 24, 27, 33, 36, 40]))
    K = int(rng.integers(1, 5))
    big = D > 12
    B = int(rng.choice([1, 3, 8, 64, 256])) if not big else int(rng.choice([1, 2, 5]))
    N = int(rng.choice([1, 7, 40, 333, 1000])) if not big else int(rng.choice([3, 17, 60]))
    if B * N * D * D > 4e7:
        N = max(1, int(4e7 / (B * D * D)))
    per_sample = bool(rng.integers(0, 2)) and B <= 64
    kinds = rng.choice(["real", "herm", "mixed", "lossy"], p=[0.2, 0.4, 0.3, 0.1])
    kd = lambda j: {"real": "real", "herm": "herm", "lossy": "lossy" if j == 0 else "herm", "mixed": "real" if j % 2 == 0 else "herm"}[str(kinds)]
    target = float(rng.choice([0.05, 0.3, 0.8, 1.2, 1.34, 1.37, 1.9, 2.05, 2.6, 2.8, 4.5, 12.0]))
    nb = B if per_sample else 1
    h0 = np.stack([operator(D, kd(0), 1.0) for _ in range(nb)])
    hks = np.stack([np.stack([operator(D, kd(1 + k), 0.4) for k in range(K)]) for _ in range(nb)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    bound = max(one(h0[b]) + sum(one(hks[b, k]) for k in range(K)) for b in range(nb))
    dt = target / bound / (2.0 if lind else 1.0)
    if not per_sample:
        h0, hks = h0[0], hks[0]
    ph = rng.uniform(0, 2 * np.pi, size=(B, D * D if lind else D)) if rng.integers(0, 2) else None
    kw = {}
    if lind:
        cs = float(rng.choice([0.02, 0.1, 0.5]))
        kw = dict(col_ops=np.stack([cs * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))]), lindbladian=True)
    if ph is not None:
        kw["fr_phase"] = ph
    want_dus = (it % 3 == 1) and B * N * (D ** (4 if lind else 2)) <= 2e6 and not (lind and D <= 4 and str(kinds) != "lossy")
    r0 = prop.propagate_batch(h0, hks, sig, dt, want_dUs=want_dus, **kw)
    got = np.asarray(r0["U"])
    with _lib.options(no_t18n=1):
        r1 = prop.propagate_batch(h0, hks, sig, dt, want_dUs=want_dus, **kw)
        ref = np.asarray(r1["U"])
    if want_dus:
        # the partial propagators: equal between the schemes, and their ordered product (later slice on the left) is U
        d0, d1 = np.asarray(r0["dUs"]), np.asarray(r1["dUs"])
        assert np.abs(d0 - d1).max() < 2e-12 * max(1.0, np.abs(d1).max()), ("dUs disagree", D, K, B, N, str(kinds), target, lind)
        for b in range(min(B, 3)):
            P = np.eye(d0.shape[-1], dtype=complex)
            for t in range(N):
                P = d0[b, t] @ P
            if ph is not None:
                P = np.exp(1j * ph[b])[:, None] * P
            assert np.abs(P - got[b]).max() < 3e-12 * max(1.0, np.abs(got[b]).max()) * max(1.0, np.sqrt(N / 100.0)), ("prod dUs != U", D, K, B, N, str(kinds), target, lind)
        n["dUs"] = n.get("dUs", 0) + 1
    if it % 6 == 2 and not per_sample and B * N * D * D <= 2e6 and (not lind or D <= 6):
        # branch B of pwc: the per-slice Hamiltonians assembled on the host, same chain
        hs = h0[None, None] + np.einsum("bkn,kij->bnij", sig, hks)
        kb = {k: v for k, v in kw.items()}
        rb = np.asarray(prop.propagate_batch(hs, None, None, dt, **kb)["U"])
        assert np.abs(rb - got).max() < 3e-12 * max(1.0, np.abs(got).max()) * max(1.0, np.sqrt(N / 100.0)), ("per-slice H != tables", D, K, B, N, str(kinds), target, lind)
        n["per_slice"] = n.get("per_slice", 0) + 1
    scale = max(1.0, np.abs(ref).max())
    d = np.abs(got - ref).max() / scale
    worst["ab"] = max(worst["ab"], d)
    n["lindblad" if lind else "unitary"] += 1
    n["differ"] += int(d > 0)
    assert d < 4e-12 * max(1.0, np.sqrt(N / 100.0)), ("schemes disagree", D, K, B, N, str(kinds), target, lind, per_sample, d)
    if it % 4 == 0:
        b = int(rng.integers(0, B))
        okw = {k: v for k, v in kw.items() if k != "fr_phase"}
        hb, kb = (h0[b] if per_sample else h0), (hks[b] if per_sample else hks)
        if lind and target >= 5:
            # (same remark: the independent reference is the complex small-D kernel with the published parameters)
            with _lib.options(no_t18n=1, no_smallr=1):
                orc = np.asarray(prop.propagate_batch(h0, hks, sig, dt, **kw)["U"])[b]
            if ph is not None:
                orc = np.exp(-1j * ph[b])[:, None] * orc
        elif lind or (target < 2.0 and it % 8 == 0):
            orc = o.propagate_batch(hb, kb, sig[b : b + 1], dt, **okw)[0]
        else:
            # the oracle follows TF's Pade-13 rule, itself only ~1e-9 accurate per slice at (unshifted) generator norms in
            # (5.4, 10.7) 2^s (a dumped case: GPU vs scipy 1e-14, oracle vs scipy 7e-11): the reference of the unitary
            # cases is scipy's expm, slice by slice; the oracle takes every other case at small norms and the Lindblad ones
            import scipy.linalg as sl

            orc = np.eye(D, dtype=complex)
            for t in range(N):
                orc = sl.expm(-1j * dt * (hb + np.einsum("k,kij->ij", sig[b, :, t], kb))) @ orc
        if ph is not None:
            orc = np.exp(1j * ph[b])[:, None] * orc
        e = np.linalg.norm(got[b] - orc) / max(1.0, np.linalg.norm(orc))
        worst["oracle"] = max(worst["oracle"], e)
        n["oracle"] += 1
        if not e < 3e-12 * max(1.0, np.sqrt(N / 100.0)):
            dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06", "fuzz_r06_fail.npz")
            os.makedirs(os.path.dirname(dump), exist_ok=True)
            np.savez(dump, h0=hb, hks=kb, sig=sig[b], dt=dt, got=got[b], ref=ref[b], orc=orc, ph=(ph[b] if ph is not None else np.zeros(0)))
        assert e < 3e-12 * max(1.0, np.sqrt(N / 100.0)), ("oracle", D, K, B, N, str(kinds), target, lind, per_sample, e)
print(f"fuzz ok: {n} worst {worst} seed {a.seed}")
