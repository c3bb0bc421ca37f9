"""Round 6: the launch log behind INTEGRATION.md's dispatch table, and the round's kernel changes on the headline path."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def prop():
    from c3_amd import _lib, propagation

    _lib.require_gpu()
    return propagation


def test_dispatch_table_rows_reproduce_on_this_box(prop):
    """a spread of rows of profiles/r06/dispatch_table.json, re-measured: the same kernels (template arguments included) run"""
    import dispatch_table as dt

    rows = json.load(open(os.path.join(ROOT, "profiles", "r06", "dispatch_table.json")))["rows"]
    grid = {(e, lab): dict(fixed) for e, lab, fixed, _ in dt.GRID}
    picked = [r for i, r in enumerate(rows) if i % 9 == 0 and r["D"] <= 64]
    assert len(picked) >= 12
    for r in picked:
        kw = dict(grid[(r["entry"], r["regime"])])
        flags, real = kw.pop("flags", ""), kw.pop("real", True)
        fam, det = dt.probe(r["entry"], r["D"], real=real, flags=flags, **kw)
        assert fam == r["family"] and det == r["kernels"], (r, fam, det)


def test_launch_log_names_the_headline_kernel(prop):
    import torch

    from c3_amd import _lib
    from c3_amd.workloads import make_workload
    from oracle import c3_oracle

    w = make_workload(2, B=256, N=1000)
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    U = prop.propagate_batch(t(w.h0), t(w.hks), t(w.signals), w.dt, fr_phase=t(w.fr_phase))["U"]
    torch.cuda.synchronize()
    # one launch per batch: workgroup per sample (MW), core + border form (SPLIT)
    assert _lib.last_kernel_detail() == "c3p_smalld.hip: smalld_chain_kernel<9, false, false, false, true, true>"
    idx = np.array([0, 37, 128, 255])
    ref = c3_oracle.propagate_batch(w.h0, w.hks, w.signals[idx], w.dt, fr_phase=w.fr_phase[idx])
    got = U[torch.as_tensor(idx, device="cuda:0")].cpu().numpy()
    assert max(np.linalg.norm(got[i] - ref[i]) for i in range(len(idx))) < 1e-11


@pytest.mark.parametrize("D", [5, 9])
@pytest.mark.parametrize("squarings", [0, 1, 3])
def test_border_sums_on_the_matrix_cores_all_variants(prop, D, squarings):
    """round 6's form of the core + border loop (reductions against a tile of ones, chain step without lane swaps, Horner factors as
    left operands) against the oracle: both polynomial variants (norm below / above theta_16), squarings (the c-form borders are
    only fetched there and on the first slice), workgroup-per-sample and one-wave workgroups, frame-rotation phases."""
    import torch

    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(600 + D + squarings)
    herm = lambda s: (lambda m: s * (m + m.T) / 2)(rng.normal(size=(D, D))).astype(np.complex128)
    K, N = 2, 96
    for B, scale in ((64, 0.25), (3, 0.25), (64, 0.42)):  # MW / one-wave workgroups; degree-16 and degree-18 variants
        h0 = np.diag(rng.uniform(0, 1, D)).astype(np.complex128) + herm(0.02)
        hks = np.stack([herm(0.3) for _ in range(K)])
        sig = rng.uniform(-1, 1, size=(B, K, N))
        dt = scale * 2.0**squarings
        ph = rng.uniform(0, 6.28, size=(B, D))
        t = lambda a: torch.as_tensor(a, device="cuda:0")
        U = prop.propagate_batch(t(h0), t(hks), t(sig), dt, fr_phase=t(ph))["U"].cpu().numpy()
        assert "true>" in _lib.last_kernel_detail()  # SPLIT instance
        ref = c3_oracle.propagate_batch(h0, hks, sig, dt, fr_phase=ph)
        err = max(np.linalg.norm(U[b] - ref[b]) for b in range(B))
        assert err < 2e-12 * max(1.0, 2.0**squarings), (B, scale, squarings, err)


def _sym_problem(D, K, B, N, rng, target_norm):
    """real symmetric operators scaled so that the kernels' norm bound ||G0||_1 + sum_k max|c_k| ||G_k||_1 (trace-shifted, dt = 1)
    hits `target_norm`"""
    herm = lambda s: (lambda m: s * (m + m.T) / 2)(rng.normal(size=(D, D))).astype(np.complex128)
    h0 = np.diag(rng.uniform(0, 1, D)).astype(np.complex128) + herm(0.05)
    hks = np.stack([herm(0.4) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    bound = one(h0) + sum(np.abs(sig[:, k, :]).max() * one(hks[k]) for k in range(K))
    return h0, hks, sig, target_norm / bound


@pytest.mark.parametrize("D", [3, 5, 9, 12, 14, 20, 27, 33, 36])
def test_economised_polynomials_at_their_thresholds(prop, D):
    """the real path on both sides of theta_6 = 0.83 (degree-6 pair, core + border loop), theta_7 = 1.30, theta_8 = 1.85 (the
    7-product variants; one squaring just above) and deep in the squaring regime -- forward against the oracle, and the
    real backward sweeps (same tables) against the general ones."""
    import torch

    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(900 + D)
    K, B, N = 2, 5, 40
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    for target in (0.80, 0.829, 0.831, 1.29, 1.31, 1.84, 1.86, 3.6, 7.5):
        h0, hks, sig, dt = _sym_problem(D, K, B, N, rng, target)
        U = prop.propagate_batch(t(h0), t(hks), t(sig), dt)["U"].cpu().numpy()
        ref = c3_oracle.propagate_batch(h0, hks, sig, dt)
        err = max(np.linalg.norm(U[b] - ref[b]) for b in range(B))
        assert err < 3e-13 * max(1.0, target) * np.sqrt(D), (D, target, err)
        if D <= 40 and target <= 7.5:
            Ub = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
            g = np.asarray(prop.propagate_batch_vjp(t(h0), t(hks), t(sig), dt, t(Ub)).cpu())
            with _lib.options(no_real_grad=1):
                gg = np.asarray(prop.propagate_batch_vjp(t(h0), t(hks), t(sig), dt, t(Ub)).cpu())
            assert np.abs(g - gg).max() < 1e-10 * max(1.0, np.abs(gg).max()), (D, target)


@pytest.mark.parametrize("D", [5, 9, 12, 13, 20, 27, 36])
def test_normal_generator_schemes_on_the_complex_instances(prop, D):
    """complex HERMITIAN Hamiltonians at D <= 40 (small-D and mid-D complex loops): the prep kernel flags skew-Hermitian generator
    tables and the chain kernel evaluates the four-product scheme (radius 1.35) or T18 with the economised parameters (radius 2.0),
    whichever needs fewer products -- targets on both sides of 1.35, 2.0, 2.7 (= 1.35 with one squaring) and 4.0; a NON-Hermitian
    Hamiltonian keeps the Taylor parameters.  Both against the oracle, and against each other through the no_t18n switch."""
    import torch

    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(700 + D)
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0 = np.diag(rng.uniform(0, 1, D)).astype(complex) + herm(0.05)
    hks = np.stack([herm(0.3) for _ in range(2)])
    B, N = 6, 50
    sig = rng.uniform(-1, 1, size=(B, 2, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    bound = one(h0) + sum(np.abs(sig[:, k, :]).max() * one(hks[k]) for k in range(2))
    for target in (0.9, 1.34, 1.36, 1.99, 2.01, 2.69, 2.72, 4.5, 5.2):
        dt = target / bound
        U = prop.propagate_batch(t(h0), t(hks), t(sig), dt)["U"].cpu().numpy()
        with _lib.options(no_t18n=1):
            V = prop.propagate_batch(t(h0), t(hks), t(sig), dt)["U"].cpu().numpy()
        ref = c3_oracle.propagate_batch(h0, hks, sig, dt)
        assert max(np.linalg.norm(U[b] - ref[b]) for b in range(B)) < 2e-12, (D, target)
        assert max(np.linalg.norm(U[b] - V[b]) for b in range(B)) < 2e-12, (D, target)
    # REAL symmetric drift and first control operator next to one complex Hermitian operator (bench.py --complex): tables with a zero
    # real part count as normal too
    h0r, hkm = h0.real.astype(complex), np.stack([hks[0].real.astype(complex), hks[1]])
    hkm[0] = (hkm[0] + hkm[0].T) / 2
    for target in (1.2, 2.5):
        dt = target / bound
        U = prop.propagate_batch(t(h0r), t(hkm), t(sig), dt)["U"].cpu().numpy()
        name = _lib.last_kernel_detail()
        with _lib.options(no_t18n=1):
            V = prop.propagate_batch(t(h0r), t(hkm), t(sig), dt)["U"].cpu().numpy()
        ref = c3_oracle.propagate_batch(h0r, hkm, sig, dt)
        assert max(np.linalg.norm(U[b] - ref[b]) for b in range(B)) < 2e-12, (D, target, name)
        assert 1e-17 < max(np.linalg.norm(U[b] - V[b]) for b in range(B)) < 2e-12, (D, target)  # not the same scheme
    # not Hermitian: the flag must not be set (the economised polynomial is only valid on an imaginary spectrum)
    h0n = h0 + 0.05 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    dt = 1.6 / bound
    U = prop.propagate_batch(t(h0n), t(hks), t(sig), dt)["U"].cpu().numpy()
    ref = c3_oracle.propagate_batch(h0n, hks, sig, dt)
    assert max(np.linalg.norm(U[b] - ref[b]) for b in range(B)) < 1e-11 * max(1.0, max(np.linalg.norm(ref[b]) for b in range(B)))


@pytest.mark.parametrize("D", [2, 3, 4])
def test_normal_generator_schemes_on_the_small_real_lindblad_kernel(prop, D):
    """Lindblad chains of D = 2, 3, 4 in the Hermitian basis (c3p_smallr.hip): weak dissipators (symmetric part of the real
    generator <= 0.25) take the four-product scheme or T18 with the economised parameters; generator norms from 0.3 to 12 walk
    through every plan.  Against the oracle, and the three schemes against each other (no_t18n = 2: T18N only, 1: published T18)."""
    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(640 + D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    B, K, N = 6, 2, 60
    h0, hks = herm(0.8), np.stack([herm(0.5) for _ in range(K)])
    col = np.stack([0.05 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    differ = 0
    for dt in (0.05, 0.12, 0.2, 0.3, 0.45, 0.7, 1.1, 2.0):
        got = np.asarray(prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True)["U"])
        assert "smallr_chain_kernel" in _lib.last_kernel_detail()
        with _lib.options(no_t18n=2):
            mid = np.asarray(prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True)["U"])
        with _lib.options(no_t18n=1):
            old = np.asarray(prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True)["U"])
        ref = c3_oracle.propagate_batch(h0, hks, sig[:2], dt, col_ops=col, lindbladian=True)
        for b in range(2):
            # (the oracle follows TF's Pade-13 rule, itself ~1e-11 at generator norms above 5: DESIGN 8 -- the bar here, the schemes
            #  against each other two orders tighter)
            assert np.linalg.norm(got[b] - ref[b]) < (2e-12 if dt < 1.0 else 1e-10) * max(1.0, np.linalg.norm(ref[b])), (D, dt)
        assert np.abs(got - mid).max() < 2e-12 and np.abs(got - old).max() < 2e-12, (D, dt)
        differ += int(np.abs(got - mid).max() > 0)
    assert differ >= 2  # the four-product scheme was taken at several of the norms


@pytest.mark.parametrize("D,N,B", [(9, 1000, 256), (5, 640, 256), (9, 333, 64), (12, 1000, 256)])
def test_mixed_batch_real_hermitian_and_lossy_samples_in_the_workgroup_per_sample_mode(prop, D, N, B):
    """per-sample drift Hamiltonians of three kinds in ONE batch -- real symmetric (real loop, 700 per mille split), complex Hermitian
    (complex loop with the normal-generator schemes and its own 640 per mille split: the sample re-splits on the device) and lossy
    (non-Hermitian: published parameters) -- at the workgroup-per-sample shapes of cfg2: every kind against the oracle."""
    import torch

    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(77 + D + N)
    K = 2
    sym = lambda s: (lambda m: s * (m + m.T) / 2)(rng.normal(size=(D, D)))
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    base = np.diag(rng.uniform(0, 1, D)) + sym(0.05)
    h0 = np.empty((B, D, D), dtype=complex)
    for b in range(B):
        kind = b % 3
        h0[b] = base + (0 if kind == 0 else (herm(0.05) if kind == 1 else herm(0.05) - 0.02j * np.diag(np.arange(D))))
    hks = np.stack([sym(0.3).astype(complex) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    dt = 0.9 / (np.abs(base).sum(axis=0).max() + 0.35 * D)
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    U = prop.propagate_batch(t(h0), t(hks), t(sig), dt)["U"].cpu().numpy()
    detail = _lib.last_kernel_detail()
    assert "smalld_chain_kernel" in detail, detail
    for b in (0, 1, 2, 3, 4, 5, B - 3, B - 2, B - 1):
        ref = c3_oracle.propagate_batch(h0[b], hks, sig[b : b + 1], dt)[0]
        assert np.linalg.norm(U[b] - ref) < 2e-11 * max(1.0, np.linalg.norm(ref)), (D, N, b, b % 3)


def test_evaluation_schemes_fuzz_short_run():
    """tools/fuzz_r06.py for a few seconds (the 200-second run of the round: 61 k unitary and 15 k Lindblad cases, 19 k of them against
    scipy / the oracle, worst 1.6e-12 between the schemes: profiles/r06/fuzz_r06.txt)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_r06.py"), "--seconds", "8", "--seed", "11"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("D,forced", [(4, True), (5, False), (3, True), (9, False)])
def test_long_strongly_damped_segments_stay_finite(prop, D, forced):
    """Lindblad chains with a dissipator as strong as the Hamiltonian part, many samples (few, long time segments per sample):
    the trace shift is imaginary only since round 6 -- with the real part of the trace in it the shifted segment product grew like
    e^{|Re mu| n} while e^{sum mu} underflowed, inf * 0 = NaN on the complex kernels (found by tools/fuzz_r06.py at D = 4, B = 256,
    N = 1000).  Finite, trace preserving, and equal to the oracle on a sample; D = 3, 4 forced onto the complex kernels."""
    from c3_amd import _lib
    from oracle import c3_oracle

    rng = np.random.default_rng(31 + D)
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0, hks = herm(1.0), np.stack([herm(0.4)])
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    B, N = (256, 1000) if D <= 5 else (64, 300)
    sig = rng.uniform(-1, 1, size=(B, 1, N))
    dt = 6.0 / (one(h0) + one(hks[0]))
    col = np.stack([0.5 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    with _lib.options(**({"no_smallr": 1} if forced else {})):
        U = np.asarray(prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True)["U"])
    assert np.isfinite(U).all(), _lib.last_kernel_detail()
    vecI = np.eye(D).reshape(-1)
    assert np.abs(np.einsum("i,bij->bj", vecI, U) - vecI).max() < 1e-9
    if D <= 5:
        ref = c3_oracle.propagate_batch(h0, hks, sig[:1], dt, col_ops=col, lindbladian=True)[0]
        # (the oracle's own accuracy at these generator norms is ~1e-9: DESIGN 8)
        assert np.linalg.norm(U[0] - ref) < 1e-8 * max(1.0, np.linalg.norm(ref))


def test_backward_sweeps_fuzz_short_run():
    """tools/fuzz_r06_grad.py for a few seconds (the 100-second run of the round: 120 k cases, every gradient through two different
    kernels <= 8.4e-12 relative, 45 k finite-difference entries <= 1.7e-8: profiles/r06/fuzz_r06.txt)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_r06_grad.py"), "--seconds", "8", "--seed", "13"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
