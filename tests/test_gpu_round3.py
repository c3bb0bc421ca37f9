"""Round-3 GPU parity tests (through the C ABI): regressions for the round-2 advisor findings, the lane-row ODE
kernels of c3p_ode_row.hip, and the wider full-size spot checks."""
import os

import numpy as np
import pytest

from c3_amd import workloads
from oracle import c3_oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import _lib, propagation

    _lib.require_gpu()
    return propagation


def fro_max(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return max(np.linalg.norm(a[i] - b[i]) for i in range(a.shape[0]))


# --------------------------------------------------------------------------
# the ordered combine of the register-resident / arena kernels' segment products must not
# write into the buffer it reads (uneven generic segments: the last one holds a single matrix)
# --------------------------------------------------------------------------


@pytest.mark.parametrize("B,N,no_regd", [(1, 800, False), (3, 410, False), (16, 130, False), (2, 400, True)])
def test_segment_combine_does_not_alias_its_input(prop, B, N, no_regd):
    """propagation.py:551-585 + tf_utils.py:144-193 at cfg4's operators with few samples and many time segments."""
    wl = workloads.make_workload(4, B=B, N=N)
    if no_regd:
        os.environ["C3P_NO_REGD"] = "1"
    try:
        r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
        again = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    finally:
        os.environ.pop("C3P_NO_REGD", None)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    assert fro_max(r["U"], ref) < TOL
    assert np.array_equal(np.asarray(r["U"]), np.asarray(again["U"]))


# --------------------------------------------------------------------------
# lane-row ODE kernels (c3p_ode_row.hip): SURVEY 8a rows a11 - a14
# --------------------------------------------------------------------------


def _rand_herm(rng, D, scale, real=False):
    a = rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))
    return (scale * (a + a.conj().T) / 2).astype(complex)


def _ode_problem(D, K, B, N, real, seed):
    rng = np.random.default_rng(seed)
    h0 = _rand_herm(rng, D, 0.3, real)
    hks = np.stack([_rand_herm(rng, D, 0.2, real) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ts = (np.arange(N) + 0.5) * 0.05
    return h0, hks, sig, ts


@pytest.mark.parametrize("D,K", [(2, 1), (3, 2), (4, 1), (5, 3), (6, 2), (7, 4), (9, 2), (11, 1), (12, 3), (13, 2), (16, 4)])
@pytest.mark.parametrize("real", [False, True])
def test_ode_row_schrodinger_dimensions(prop, D, K, real):
    """propagation.py:687-752 + :897-899 on every padded-dimension class of the lane-row kernel, B not a multiple of the
    four samples per wavefront, N not a multiple of the 14-step signal chunk, trajectory and final state."""
    from c3_amd import _lib

    B, N = 7, 33
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 100 * D + K)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "tsit5"):
        out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert _lib.last_kernel() == "ode_row"
        fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
        for b in range(B):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger")["states"]
            assert np.abs(out[b] - ref).max() < 1e-11
            assert np.abs(fin[b] - ref[-1]).max() < 1e-11


@pytest.mark.parametrize("solver", ["rk4", "rk38", "rk5", "tsit5"])
@pytest.mark.parametrize("D,K,C", [(3, 1, 1), (4, 2, 2), (6, 3, 0), (9, 2, 2), (12, 4, 3), (16, 2, 1)])
def test_ode_row_density_matrices(prop, solver, D, K, C):
    """von_neumann (:902-904) for C = 0 / real operators, lindblad (:886-894) with C collapse operators."""
    from c3_amd import _lib

    B, N = 5, 19
    h0, hks, sig, ts = _ode_problem(D, K, B, N, C == 0 and D % 2 == 0, 7 * D + C)
    rng = np.random.default_rng(D + C)
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    rho = a @ a.conj().T
    rho /= np.trace(rho)
    col = np.stack([0.4 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)]) if C else None
    step = "lindblad" if C else "von_neumann"
    out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col))
    assert _lib.last_kernel() == "ode_row"
    fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col, final_only=True))
    for b in range(B):
        ref = o.ode_solver_arrays(h0, hks, sig[b], ts, rho, solver, step, col=col)["states"]
        assert np.abs(out[b] - ref).max() < 1e-11
        assert np.abs(fin[b] - ref[-1]).max() < 1e-11


def test_ode_row_matches_workgroup_kernel_and_cfg2(prop):
    """Same arithmetic as the round-1 workgroup-per-sample kernel (C3P_ODE_WG=1 selects it) on cfg2's operators at
    N = 1000: identical to rounding; trajectory against the oracle on three samples."""
    from c3_amd import _lib

    wl = workloads.make_workload(2, B=37)
    psi = np.zeros((wl.D, 1), complex)
    psi[0, 0] = 1.0
    new = {s: np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, psi, s, "schrodinger")) for s in ("rk4", "rk5")}
    assert _lib.last_kernel() == "ode_row"
    os.environ["C3P_ODE_WG"] = "1"
    try:
        old = {s: np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, psi, s, "schrodinger")) for s in ("rk4", "rk5")}
        assert _lib.last_kernel() == "ode_wg"
    finally:
        os.environ.pop("C3P_ODE_WG")
    for s in new:
        assert np.abs(new[s] - old[s]).max() < 1e-12
    for b in (0, 17, 36):
        ref = o.ode_solver_arrays(wl.h0, wl.hks, wl.signals[b], wl.ts, psi, "rk4", "schrodinger")["states"]
        assert np.abs(new["rk4"][b] - ref).max() < 1e-11


def test_ode_row_rk4_unitary(prop):
    """rk4_unitary (propagation.py:71-101,221-255): the columns of the propagator as B x D vector problems."""
    from c3_amd import _lib

    D, K, B = 6, 2, 3
    Ns = 41
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, False, 5)
    lib = _lib.load()
    import ctypes

    U = np.zeros((B, D, D), complex)
    dUs = np.zeros((B, (Ns - 1) // 2, D, D), complex)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.1, B, K, Ns, D, 1, p(U), p(dUs), None)
    assert rc == 0 and _lib.last_kernel() == "ode_row"
    for b in range(B):
        Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
        ref = o.rk4_unitary_arrays(Hs, 0.1, D)
        assert np.abs(U[b] - ref["U"]).max() < 1e-12
        assert np.abs(dUs[b] - ref["dUs"]).max() < 1e-12
