"""Parity of the HIP path against the CPU oracle, through the C ABI.  -m gpu."""
import numpy as np
import pytest

from oracle import c3_oracle as o
from c3_amd import workloads

pytestmark = pytest.mark.gpu

TOL = 1e-10  # north_star: |U_gpu - U_ref|_F < 1e-10


@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import propagation, _lib

    _lib.require_gpu()
    return propagation


def fro_max(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return max(np.linalg.norm(a[i] - b[i]) for i in range(a.shape[0]))


@pytest.mark.parametrize("cfg,B,N", [(1, 3, 200), (2, 5, 300), (3, 2, 60), (5, 2, 40)])
def test_unitary_batch_vs_oracle(prop, cfg, B, N):
    wl = workloads.make_workload(cfg, B=B, N=N)
    r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, fr_phase=wl.fr_phase, want_dUs=True)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, fr_phase=wl.fr_phase)
    assert fro_max(r["U"], ref) < TOL
    dref = o.tf_propagation_vectorized(wl.h0, wl.hks, wl.signals[0], wl.dt)
    assert np.abs(np.asarray(r["dUs"][0]) - dref).max() < 1e-13


def test_lindblad_batch_vs_oracle(prop):
    wl = workloads.make_workload(4, B=2, N=24)
    ph = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
    r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=ph)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=wl.fr_phase)
    assert fro_max(r["U"], ref) < TOL


def test_golden_two_qubit(prop, golden_dir):
    g = np.load(golden_dir + "/two_qubit.npz")
    sig = np.stack([g["sig_d1"], g["sig_d2"]])
    hks = np.stack([g["hk_d1"], g["hk_d2"]])
    dt = g["ts"][1] - g["ts"][0]
    r = prop.propagate_batch(g["hdrift"], hks, sig[None], dt)
    assert np.linalg.norm(np.asarray(r["U"][0]) - g["propagator"]) < 1e-11


def test_golden_tunable_coupler(prop, golden_dir):
    """D = 27 known-answer test on the MFMA mid-D kernel: the reference's stored partial propagators of
    the 10 000-slice CPHASE gate (test/test_tunable_coupler.py:393-403), branch A and branch B."""
    from c3_amd.workloads import tunable_coupler_problem

    g = np.load(golden_dir + "/tunable_coupler.npz")
    h0, hk = tunable_coupler_problem()
    sig = g["tc_signal"]
    dt = g["tc_ts"][1] - g["tc_ts"][0]
    r = prop.propagate_batch(h0, hk[None], sig[None, None, :], dt, want_dUs=True)
    dUs = np.asarray(r["dUs"][0])
    assert np.abs(dUs[g["dU_slice_index"]] - g["dUs"]).max() < 1e-12
    U_ref = o.tf_matmul_left(o.expm(-1j * dt * (h0[None] + sig[:, None, None] * hk[None])))
    assert np.linalg.norm(np.asarray(r["U"][0]) - U_ref) < TOL
    # branch B: the per-slice Hamiltonians handed over whole (propagation.py:296-298)
    n = 1500
    H = h0[None] + sig[:n, None, None] * hk[None]
    rb = prop.tf_batch_propagate(H, None, None, dt, batch_size=n)
    assert np.abs(np.asarray(rb)[g["dU_slice_index"][:6]] - g["dUs"][:6]).max() < 1e-12


@pytest.mark.parametrize("solver", ["rk4", "rk38", "rk5", "tsit5"])
@pytest.mark.parametrize("step", ["schrodinger", "von_neumann", "lindblad"])
def test_ode_vs_oracle(prop, solver, step):
    wl = workloads.make_workload(4, B=2, N=40)
    D = wl.D
    psi = np.zeros((D, 1), complex)
    psi[1, 0] = 1.0
    init = psi if step == "schrodinger" else psi @ psi.conj().T
    col = wl.col_ops if step == "lindblad" else None
    out = prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, init, solver, step, col_ops=col)
    for b in range(wl.B):
        ref = o.ode_solver_arrays(wl.h0, wl.hks, wl.signals[b], wl.ts, init, solver, step, col=col)
        assert np.abs(np.asarray(out[b]) - ref["states"]).max() < 1e-11
