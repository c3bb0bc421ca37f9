"""Round-2 GPU parity tests (through the C ABI): the register-resident kernel of c3p_regd.hip, the remaining
entry points of SURVEY 8a rows a10 / a15, the reference's fidelity known answers and ODE invariants, and wider
full-size spot parity."""
import os

import numpy as np
import pytest

from c3_amd import _lib, workloads
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


def _rand_herm(rng, D, scale):
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    return scale * (a + a.conj().T) / 2


# --------------------------------------------------------------------------
# c3p_regd.hip: Dm = 49, 65, 81
# --------------------------------------------------------------------------


@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 5), (2, 41), (5, 70), (300, 9)])
def test_regd_lindblad_81_vs_oracle(prop, B, N):
    """propagation.py:551-585 at cfg4's operators: one slice, first-slice / chain paths, several time segments
    (B < 256), more chains than workgroups (B = 300)."""
    from c3_amd import _lib

    wl = workloads.make_workload(4, B=B, N=N)
    ph = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
    r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=ph)
    assert _lib.last_kernel() == "mfma"
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=wl.fr_phase)
    assert fro_max(r["U"], ref) < TOL


def test_regd_lindblad_partials_and_old_kernel(prop):
    """dUs of the new kernel vs the oracle's per-slice superoperators, and U vs the arena kernel it replaces."""
    wl = workloads.make_workload(4, B=2, N=6)
    r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, want_dUs=True)
    d = o.tf_propagation_lind(wl.h0, wl.hks, wl.col_ops, wl.signals[1], wl.dt)
    assert np.abs(np.asarray(r["dUs"][1]) - d).max() < 1e-13
    _lib.set_option("no_regd", "1")
    try:
        old = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    finally:
        _lib.set_option("no_regd", None)
    assert fro_max(r["U"], old["U"]) < 1e-12


def test_regd_lindblad_49(prop):
    """D = 7 -> 49 x 49 superoperators (n = 3 instance): a 7-level system with one collapse operator."""
    rng = np.random.default_rng(11)
    D, B, N, K = 7, 3, 21, 2
    h0 = _rand_herm(rng, D, 0.05)
    hks = np.stack([_rand_herm(rng, D, 0.02) for _ in range(K)])
    col = np.stack([0.03 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(2)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    r = prop.propagate_batch(h0, hks, sig, 1.0, col_ops=col, lindbladian=True, want_dUs=True)
    ref = o.propagate_batch(h0, hks, sig, 1.0, col_ops=col, lindbladian=True)
    assert fro_max(r["U"], ref) < TOL
    assert np.abs(np.asarray(r["dUs"][0]) - o.tf_propagation_lind(h0, hks, col, sig[0], 1.0)).max() < 1e-13


@pytest.mark.parametrize("D", [49, 65, 81])
@pytest.mark.parametrize("scale", [0.02, 0.2])
def test_regd_unitary_dimensions(prop, D, scale):
    """Unitary mode of the same kernel, complex Hermitian operators, norms with 0 .. 3 squarings, per-sample operators."""
    rng = np.random.default_rng(D)
    B, N, K = 3, 17, 2
    h0 = np.stack([_rand_herm(rng, D, scale) for _ in range(B)])  # per-sample drift
    hks = np.stack([_rand_herm(rng, D, scale / 2) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    r = prop.propagate_batch(h0, hks, sig, 1.0, want_dUs=True)
    for b in range(B):
        ref = o.pwc_arrays(h0[b], hks, sig[b], 1.0)
        assert np.linalg.norm(np.asarray(r["U"][b]) - ref["U"]) < TOL
        assert np.abs(np.asarray(r["dUs"][b]) - ref["dUs"]).max() < 1e-12
    U = np.asarray(r["U"])
    assert np.abs(U @ U.conj().transpose(0, 2, 1) - np.eye(D)).max() < 1e-11


def test_regd_full_size_cfg4_properties(prop):
    """BASELINE cfg4 at full size (B = 512, N = 1000): trace preservation of every superoperator, split / repeat
    invariance, spot parity on 4 samples."""
    import torch

    wl = workloads.make_workload(4)
    assert wl.B == 512 and wl.N == 1000
    dev = "cuda:0"
    a = [torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals, wl.col_ops)]
    U = prop.propagate_batch(a[0], a[1], a[2], wl.dt, col_ops=a[3], lindbladian=True)["U"]
    Uh = U.cpu().numpy()
    D = wl.D
    # trace preservation: vec(I)^T S = vec(I)^T for a trace-preserving map in the row-major vec convention
    vI = np.eye(D).ravel()
    assert np.abs(np.einsum("i,bij->bj", vI, Uh) - vI).max() < 1e-10
    again = prop.propagate_batch(a[0], a[1], a[2], wl.dt, col_ops=a[3], lindbladian=True)["U"].cpu().numpy()
    assert np.array_equal(again, Uh)
    idx = [0, 171, 340, 511]
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals[idx], wl.dt, col_ops=wl.col_ops, lindbladian=True)
    assert fro_max(Uh[idx], ref) < TOL
    # time split: U = U2 U1 over the two halves of the slices (8 samples)
    s = wl.signals[:8]
    U1 = np.asarray(prop.propagate_batch(wl.h0, wl.hks, s[:, :, :500], wl.dt, col_ops=wl.col_ops, lindbladian=True)["U"])
    U2 = np.asarray(prop.propagate_batch(wl.h0, wl.hks, s[:, :, 500:], wl.dt, col_ops=wl.col_ops, lindbladian=True)["U"])
    assert fro_max(U2 @ U1, Uh[:8]) < 1e-11


# --------------------------------------------------------------------------
# the legacy / variant entry points, on the device
# --------------------------------------------------------------------------


def test_a15_single_slice_and_legacy_loop(prop):
    """tf_dU_of_t (propagation.py:349-379), tf_dU_of_t_lind (:382-423), tf_propagation (:518-548)."""
    wl = workloads.make_workload(2, B=1, N=12)
    c = wl.signals[0][:, 3]
    got = prop.tf_dU_of_t(wl.h0, wl.hks, c, wl.dt)
    assert np.abs(np.asarray(got) - o.tf_dU_of_t(wl.h0, wl.hks, c, wl.dt)).max() < 1e-13
    wl4 = workloads.make_workload(4, B=1, N=4)
    c4 = wl4.signals[0][:, 1]
    gotl = prop.tf_dU_of_t_lind(wl4.h0, wl4.hks, wl4.col_ops, c4, wl4.dt)
    assert np.abs(np.asarray(gotl) - o.tf_dU_of_t_lind(wl4.h0, wl4.hks, wl4.col_ops, c4, wl4.dt)).max() < 1e-13
    lst = prop.tf_propagation(wl.h0, wl.hks, wl.signals[0], wl.dt)
    ref = o.tf_propagation(wl.h0, wl.hks, wl.signals[0], wl.dt)
    assert isinstance(lst, list) and len(lst) == len(ref) == 12
    assert max(np.abs(np.asarray(a) - b).max() for a, b in zip(lst, ref)) < 1e-13
    assert "tf_propagation" in prop.unitary_provider


def test_a15_pwc_trott_drift(prop):
    """propagation.py:443-457: dU0 expm(-i Ht dt) (dU0 + [H0,Ht] dt^2 / 2) with the reference's eigh / v.T form."""
    wl = workloads.make_workload(2, B=1, N=4)
    c = wl.signals[0][:, 2].reshape(-1, 1, 1)
    got = np.asarray(prop.pwc_trott_drift(wl.h0, wl.hks, c, wl.dt))
    ref = o.pwc_trott_drift(wl.h0, wl.hks, c, wl.dt)
    assert np.abs(got - ref).max() < 1e-12


def test_a15_evaluate_sequences(prop):
    """propagation.py:588-627: left-multiplied gate sequences, the empty sequence, repeated gates."""
    wl = workloads.make_workload(2, B=3, N=30)
    U = np.asarray(prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt)["U"])
    gates = {"a": U[0], "b": U[1], "c": U[2]}
    seqs = [["a"], ["a", "b"], ["c", "a", "b", "a"], [], ["b"] * 7]
    got = prop.evaluate_sequences(gates, seqs)
    ref = o.evaluate_sequences(gates, seqs)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert np.abs(np.asarray(g) - np.asarray(r)).max() < 1e-12
    assert np.abs(np.asarray(got[1]) - U[1] @ U[0]).max() < 1e-12  # later gate on the left


# --------------------------------------------------------------------------
# dephasing channel, positive path (experiment.py:510-522, model.py:597-639)
# --------------------------------------------------------------------------


class _Awg:
    def __init__(self, amp):
        self.amp = amp

    def get_average_amp(self):
        return self.amp, self.amp * 10


class _PMap:
    def __init__(self, model, generator, instructions):
        self.model, self.generator, self.instructions = model, generator, instructions


def test_dephasing_channel_positive_path(prop):
    from c3_amd.experiment import Experiment

    N = 40
    T = N * 1e-11
    m = workloads.ChipModel((3, 3), (5e9, 5.6e9), (-210e6, -240e6), {(0, 1): 20e6}, {"d1": 0, "d2": 1}, t1=(27e-6, 23e-6), t2star=(39e-6, 31e-6))
    m.set_lindbladian(True)
    m.set_FR(True)
    wl = workloads.make_workload(2, B=1, N=N)
    ts = (np.arange(N) + 0.5) * 1e-11
    sig = {"g": {"d1": {"values": wl.signals[0, 0], "ts": ts}, "d2": {"values": wl.signals[0, 1], "ts": ts}}}
    gen = workloads.SignalSource(sig)
    gen.devices = {"awg": _Awg(0.37)}
    g = workloads.Gate("g", 0.0, T, ["d1", "d2"], carrier_freqs={"d1": 2 * np.pi * 5.05e9, "d2": 2 * np.pi * 5.65e9}, framechanges={"d1": 0.3, "d2": -0.2})
    exp = Experiment(_PMap(m, gen, {"g": g}), sim_res=100e9)
    plain = exp.compute_propagators()["g"]
    m.dephasing_strength = 1.1e9  # p = T * amp * strength = 0.16 per line
    out = exp.compute_propagators()["g"]
    assert out.shape == (81, 81) and np.linalg.norm(out - plain) > 1e-3
    # independent restatement: oracle pwc + frame rotation + the oracle's dephasing channel
    ref = o.pwc(m, gen, g, None, None)["U"]
    ph = np.exp(1j * m.frame_rotation_phases(T, g.carrier_freqs, g.framechanges))
    ref = np.kron(ph, ph.conj())[:, None] * ref
    nums = [m.number_operator("d1"), m.number_operator("d2")]
    ch = o.dephasing_channel(nums, [0.37, 0.37], T, 1.1e9, 9)
    assert np.linalg.norm(out - ch @ ref) < TOL
    m.dephasing_strength = 1e12  # p > 1 -> the reference's ValueError (model.py:631-636)
    with pytest.raises(ValueError, match="outside"):
        exp.compute_propagators()


def test_experiment_batch_with_excitation_cut(prop):
    """compute_propagators_batch returns full-dimension propagators when max_excitations is set, like pwc."""
    from c3_amd.experiment import Experiment

    N = 32
    T = N * 1e-11
    m = workloads.ChipModel((3, 3), (5e9, 5.6e9), (-210e6, -240e6), {(0, 1): 20e6}, {"d1": 0, "d2": 1})
    m.set_max_excitations(2)
    m.set_FR(True)
    wl = workloads.make_workload(2, B=4, N=N)
    ts = (np.arange(N) + 0.5) * 1e-11
    gen = workloads.SignalSource({"g": {"d1": {"values": wl.signals[0, 0], "ts": ts}, "d2": {"values": wl.signals[0, 1], "ts": ts}}})
    g = workloads.Gate("g", 0.0, T, ["d1", "d2"], carrier_freqs={"d1": 2 * np.pi * 5.05e9, "d2": 2 * np.pi * 5.65e9}, framechanges={"d1": 0.1, "d2": 0.2})
    exp = Experiment(_PMap(m, gen, {"g": g}), sim_res=100e9)
    U = exp.compute_propagators_batch("g", wl.signals)
    assert U.shape == (4, 9, 9)
    one = exp.compute_propagators()["g"]  # serial route: pwc blows up, the adapter applies FR on the full space
    assert np.linalg.norm(U[0] - one) < TOL


# --------------------------------------------------------------------------
# Fidelity known answers held by the reference (test/test_fidelities.py:23-140) on c3p_gate_overlap
# --------------------------------------------------------------------------

X = np.array([[0, -1j], [-1j, 0]], dtype=np.complex128)  # GATES["rxp"] (c3/libraries/constants.py:54)
Y = np.array([[0, -1], [1, 0]], dtype=np.complex128)     # GATES["ryp"] (:57)
Id = np.eye(2, dtype=np.complex128)
_LEAK0 = np.array([[0 + 0j, 1, 0], [1, 0, 0], [0, 0, 0]])
_LEAK = np.array([[0 + 0j, 1, 0], [1, 0, 0], [0, 0, 34345j]])


def test_reference_fidelity_known_answers(prop):
    from c3_amd import fidelities as F

    assert abs(F.unitary_infid(X, X, dims=[2])) < 1e-12                                        # test_unitary_infid_1
    assert F.unitary_infid(X, Y, dims=[2]) == 1                                               # _2 (exact in the reference)
    XI = np.kron(X, Id)
    assert abs(F.unitary_infid(XI, XI, index=[0, 1], dims=[2, 2])) < 1e-12                     # _3
    assert abs(F.unitary_infid(X, XI, index=[0], dims=[2, 2])) < 1e-12                         # projection
    assert abs(F.unitary_infid(X, np.kron(Id, X), index=[1], dims=[2, 2])) < 1e-12             # projection_2
    assert abs(F.unitary_infid(ideal=X, actual=_LEAK0, index=[0], dims=[3])) < 1e-12           # projection_3
    assert abs(F.unitary_infid(ideal=X, actual=_LEAK, index=[0], dims=[3])) < 1e-12            # projection_4
    assert abs(F.unitary_infid(ideal=X, actual=np.kron(_LEAK, Id), index=[0], dims=[3, 2])) < 1e-12   # projection_5
    assert abs(F.average_infid(X, X)) < 1e-12                                                  # test_average_infid_1
    assert abs(F.average_infid(X, Y) - 2.0 / 3) < 1e-12                                        # _2
    assert abs(F.average_infid(X, XI, index=[0], dims=[2, 2])) < 1e-12
    assert abs(F.average_infid(X, np.kron(Id, X), index=[1], dims=[2, 2])) < 1e-12
    assert abs(F.average_infid(ideal=X, actual=_LEAK0, index=[0], dims=[3])) < 1e-12
    assert abs(F.average_infid(ideal=X, actual=_LEAK, index=[0], dims=[3])) < 1e-12
    assert abs(F.average_infid(ideal=X, actual=np.kron(_LEAK, Id), index=[0], dims=[3, 2])) < 1e-12
    # a batch mixes them all in one device call
    batch = np.stack([np.kron(_LEAK, Id), np.kron(_LEAK0, Id)])
    assert np.abs(np.asarray(F.unitary_infid(X, batch, index=[0], dims=[3, 2]))).max() < 1e-12


# --------------------------------------------------------------------------
# The reference's ODE invariant checks (test/test_two_qubits.py:228-251) through c3p_ode_solve, on the inputs
# the reference's own fixture stores (tests/golden/two_qubit.npz)
# --------------------------------------------------------------------------


def test_ode_invariants_on_reference_inputs(prop, golden_dir):
    g = np.load(golden_dir + "/two_qubit.npz")
    sig = np.stack([g["sig_d1"], g["sig_d2"]])[None]
    hks = np.stack([g["hk_d1"], g["hk_d2"]])
    dt = float(g["ts"][1] - g["ts"][0])
    psi = np.zeros((4, 1), dtype=np.complex128)
    psi[0, 0] = 1.0
    st = np.asarray(prop.ode_solve_batch(g["hdrift"], hks, sig, dt, psi, "rk4", "schrodinger"))[0]
    assert st.shape == (700, 4, 1)
    assert abs(np.linalg.norm(st[-1]) - 1) < 0.5e-2                      # decimal=2 in the reference
    rho = np.asarray(prop.ode_solve_batch(g["hdrift"], hks, sig, dt, psi @ psi.conj().T, "rk4", "von_neumann"))[0]
    assert abs(np.trace(rho[-1]) - 1) < 0.5e-6                           # decimal=6
    m = workloads.ChipModel((2, 2), (5e9, 5.6e9), (0, 0), {(0, 1): 20e6}, {"d1": 0, "d2": 1}, t1=(20e-6, 20e-6), t2star=(40e-6, 40e-6))
    rl = np.asarray(prop.ode_solve_batch(g["hdrift"], hks, sig, dt, psi @ psi.conj().T, "rk4", "lindblad", col_ops=np.asarray(m.col_ops)))[0]
    assert abs(np.trace(rl[-1]) - 1) < 0.5e-6
    # and the values themselves against the oracle's solver (same tableau)
    ref = o.ode_solver_arrays(g["hdrift"], hks, sig[0], g["ts"], psi, "rk4", "schrodinger")["states"]
    assert np.abs(st - ref).max() < 1e-11
    # bad operator shapes are refused (one operator set per call)
    from c3_amd._lib import C3PropError

    with pytest.raises(C3PropError, match="C3:Error"):
        prop.ode_solve_batch(np.stack([g["hdrift"]] * 2), hks, sig, dt, psi)
    with pytest.raises(C3PropError, match="C3:Error"):
        prop.ode_solve_batch(g["hdrift"], hks[:1], sig, dt, psi)


# --------------------------------------------------------------------------
# Full-size spot parity on >= 32 samples (cfg1-3)
# --------------------------------------------------------------------------


@pytest.mark.parametrize("cfg,B", [(1, 256), (2, 256), (3, 512)])
def test_full_size_spot_parity_32_samples(prop, cfg, B):
    import torch

    wl = workloads.make_workload(cfg, B=B)
    dev = "cuda:0"
    U = prop.propagate_batch(torch.as_tensor(wl.h0, device=dev), torch.as_tensor(wl.hks, device=dev), torch.as_tensor(wl.signals, device=dev),
                             wl.dt, fr_phase=torch.as_tensor(wl.fr_phase, device=dev))["U"].cpu().numpy()
    idx = np.unique(np.linspace(0, B - 1, 32).astype(int))
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals[idx], wl.dt, fr_phase=wl.fr_phase[idx])
    assert len(idx) >= 32 and fro_max(U[idx], ref) < TOL


# --------------------------------------------------------------------------
# Calls from several streams of one device share the per-device workspace: they must be ordered, not racing
# --------------------------------------------------------------------------


def test_two_streams_share_the_workspace_safely(prop):
    import torch

    dev = torch.device("cuda:0")
    wa = workloads.make_workload(2, B=64, N=400)           # small-D kernel, fused combine (arrival counters, partials)
    wb = workloads.make_workload(3, B=24, N=120)           # mid-D kernel (tables, segment products)
    ta = [torch.as_tensor(x, device=dev) for x in (wa.h0, wa.hks, wa.signals)]
    tb = [torch.as_tensor(x, device=dev) for x in (wb.h0, wb.hks, wb.signals)]
    ref_a = prop.propagate_batch(ta[0], ta[1], ta[2], wa.dt)["U"].clone()
    ref_b = prop.propagate_batch(tb[0], tb[1], tb[2], wb.dt)["U"].clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    for rep in range(6):  # interleaved enqueues, no host synchronisation in between
        with torch.cuda.stream(s1):
            ua = prop.propagate_batch(ta[0], ta[1], ta[2], wa.dt)["U"]
        with torch.cuda.stream(s2):
            ub = prop.propagate_batch(tb[0], tb[1], tb[2], wb.dt)["U"]
        outs.append((ua, ub))
    torch.cuda.synchronize()
    for ua, ub in outs:
        assert torch.equal(ua, ref_a) and torch.equal(ub, ref_b)


# --------------------------------------------------------------------------
# c3p_tiled.hip: matrices in HBM, one batched MFMA GEMM per product (Dm >= 93; supplied generators at Dm >= 41)
# --------------------------------------------------------------------------


@pytest.mark.parametrize("D,B,N,K,scale", [(93, 2, 4, 2, 0.01), (100, 3, 7, 2, 0.02), (128, 2, 5, 1, 0.05), (200, 2, 3, 2, 0.01), (257, 1, 2, 1, 0.004)])
def test_tiled_unitary_dimensions(prop, D, B, N, K, scale):
    """Beyond the on-chip kernels (and beyond the old Dm <= 256 cap): padded / unpadded tile edges, frame phases, dUs."""
    from c3_amd import _lib

    rng = np.random.default_rng(D)
    h0 = _rand_herm(rng, D, scale)
    hks = np.stack([_rand_herm(rng, D, scale / 2) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ph = rng.uniform(0, 6, size=(B, D))
    r = prop.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph, want_dUs=True)
    assert _lib.last_kernel() == "mfma"
    ref = o.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph)
    assert fro_max(r["U"], ref) < TOL
    assert np.abs(np.asarray(r["dUs"][B - 1]) - o.tf_propagation_vectorized(h0, hks, sig[B - 1], 1.0)).max() < 1e-12


def test_tiled_per_sample_operators_and_chunks(prop):
    """Per-sample drift Hamiltonians (one table set per sample)."""
    rng = np.random.default_rng(3)
    D, B, N, K = 96, 3, 4, 1
    h0 = np.stack([_rand_herm(rng, D, 0.03) for _ in range(B)])
    hks = np.stack([_rand_herm(rng, D, 0.01) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    r = prop.propagate_batch(h0, hks, sig, 1.0)
    for b in range(B):
        assert np.linalg.norm(np.asarray(r["U"][b]) - o.pwc_arrays(h0[b], hks, sig[b], 1.0)["U"]) < TOL


@pytest.mark.parametrize("D", [45, 120])
def test_tiled_supplied_generators(prop, D):
    """Branch B of pwc (propagation.py:295-308) above the mid-D kernel."""
    rng = np.random.default_rng(D)
    H = np.stack([np.stack([_rand_herm(rng, D, 0.03) for _ in range(5)]) for _ in range(2)])
    r = prop.propagate_batch(H, None, None, 1.0, want_dUs=True)
    for b in range(2):
        U = np.eye(D, dtype=complex)
        for n in range(5):
            E = o.expm(-1j * H[b, n])
            assert np.abs(np.asarray(r["dUs"][b][n]) - E).max() < 1e-12
            U = E @ U
        assert np.linalg.norm(np.asarray(r["U"][b]) - U) < TOL


def test_tiled_lindblad_three_qutrits_729(prop):
    """The reference's own 'takes way too long' case (test/test_tunable_coupler.py:406-418): three qutrits with
    collapse operators, 729 x 729 superoperators -- two slices against the oracle, trace preservation."""
    B, N, K = 2, 2, 2
    wl = workloads.make_workload(3, B=B, N=N)
    a = workloads.annihilators((3, 3, 3))
    T = workloads.dressing_transform(workloads.bare_drift((3, 3, 3), (5.0e9, 5.6e9, 6.2e9), (-210e6, -240e6, -235e6), {(0, 1): 20e6, (0, 2): 20e6, (1, 2): 20e6}))
    col = np.stack([workloads.dress(workloads.qubit_collapse_op(a[q], (27e-6, 23e-6, 25e-6)[q], (39e-6, 31e-6, 35e-6)[q]), T) for q in range(3)])
    r = prop.propagate_batch(wl.h0, wl.hks[:K], wl.signals[:, :K], wl.dt, col_ops=col, lindbladian=True)
    U = np.asarray(r["U"])
    assert U.shape == (B, 729, 729)
    ref = o.propagate_batch(wl.h0, wl.hks[:K], wl.signals[:, :K], wl.dt, col_ops=col, lindbladian=True)
    assert fro_max(U, ref) < TOL
    vI = np.eye(27).ravel()
    assert np.abs(np.einsum("i,bij->bj", vI, U) - vI).max() < 1e-11


def test_tiled_lindblad_per_slice_hamiltonians(prop):
    """Lindblad propagation with supplied per-slice Hamiltonians (controllability off + lindbladian) at D = 7."""
    rng = np.random.default_rng(2)
    D, N = 7, 6
    H = np.stack([_rand_herm(rng, D, 0.05) for _ in range(N)])
    col = np.stack([0.04 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    r = prop.propagate_batch(H, None, None, 1.0, col_ops=col, lindbladian=True)
    ref = o.pwc_arrays(H, None, None, 1.0, col_ops=col, lindbladian=True)["U"]
    assert np.linalg.norm(np.asarray(r["U"][0]) - ref) < TOL


# ---------------------------------------------------------------------------------------------------------------
# real-Hamiltonian backward sweep (smalld_grad_real_kernel): reverse mode through cos Y / sin Y
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("amp", [3e9, 3e10, 2e11, 8e11])
@pytest.mark.parametrize("D", list(range(2, 13)))
def test_vjp_real_hamiltonian_small_dims(prop, D, amp, monkeypatch):
    """every template instance, 0 .. 3 squarings and the hand-over to the general sweep above that; against the
    FD-pinned oracle gradient and against the general (complex T18 pair) sweep on the same inputs"""
    if D == 2 and amp > 5e11:
        pytest.skip("two levels at |H| dt ~ 10 rad: the gradient itself is at roundoff level")
    rng = np.random.default_rng(100 * D + int(np.log10(amp)))
    B, K, N = 3, 2, 29

    def sym():
        a = rng.normal(size=(D, D))
        return (a + a.T) / 2

    h0 = np.stack([sym() * amp for _ in range(B)]).astype(np.complex128)
    hks = np.stack([sym() for _ in range(K)]).astype(np.complex128)
    sig = rng.normal(size=(B, K, N)) * 2e9
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar))
    _lib.set_option("no_real_grad", "1")
    g0 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar))
    _lib.set_option("no_real_grad", None)
    # (at 8e11 most dimensions are past three squarings of theta_16 and are handed to the general sweep, whose error
    # grows with the number of squarings: |H| dt ~ 10 rad per slice is two orders above any C3 model)
    tol = 1e-10 if amp < 5e11 else 2e-9
    for b in range(B):
        want = o.pwc_signal_gradient(h0[b], hks, sig[b], 1e-11, Ubar[b])
        assert np.abs(g[b] - want).max() < tol * np.abs(want).max()
    assert np.abs(g - g0).max() < tol * np.abs(g0).max()


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [2e10, 1e11, 2.5e11, 8e11])
@pytest.mark.parametrize("D", [13, 16, 17, 20, 24, 27, 28, 32, 33, 36, 40])
def test_vjp_real_hamiltonian_mid_dims(prop, D, amp, monkeypatch):
    """midd_grad_real_kernel: every geometry class, 0 .. 2 squarings and the hand-over to the general sweep; shared and
    per-sample Hamiltonians, frame-rotation row phases; against the oracle and the general sweep"""
    from c3_amd import _lib

    rng = np.random.default_rng(1000 * D + int(np.log10(amp)))
    B, K, N = 2, 3, 11

    def sym():
        a = rng.normal(size=(D, D))
        return (a + a.T) / 2

    per_sample = D % 2 == 0
    h0 = (np.stack([sym() for _ in range(B)]) if per_sample else sym()).astype(np.complex128) * (amp / np.sqrt(D))
    hks = np.stack([sym() for _ in range(K)]).astype(np.complex128)
    sig = rng.normal(size=(B, K, N)) * 2e9
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    ph = rng.uniform(0, 6, size=(B, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    assert _lib.last_kernel() == "mfma"
    _lib.set_option("no_real_grad", "1")
    g0 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    _lib.set_option("no_real_grad", None)
    tol = 1e-10 if amp < 5e11 else 2e-9
    for b in range(B):
        want = o.pwc_signal_gradient(h0[b] if per_sample else h0, hks, sig[b], 1e-11, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < tol * np.abs(want).max()
    assert np.abs(g - g0).max() < tol * np.abs(g0).max()


@pytest.mark.gpu
def test_vjp_hermitian_check_is_cached_per_tensor_version(prop):
    """device operators are reduced once; an in-place write invalidates the pass"""
    import torch
    from c3_amd.propagation import C3PropError

    w = workloads.make_workload(2, B=2, N=16)
    dev = "cuda:0"
    h0, hks = torch.as_tensor(w.h0, device=dev), torch.as_tensor(w.hks, device=dev)
    sig = torch.as_tensor(w.signals, device=dev)
    Ubar = torch.randn(2, w.D, w.D, dtype=torch.complex128, device=dev)
    g1 = prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar)
    g2 = prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar)
    assert torch.equal(g1, g2)
    h0[0, 1] += 1e9j  # no longer Hermitian
    with pytest.raises(C3PropError, match="Hermitian"):
        prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar)


@pytest.mark.gpu
@pytest.mark.parametrize("D,K", [(14, 4), (27, 5), (36, 6)])
def test_mid_dims_real_path_more_than_three_control_lines(prop, D, K):
    """the real cos / sin instance keeps three control tables in registers and reads further ones per slice"""
    rng = np.random.default_rng(77 * D + K)
    B, N = 3, 23

    def sym():
        a = rng.normal(size=(D, D))
        return (a + a.T) / 2

    h0 = sym().astype(np.complex128) * (2e11 / np.sqrt(D))
    hks = np.stack([sym() for _ in range(K)]).astype(np.complex128)
    sig = rng.normal(size=(B, K, N)) * 2e9
    ph = rng.uniform(0, 6, size=(B, D))
    out = prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph, want_dUs=True)
    U, dUs = np.asarray(out["U"]), np.asarray(out["dUs"])
    ref = o.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)
    assert np.abs(U - ref).max() < 1e-10
    for b in range(B):
        d = o.tf_propagation_vectorized(h0, hks, sig[b], 1e-11)
        assert np.abs(dUs[b] - d).max() < 1e-11
    # and its gradient (general sweep: the real sweep keeps K <= 3)
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    for b in range(B):
        want = o.pwc_signal_gradient(h0, hks, sig[b], 1e-11, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


# ---------------------------------------------------------------------------------------------------------------
# small-D kernel: one workgroup per sample (S / 4 waves, combine through LDS) against one-wave workgroups (ticket)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B,N", [(2, 256, 1000), (2, 512, 640), (2, 1024, 256), (2, 300, 500), (2, 2, 257), (1, 256, 200)])
def test_smalld_workgroup_per_sample_mode(prop, cfg, B, N, monkeypatch):
    """with equal segments (C3P_MW_SKEW=500) the same arithmetic in the same order: bit-identical to the one-wave-workgroup
    mode; with the default uneven segments (older waves take the longer ones) equal to rounding; spot parity against the oracle"""
    w = workloads.make_workload(cfg, B=B, N=N)
    U = np.asarray(prop.propagate_batch(w.h0, w.hks, w.signals, w.dt, fr_phase=w.fr_phase)["U"])
    _lib.set_option("mw_skew", "500")
    Ue = np.asarray(prop.propagate_batch(w.h0, w.hks, w.signals, w.dt, fr_phase=w.fr_phase)["U"])
    _lib.set_option("mw_skew", None)
    _lib.set_option("no_mw", "1")
    U0 = np.asarray(prop.propagate_batch(w.h0, w.hks, w.signals, w.dt, fr_phase=w.fr_phase)["U"])
    _lib.set_option("no_mw", None)
    assert np.array_equal(Ue, U0)
    assert np.abs(U - U0).max() < 1e-12
    idx = np.unique(np.linspace(0, B - 1, 4).astype(int))
    ref = o.propagate_batch(w.h0, w.hks, w.signals[idx], w.dt, fr_phase=w.fr_phase[idx])
    assert np.abs(U[idx] - ref).max() < 1e-11


@pytest.mark.gpu
def test_smalld_workgroup_per_sample_mode_per_sample_and_lindblad(prop, monkeypatch):
    """per-sample Hamiltonians (tables per workgroup) and a 9 x 9 Lindblad superoperator batch through the same mode"""
    rng = np.random.default_rng(5)
    D, B, K, N = 7, 256, 2, 320

    def sym():
        a = rng.normal(size=(D, D))
        return (a + a.T) / 2

    h0 = np.stack([sym() * 4e10 for _ in range(B)]).astype(np.complex128)
    hks = np.stack([sym() for _ in range(K)]).astype(np.complex128)
    sig = rng.normal(size=(B, K, N)) * 2e9
    U = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11)["U"])
    _lib.set_option("no_mw", "1")
    U0 = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11)["U"])
    _lib.set_option("no_mw", None)
    assert np.abs(U - U0).max() < 1e-12
    for b in (0, 100, 255):
        assert np.abs(U[b] - o.propagate_batch(h0[b], hks, sig[b : b + 1], 1e-11)[0]).max() < 1e-11
    # Lindblad, D = 3 (Dm = 9)
    D = 3
    a = rng.normal(size=(D, D))
    h0 = ((a + a.T) / 2 * 3e10).astype(np.complex128)
    hks = np.stack([sym()[:D, :D] for _ in range(K)]).astype(np.complex128)
    col = (rng.normal(size=(2, D, D)) + 1j * rng.normal(size=(2, D, D))) * 3e3
    sig = rng.normal(size=(B, K, N)) * 2e9
    U = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11, col_ops=col, lindbladian=True)["U"])
    _lib.set_option("no_mw", "1")
    U0 = np.asarray(prop.propagate_batch(h0, hks, sig, 1e-11, col_ops=col, lindbladian=True)["U"])
    _lib.set_option("no_mw", None)
    assert np.abs(U - U0).max() < 1e-12
    ref = o.propagate_batch(h0, hks, sig[:2], 1e-11, col_ops=col, lindbladian=True)
    assert np.abs(U[:2] - ref).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("N", [32, 33, 47, 128, 129, 1001])
def test_smalld_uneven_segments_edge_lengths(prop, N):
    """uneven segment lengths of the workgroup-per-sample mode at awkward slice counts (every short chain keeps >= 1 slice)"""
    w = workloads.make_workload(2, B=256, N=N)
    U = np.asarray(prop.propagate_batch(w.h0, w.hks, w.signals, w.dt, fr_phase=w.fr_phase)["U"])
    idx = [0, 127, 255]
    ref = o.propagate_batch(w.h0, w.hks, w.signals[idx], w.dt, fr_phase=w.fr_phase[idx])
    assert np.abs(U[idx] - ref).max() < 1e-11
