"""Round-4 GPU parity tests (through the C ABI): the fused goal + gradient entry (one optimiser evaluation from one pass
over the chains) and the on-chip Lindblad backward sweep in the Hermitian basis."""
import numpy as np
import pytest

from c3_amd import _lib, workloads
from oracle import c3_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import _lib, propagation

    _lib.require_gpu()
    return propagation


def _herm(rng, D, scale):
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    return scale * (a + a.conj().T) / 2


def _goal_case(D, dims, index, B, K, N, seed, real):
    rng = np.random.default_rng(seed)
    h0 = _herm(rng, D, 4e11 / D)
    hks = np.stack([_herm(rng, D, 1.0) for _ in range(K)])
    if real:
        h0, hks = h0.real.astype(complex), hks.real.astype(complex)
    sig = rng.normal(size=(B, K, N)) * 2e9
    ph = rng.uniform(0, 6, size=(B, D))
    L = 2 ** len(index)
    q, _ = np.linalg.qr(rng.normal(size=(L, L)) + 1j * rng.normal(size=(L, L)))
    return h0, hks, sig, ph, q


@pytest.mark.parametrize("D,dims,index,B,N,real,kernel", [
    (9, [3, 3], [0, 1], 5, 60, True, "smalld"),      # cfg2's shape: real-Hamiltonian sweep
    (9, [3, 3], [1], 3, 41, False, "smalld"),        # complex Hamiltonians, one-qubit goal on a two-qutrit space
    (3, [3], [0], 4, 33, True, "smalld"),
    (12, [3, 4], [0, 1], 2, 25, False, "smalld"),
    (27, [3, 3, 3], [0, 2], 3, 23, True, "mfma"),    # mid-D sweep, LDS scan
    (36, [3, 3, 4], [0, 1, 2], 2, 17, False, "mfma"),  # mid-D sweep, global-scratch scan
    (48, [6, 8], [0, 1], 2, 9, False, "generic_global"),  # VALU sweep
])
@pytest.mark.parametrize("kind", ["unitary", "average"])
def test_fused_goal_vjp_matches_three_call_form_and_oracle(prop, D, dims, index, B, N, real, kernel, kind):
    """c3p_pwc_unitary_goal_vjp against c3p_pwc_unitary -> fidelities.*_cotangent -> c3p_pwc_unitary_vjp (the taped goal of
    optimizers/optimizer.py:206-216 over optimalcontrol.py:200-228) and against the oracle's infidelity."""
    from c3_amd import fidelities as fid

    K = 2
    h0, hks, sig, ph, G = _goal_case(D, dims, index, B, K, N, 100 + D + len(index), real)
    dt = 1e-11
    r = prop.propagate_batch_goal_vjp(h0, hks, sig, dt, G, index, dims, kind=kind, fr_phase=ph)
    assert _lib.last_kernel() == kernel
    U = np.asarray(prop.propagate_batch(h0, hks, sig, dt, fr_phase=ph)["U"])
    cot = fid.unitary_infid_cotangent if kind == "unitary" else fid.average_infid_cotangent
    Ubar, goal = cot(G, U, index, dims)
    g3 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, dt, Ubar, fr_phase=ph))
    gph3 = -(np.conj(Ubar) * U).sum(axis=-1).imag
    ref_U = o.propagate_batch(h0, hks, sig, dt, fr_phase=ph)
    ofid = o.unitary_infid if kind == "unitary" else o.average_infid
    for b in range(B):
        assert abs(r["goal"][b] - ofid(G, ref_U[b], index=index, dims=dims)) < 1e-11
        assert np.linalg.norm(np.asarray(r["U"])[b] - ref_U[b]) < 1e-10
    assert np.abs(np.asarray(r["goal"]) - np.asarray(goal)).max() < 1e-12
    scale = np.abs(g3).max()
    assert np.abs(np.asarray(r["grad_signals"]) - g3).max() < 1e-11 * scale
    assert np.abs(np.asarray(r["grad_fr_phase"]) - gph3).max() < 1e-11 * max(np.abs(gph3).max(), 1e-3)


def test_fused_goal_vjp_on_device_tensors_and_errors(prop):
    import torch

    h0, hks, sig, ph, G = _goal_case(9, [3, 3], [0, 1], 4, 2, 50, 7, True)
    host = prop.propagate_batch_goal_vjp(h0, hks, sig, 1e-11, G, [0, 1], [3, 3], fr_phase=ph)
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)
    r = prop.propagate_batch_goal_vjp(t(h0), t(hks), t(sig), 1e-11, t(G), [0, 1], [3, 3], fr_phase=t(ph), want_U=False)
    torch.cuda.synchronize()
    assert r["U"] is None
    assert np.array_equal(r["goal"].cpu().numpy(), host["goal"])
    assert np.array_equal(r["grad_signals"].cpu().numpy(), host["grad_signals"])
    with pytest.raises(Exception, match="C3:Error"):  # the tiled sweep's shapes keep the three-call form
        prop.propagate_batch_goal_vjp(np.eye(81, dtype=complex), np.eye(81, dtype=complex)[None], np.zeros((1, 1, 4)), 1e-11, np.eye(2), [0],
                                      [81], check_hermitian=False)
    with pytest.raises(Exception, match="ideal gate"):
        prop.propagate_batch_goal_vjp(h0, hks, sig, 1e-11, np.eye(2), [0, 1], [3, 3])


def test_goal_run_with_grad_fused_equals_unfused(prop):
    """optimal_control.goal_run_with_grad: the fused evaluation (default) and the three-call form give the same goal and
    the same gradients w.r.t. envelope rows, carriers and frame-rotation phases."""
    from c3_amd import optimal_control as oc, signals as sg

    w = workloads.make_workload(2, B=1, N=8)
    T, awg_res, sim_res = 7e-9, 2e9, 100e9
    TWO_PI = 2 * np.pi
    B = 6
    rng = np.random.default_rng(11)
    chans = [[dict(shape="gaussian_nonorm", amp=rng.uniform(0.2, 0.5, size=B), xy_angle=0.2, freq_offset=-53e6 * TWO_PI, delta=-0.6, t_final=T, sigma=T / 4, drag=True)],
             [dict(shape="flattop_risefall", amp=0.1, xy_angle=-0.4, freq_offset=10e6 * TWO_PI, delta=0.3, t_final=T, risefall=0.8e-9, drag=True)]]
    env, shapes = sg.pack_components(chans, B=B)
    carrier = np.tile(np.array([[5.05e9 * TWO_PI, 1e9 * TWO_PI], [5.65e9 * TWO_PI, 1e9 * TWO_PI]]), (B, 1, 1))
    phases = rng.uniform(0, 6, size=(B, 9))
    ideal = np.kron(np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2), np.eye(2))
    for fid_func in ("unitary_infid", "average_infid"):
        a = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], fr_phase=phases, fid_func=fid_func)
        b = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], fr_phase=phases, fid_func=fid_func, fused=False)
        for key in ("goal", "grad_env", "grad_carrier", "grad_fr_phase", "U"):
            x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
            assert np.abs(x - y).max() <= 1e-11 * max(np.abs(y).max(), 1e-30), (fid_func, key)
