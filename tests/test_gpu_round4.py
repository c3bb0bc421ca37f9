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


# --------------------------------------------------------------------------
# Lindblad control gradient at D = 7, 8, 9 (49 x 49 .. 81 x 81 superoperators): on-chip backward sweep in the Hermitian basis
# (c3p_regrg.hip) -- the taped tf_propagation_lind (propagation.py:551-585 under optimizers/optimizer.py:206-216)
# --------------------------------------------------------------------------


def _lind_case(D, B, K, N, C, seed, per_sample=False, hscale=0.8, cscale=0.25):
    rng = np.random.default_rng(seed)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    nb = B if per_sample else 1
    h0 = np.stack([herm(hscale) for _ in range(nb)])
    hks = np.stack([np.stack([herm(0.5 * hscale / 0.8) for _ in range(K)]) for _ in range(nb)])
    if not per_sample:
        h0, hks = h0[0], hks[0]
    col = np.stack([cscale * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Dm = D * D
    Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm))
    return h0, hks, col, sig, Ubar, ph


@pytest.mark.parametrize("D,N,B,K,C,per_sample,segments,dt", [
    (9, 9, 2, 2, 2, False, None, 0.1),   # one segment per sample, no squarings at the highest degree
    (9, 17, 3, 2, 1, True, 4, 0.3),      # several segments (scan: prefix, suffix, fold), per-sample operators, squarings
    (8, 12, 2, 3, 2, False, 3, 0.25),    # 64 x 64 in the zero-padded 65 class
    (7, 16, 2, 1, 1, True, 2, 0.3),      # 49 x 49
])
def test_lindblad_vjp_hermitian_basis_sweep(prop, D, N, B, K, C, per_sample, segments, dt):
    """against the FD-pinned oracle gradient (one Frechet derivative per (k, n)) and against the tiled sweep on the same inputs"""
    h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, C, 1000 + 10 * D + N, per_sample)
    _lib.set_option("segments", segments)
    try:
        g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
        assert _lib.last_kernel() == "mfma"
        _lib.set_option("tiled_grad", "1")
        gt = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
    finally:
        _lib.set_option("segments", None)
        _lib.set_option("tiled_grad", None)
    assert np.abs(g - gt).max() < 1e-10 * np.abs(gt).max()
    for b in (0, B - 1):
        hb0 = h0[b] if per_sample else h0
        hbk = hks[b] if per_sample else hks
        want = o.pwc_lindblad_signal_gradient(hb0, hbk, col, sig[b], dt, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


def test_lindblad_vjp_hermitian_basis_degrees_chunks_and_fallback(prop):
    """every Taylor degree of the pair evaluation (8, 12, 16, 20: different Horner depths and squaring counts) gives the same
    gradient; sample chunks reproduce the single-chunk result; a non-Hermitian Hamiltonian (complex generator in the Hermitian
    basis) falls back to the tiled sweep and still matches the oracle."""
    D, B, K, N = 9, 5, 2, 11
    h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, 1, 77, per_sample=True)
    ref = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
    for deg in (8, 12, 16, 20):
        with _lib.options(regr_grad_degree=deg):
            g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
        assert np.abs(g - ref).max() < 2e-11 * np.abs(ref).max(), deg
    with _lib.options(grad_chunk=2, segments=3):
        many = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
    assert np.abs(many - ref).max() < 1e-11 * np.abs(ref).max()
    want = o.pwc_lindblad_signal_gradient(h0[3], hks[3], col, sig[3], 0.2, Ubar[3], ph[3])
    assert np.abs(ref[3] - want).max() < 1e-10 * np.abs(want).max()
    # lossy (non-Hermitian) drift: the generator is complex in the Hermitian basis
    hn = h0[0] - 0.05j * np.diag(np.arange(D))
    gn = np.asarray(prop.propagate_batch_lindblad_vjp(hn, hks[0], sig[:2], 0.2, col, Ubar[:2], fr_phase=ph[:2]))
    want = o.pwc_lindblad_signal_gradient(hn, hks[0], col, sig[1], 0.2, Ubar[1], ph[1])
    assert np.abs(gn[1] - want).max() < 1e-10 * np.abs(want).max()


def test_lindblad_vjp_hermitian_basis_strong_dissipation_and_long_chain(prop):
    """a strongly damped two-qutrit chain (nothing is inverted in the sweep) over 70 slices in 5 segments: signal chunks of 32
    slices, ragged segment lengths; against the tiled sweep (itself oracle-checked above) and the oracle on one sample."""
    D, B, K, N = 9, 2, 2, 70
    rng = np.random.default_rng(3)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0, hks = herm(1.0), np.stack([herm(0.5) for _ in range(K)])
    a = np.kron(np.diag(np.sqrt(np.arange(1, 3)), 1), np.eye(3))
    col = np.stack([0.8 * a, 0.5 * np.kron(np.eye(3), np.diag(np.arange(3.0)))]).astype(complex)
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Ubar = rng.normal(size=(B, 81, 81)) + 1j * rng.normal(size=(B, 81, 81))
    with _lib.options(segments=5):
        g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.25, col, Ubar))
    with _lib.options(tiled_grad=1):
        gt = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.25, col, Ubar))
    assert np.abs(g - gt).max() < 1e-10 * np.abs(gt).max()


# --------------------------------------------------------------------------
# core + border form of the real small-D path (8 + 1 split at D = 9, 4 + 1 at D = 5): c3p_smalld.hip SMat / GMat
# (propagation.py:426-440 + tf_utils.py:144-193 on real Hamiltonians)
# --------------------------------------------------------------------------


@pytest.mark.parametrize("D,B,N,amp,mw", [
    (9, 256, 96, 1.0, True),     # cfg2's shape: workgroup per sample, degree-16 variant
    (9, 8, 250, 1.0, True),      # uneven segments (eight waves per sample)
    (9, 7, 37, 1.0, False),      # one-wave workgroups + ticket, ragged chains (7 x S not a multiple of four)
    (9, 5, 40, 2.6, False),      # stronger drive: degree-18 variant
    (9, 4, 30, 14.0, False),     # squarings
    (5, 6, 64, 1.0, True),       # 4 + 1
    (5, 3, 21, 9.0, False),
])
def test_smalld_core_plus_border_form(prop, D, B, N, amp, mw):
    """the same propagators from the core + border form (default) and from the padded tiles (no_split81), both against the
    oracle on real symmetric Hamiltonians with frame-rotation phases"""
    rng = np.random.default_rng(D * 100 + N)

    def rsym(scale):
        a = rng.normal(size=(D, D))
        return (scale * (a + a.T) / 2).astype(complex)

    h0 = rsym(6e10)
    hks = np.stack([rsym(1.0), rsym(1.0)])
    sig = rng.normal(size=(B, 2, N)) * 2e9 * amp
    ph = rng.uniform(0, 6, size=(B, D))
    dt = 1e-11
    with _lib.options(no_mw=None if mw else 1):
        a = np.asarray(prop.propagate_batch(h0, hks, sig, dt, fr_phase=ph)["U"])
        assert _lib.last_kernel() == "smalld"
        with _lib.options(no_split81=1):
            b = np.asarray(prop.propagate_batch(h0, hks, sig, dt, fr_phase=ph)["U"])
    ref = o.propagate_batch(h0, hks, sig[:3], dt, fr_phase=ph[:3])
    for i in range(min(3, B)):
        assert np.linalg.norm(a[i] - ref[i]) < 1e-10
        assert np.linalg.norm(b[i] - ref[i]) < 1e-10
    assert np.abs(a - b).max() < 1e-11
    assert np.abs(a @ a.conj().transpose(0, 2, 1) - np.eye(D)).max() < 1e-10


def test_lindblad_taped_evaluation_matches_the_untaped_pair(prop):
    """c3p_pwc_lindblad_taped + c3p_pwc_lindblad_vjp_taped (one forward pass, caller-owned tape) against c3p_pwc_lindblad +
    c3p_pwc_lindblad_vjp and the oracle; per-sample operators, several segments; a non-Hermitian Hamiltonian is refused."""
    import torch

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)
    for D, B, K, N, per_sample, segs in ((9, 3, 2, 14, True, 3), (7, 2, 1, 9, False, None), (8, 2, 2, 8, False, 2)):
        h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, 1, 500 + D, per_sample)
        with _lib.options(segments=segs):
            r = prop.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), 0.2, t(col), fr_phase=t(ph))
            g = r["tape"].vjp(t(Ubar)).cpu().numpy()
            g2 = r["tape"].vjp(t(2.0 * Ubar)).cpu().numpy()  # the tape is reusable
        U = r["U"].cpu().numpy()
        U0 = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
        g0 = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
        assert np.abs(U - U0).max() < 1e-11
        assert np.abs(g - g0).max() < 1e-11 * np.abs(g0).max()
        assert np.abs(g2 - 2.0 * g0).max() < 2e-11 * np.abs(g0).max()
        ref = o.propagate_batch(h0[0] if per_sample else h0, hks[0] if per_sample else hks, sig[:1], 0.2, col_ops=col, lindbladian=True)
        assert np.linalg.norm(U[0] - np.exp(1j * ph[0])[:, None] * ref[0]) < 1e-10
    assert prop.lindblad_tape_supported(4, 2, 10, 3) and not prop.lindblad_tape_supported(4, 2, 10, 5)
    hn = h0 - 0.05j * np.diag(np.arange(D))
    with pytest.raises(Exception, match="Hermitian"):
        prop.propagate_batch_lindblad_taped(t(hn), t(hks), t(sig), 0.2, t(col))


def test_lindblad_taped_evaluation_small_superoperators(prop):
    """The taped pair at D = 2, 3 (4 x 4 / 9 x 9 superoperators on the small-D kernels: the tape holds the tables, the segment
    products and the slice propagators of the forward half of the general sweep), against the untaped pair and the oracle's
    finite differences: per-sample operators, non-Hermitian Hamiltonians (no restriction here), one and many segments."""
    import torch

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)
    for D, B, K, N, per_sample, lossy in ((3, 5, 2, 41, False, False), (3, 4, 1, 17, True, True), (2, 7, 3, 30, False, False), (3, 64, 1, 200, False, False)):
        h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, 1, 900 + D + N, per_sample)
        if lossy:
            h0 = h0 - 0.05j * np.diag(np.arange(D))
        r = prop.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), 0.2, t(col), fr_phase=t(ph))
        assert _lib.last_kernel() == "smalld"
        g = r["tape"].vjp(t(Ubar)).cpu().numpy()
        g2 = r["tape"].vjp(t(-3.0 * Ubar)).cpu().numpy()
        U = r["U"].cpu().numpy()
        g1 = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))  # the untaped entry, same kernels
        with _lib.options(no_smallr=1):  # the complex small-D kernels (Hermitian cases run in real arithmetic by default)
            U0 = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
            g0 = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
        assert np.abs(U - U0).max() < 2e-12
        assert np.abs(g - g0).max() < 1e-11 * np.abs(g0).max()
        assert np.abs(g1 - g0).max() < 1e-11 * np.abs(g0).max()
        assert np.abs(g2 + 3.0 * g0).max() < 4e-11 * np.abs(g0).max()
    # finite differences of the oracle's forward pass on the last small case but one
    D, B, K, N = 3, 2, 2, 12
    h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, 1, 77, False)
    r = prop.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), 0.2, t(col))
    g = r["tape"].vjp(t(Ubar)).cpu().numpy()
    f = lambda sg: float(np.real(np.sum(np.conj(Ubar) * o.propagate_batch(h0, hks, sg, 0.2, col_ops=col, lindbladian=True))))
    for (b, k, n) in ((0, 0, 3), (1, 1, 11), (1, 0, 0)):
        e = np.zeros_like(sig)
        e[b, k, n] = 1e-6
        fd = (f(sig + e) - f(sig - e)) / 2e-6
        assert abs(fd - g[b, k, n]) < 1e-6 * max(1.0, abs(fd))


def test_goal_run_with_grad_open_system_taped_equals_untaped(prop):
    """optimal_control.goal_run_with_grad(col_ops=...) at two qutrits: the taped evaluation (default) and the untaped pair give
    the same goal and gradients; a lossy Hamiltonian takes the untaped pair by itself."""
    from c3_amd import optimal_control as oc, signals as sg

    w = workloads.make_workload(4, B=1, N=8)
    T, awg_res, sim_res = 0.6e-9, 20e9, 100e9
    TWO_PI = 2 * np.pi
    B = 3
    rng = np.random.default_rng(2)
    chans = [[dict(shape="gaussian_nonorm", amp=rng.uniform(0.2, 0.5, size=B), xy_angle=0.2 * k, freq_offset=-53e6 * TWO_PI, delta=-0.6, t_final=T, sigma=T / 4, drag=True)]
             for k in range(2)]
    env, shapes = sg.pack_components(chans, B=B)
    carrier = np.tile(np.array([[5.05e9 * TWO_PI, 1e9 * TWO_PI], [5.65e9 * TWO_PI, 1e9 * TWO_PI]]), (B, 1, 1))
    ph = rng.uniform(0, 6, size=(B, 81))
    ideal = np.kron(np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2), np.eye(2))
    kw = dict(fr_phase=ph, fid_func="lindbladian_unitary_infid", col_ops=w.col_ops)
    a = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], **kw)
    b = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], fused=False, **kw)
    for key in ("goal", "grad_env", "grad_carrier", "grad_fr_phase"):
        x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
        assert np.abs(x - y).max() <= 1e-10 * max(np.abs(y).max(), 1e-30), key
    hn = w.h0 - 1e6j * np.diag(np.arange(9.0))
    c = oc.goal_run_with_grad(hn, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], **kw)
    assert np.isfinite(c["goal"].cpu().numpy()).all()


@pytest.mark.parametrize("D,B,N,segments", [
    (7, 2, 75, 1),    # one chain of 75 slices: the control amplitudes are staged in chunks of 32 slices (descending)
    (7, 40, 16, 8),   # 320 chains on 256 workgroups: a workgroup sweeps a second chain (arena and LDS reuse)
])
def test_lindblad_vjp_hermitian_basis_long_chains_and_many_chains(prop, D, B, N, segments):
    h0, hks, col, sig, Ubar, ph = _lind_case(D, B, 2, N, 1, 31 * D + N, per_sample=False)
    with _lib.options(segments=segments):
        g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.15, col, Ubar, fr_phase=ph))
    with _lib.options(tiled_grad=1):
        gt = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.15, col, Ubar, fr_phase=ph))
    assert np.abs(g - gt).max() < 1e-10 * np.abs(gt).max()
    want = o.pwc_lindblad_signal_gradient(h0, hks, col, sig[B - 1], 0.15, Ubar[B - 1], ph[B - 1])
    assert np.abs(g[B - 1] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("D,K,real", [(3, 1, True), (9, 2, True), (9, 2, False), (12, 3, False)])
def test_ode_trajectory_of_small_batches_in_time_segments(prop, D, K, real):
    """ode_solver (every state of the trajectory, propagation.py:687-752) for a SMALL batch: segment maps -> state at the start
    of every segment -> the pieces integrated side by side.  Same numbers as the direct integration and as the oracle's
    sequential solver; uneven last segment, all four solvers, per-sample initial states."""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from test_gpu_round3 import _ode_problem

    B, N = 5, 203
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 1900 + D)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "rk38", "rk5", "tsit5"):
        seg = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert _lib.last_kernel() == "ode_row"
        with _lib.options(ode_no_seg=1):
            direct = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert seg.shape == direct.shape
        assert np.abs(seg - direct).max() < 1e-12 * max(1.0, np.abs(direct).max())
        for b in (0, B - 1):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger")["states"]
            assert np.abs(seg[b] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("D,K,real,B", [(20, 2, False, 3), (27, 3, True, 2), (36, 3, True, 5), (40, 1, False, 4)])
def test_ode_trajectory_time_segments_mid_dimensions(prop, D, K, real, B):
    """the same at 17 <= D <= 48: segment maps on the matrix-core kernel, the pieces on the lane-row kernel of c3p_ode_rowq.hip
    (one time segment per wavefront; odd batches leave half-empty wavefronts); against the direct integration and the oracle"""
    import sys

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from test_gpu_round3 import _ode_problem

    N = 150
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 2900 + D)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "tsit5"):
        seg = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        with _lib.options(ode_no_seg=1):
            direct = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert np.abs(seg - direct).max() < 1e-11 * max(1.0, np.abs(direct).max())
        ref = o.ode_solver_arrays(h0, hks, sig[B - 1], ts, psi[B - 1], solver, "schrodinger")["states"]
        assert np.abs(seg[B - 1] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_randomised_parity_sweep_of_the_round4_entry_points(lib):
    """tools/fuzz_r04.py for a few seconds: fused goal vs three calls, Hermitian-basis Lindblad sweep and taped pair vs the tiled
    sweep, core + border form vs padded tiles, segmented ODE trajectories vs direct -- random shapes (4 minutes of it:
    55 000 cases, worst deviation 4e-12)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_r04.py"), "--seconds", "8", "--seed", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------------
# ODE solvers: more than four control lines and supplied per-sample-index Hamiltonians on the lane-row kernels
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,K", [(3, 5), (5, 6), (9, 7), (12, 5), (16, 6)])
def test_ode_row_more_than_four_control_lines(prop, D, K):
    """propagation.py:687-752 with K > 4 at D <= 16: H(t_n) is assembled for every sample index and the lane-row kernels
    interpolate it between samples (the same linear interpolation as tf_utils.py:521-559, after the sum instead of before):
    vector states; rho-valued states read the operator rows of the extra lines from LDS.  Trajectory and final state, against
    the workgroup kernel (ode_wg) and the oracle."""
    from c3_amd import _lib
    import oracle.c3_oracle as o

    rng = np.random.default_rng(100 + D)
    B, N = 6, 37
    herm = lambda s, real=False: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D)))
    h0, hks = herm(0.3), np.stack([herm(0.2, k % 2 == 0) for k in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ts = (np.arange(N) + 0.5) * 0.05
    dt = ts[1] - ts[0]
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    rho = np.einsum("bik,bjk->bij", psi, psi.conj())
    col = np.stack([0.2 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    cases = [("rk4", "schrodinger", psi, None), ("tsit5", "schrodinger", psi, None), ("rk4", "von_neumann", rho, None), ("rk5", "lindblad", rho, col)]
    for solver, step, init, c in cases:
        for fin in (False, True):
            got = np.asarray(prop.ode_solve_batch(h0, hks, sig, dt, init, solver, step, col_ops=c, final_only=fin))
            # (rho-valued states: the operator rows of the lines beyond four sit in LDS)
            # (von Neumann without collapse operators at this batch: lane rows AND workgroup kernel are launched, the device picks)
            assert _lib.last_kernel() == ("ode_row_or_wg" if (step == "von_neumann" and c is None) else "ode_row"), (solver, step)
            with _lib.options(ode_wg=1):
                ref = np.asarray(prop.ode_solve_batch(h0, hks, sig, dt, init, solver, step, col_ops=c, final_only=fin))
                assert _lib.last_kernel() == "ode_wg"
            assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (solver, step, fin)
        full = np.asarray(prop.ode_solve_batch(h0, hks, sig, dt, init, solver, step, col_ops=c))
        orc = o.ode_solver_arrays(h0, hks, sig[2], ts, init[2], solver, step, col=None if c is None else list(c))["states"]
        assert np.abs(full[2] - orc).max() < 1e-11 * max(1.0, np.abs(orc).max()), (solver, step)


@pytest.mark.parametrize("D", [2, 5, 9, 14])
def test_rk4_unitary_supplied_hamiltonians_on_lane_rows(prop, D):
    """Branch B of get_hs_of_t_ts (propagation.py:164-204): per-sample-index Hamiltonians through c3p_rk4_unitary -- the
    lane-row kernel reads row i of the sample nearest to every stage position (round 1-3: the workgroup kernel)."""
    from c3_amd import _lib
    import oracle.c3_oracle as o

    rng = np.random.default_rng(7 + D)
    Ns = 41
    Hs = rng.normal(size=(Ns, D, D)) + 1j * rng.normal(size=(Ns, D, D))
    Hs = 0.4 * (Hs + Hs.conj().transpose(0, 2, 1))
    dt = 0.07
    U, dUs = prop._rk4_unitary_device(Hs=Hs, dt=dt)
    assert _lib.last_kernel() == "ode_row"
    ref = o.rk4_unitary_arrays(Hs, dt, D)
    assert np.abs(np.asarray(U) - ref["U"]).max() < 1e-12
    assert np.abs(np.asarray(dUs) - ref["dUs"]).max() < 1e-12
    with _lib.options(ode_wg=1):
        U2, _ = prop._rk4_unitary_device(Hs=Hs, dt=dt)
        assert _lib.last_kernel() == "ode_wg"
    assert np.abs(np.asarray(U) - np.asarray(U2)).max() < 1e-13


@pytest.mark.parametrize("D,K,real", [(18, 5, False), (20, 6, True), (27, 6, False), (27, 8, True), (33, 5, True)])
def test_ode_rowq_five_to_eight_control_lines(prop, D, K, real):
    """Vector states at 17 <= D <= 48 with 5 .. 8 control lines: the second instance of ode_vecq_kernel (control amplitudes of
    eight lines per chunk; operators in LDS as before, as long as they fit) against the workgroup kernel and the oracle."""
    from c3_amd import _lib
    import oracle.c3_oracle as o

    rng = np.random.default_rng(300 + D)
    B, N = 5, 33
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))).astype(complex)
    h0, hks = herm(0.2), np.stack([herm(0.1) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ts = (np.arange(N) + 0.5) * 0.05
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "tsit5"):
        got = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        # (operators + the seven stage slots of tsit5 may exceed the LDS at the larger K D^2: the workgroup kernel takes those)
        assert _lib.last_kernel() == "ode_row" or (solver == "tsit5" and (1 + K) * D * D > 4000)
        with _lib.options(ode_wg=1):
            ref = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
            assert _lib.last_kernel() == "ode_wg"
        assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), solver
        orc = o.ode_solver_arrays(h0, hks, sig[1], ts, psi[1], solver, "schrodinger")["states"]
        assert np.abs(got[1] - orc).max() < 1e-11 * max(1.0, np.abs(orc).max()), solver


def test_c_client_of_the_abi_on_the_device(lib, tmp_path):
    """tests/checks/abi_client.c (strict C99, dlopen of the in-tree library, no torch in the process): one batch through
    c3p_pwc_unitary with host pointers; U is unitary and equals the ordered product of the slice propagators it returns."""
    import os
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "abi_client")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "checks", "abi_client.c"), "-o", exe, "-ldl", "-lm"], check=True)
    out = subprocess.run([exe, _lib.LIB_PATH, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ABI_CLIENT_GPU_OK" in out.stdout


# ------------------------------------------------------------------------------------------------------------------
# Lindblad chains of one qubit / qutrit in real arithmetic in the Hermitian basis (c3p_smallr.hip, flag C3P_HERMITIAN_H)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,B,K,N,per_sample,amp", [(3, 5, 2, 41, False, 1.0), (3, 4, 1, 17, True, 1.0), (2, 7, 3, 30, False, 1.0), (3, 64, 1, 200, False, 1.0),
                                                     (2, 1, 1, 1, False, 1.0), (3, 3, 2, 9, False, 25.0), (3, 9, 0, 12, False, 1.0), (3, 256, 2, 1000, False, 0.3)])
def test_lindblad_small_real_path(prop, D, B, K, N, per_sample, amp):
    """propagation.py:551-585 at D = 2, 3 with Hermitian Hamiltonians: the real kernels (propagate_batch sets C3P_HERMITIAN_H
    after checking the arrays) against the complex small-D kernels (option no_smallr) and the oracle; per-sample operators,
    frame phases, no control line, large generators (squarings), one slice; a non-Hermitian Hamiltonian keeps the complex path."""
    h0, hks, col, sig, _, ph = _lind_case(D, B, max(K, 1), N, 2 if D == 3 else 1, 4000 + 10 * D + N, per_sample, hscale=0.8 * amp)
    if K == 0:
        hks, sig = hks[..., :0, :, :], sig[:, :0, :]
    got = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
    assert _lib.last_kernel() == "smalld"
    with _lib.options(no_smallr=1):
        ref = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
    assert np.abs(got - ref).max() < 2e-12 * max(1.0, np.abs(ref).max())
    for b in sorted({0, B - 1}):
        orc = o.propagate_batch(h0[b] if per_sample else h0, hks[b] if per_sample else hks, sig[b : b + 1], 0.2, col_ops=col, lindbladian=True)[0]
        assert np.linalg.norm(got[b] - np.exp(1j * ph[b])[:, None] * orc) < 1e-10 * max(1.0, np.linalg.norm(orc))
    # trace preservation: vec(1)^T S = vec(1)^T (frame phases are row phases: take them out first)
    vecI = np.eye(D).reshape(-1)
    S = np.exp(-1j * ph)[:, :, None] * got
    assert np.abs(np.einsum("i,bij->bj", vecI, S) - vecI).max() < 1e-10


def test_lindblad_small_real_path_is_not_taken_for_lossy_hamiltonians(prop):
    import torch

    D, B, K, N = 3, 4, 2, 25
    h0, hks, col, sig, _, ph = _lind_case(D, B, K, N, 1, 4242)
    hn = h0 - 0.05j * np.diag(np.arange(D))
    x = np.asarray(prop.propagate_batch(hn, hks, sig, 0.2, col_ops=col, lindbladian=True)["U"])
    with _lib.options(no_smallr=1):
        y = np.asarray(prop.propagate_batch(hn, hks, sig, 0.2, col_ops=col, lindbladian=True)["U"])
    assert np.array_equal(x, y)  # the same kernels: the flag was not set
    orc = o.propagate_batch(hn, hks, sig[:1], 0.2, col_ops=col, lindbladian=True)[0]
    assert np.linalg.norm(x[0] - orc) < 1e-10
    # device tensors: the Hermitian check is cached per tensor object
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    h0d, hkd, sgd, cold = t(h0), t(hks), t(sig), t(col)
    a = prop.propagate_batch(h0d, hkd, sgd, 0.2, col_ops=cold, lindbladian=True)["U"].cpu().numpy()
    b = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True)["U"])
    assert np.abs(a - b).max() < 1e-13
    # the pre-bound call checks once and keeps the flag; a lossy Hamiltonian keeps the complex kernels
    bp = prop.BatchPropagator(h0d, hkd, sgd, 0.2, col_ops=cold)
    assert bp.flags & _lib.HERMITIAN_H
    assert np.abs(bp.run().cpu().numpy() - b).max() < 1e-13
    assert not (prop.BatchPropagator(t(hn), hkd, sgd, 0.2, col_ops=cold).flags & _lib.HERMITIAN_H)


@pytest.mark.parametrize("B,K,N,per_sample,amp", [(5, 2, 37, False, 1.0), (4, 1, 16, True, 1.0), (64, 2, 200, False, 0.5), (3, 2, 9, False, 12.0), (1, 1, 1, False, 1.0)])
def test_lindblad_two_qubits_real_path(prop, B, K, N, per_sample, amp):
    """Two qubits (D = 4, 16 x 16 superoperators: the size of the reference's own Lindblad golden test) with Hermitian
    Hamiltonians on the real Hermitian-basis kernels: forward against the complex mid-D kernels (no_smallr) and the oracle,
    trace preservation; the gradient (untaped and taped) against the complex sweep and finite differences of the oracle."""
    import torch

    D = 4
    h0, hks, col, sig, Ubar, ph = _lind_case(D, B, K, N, 2, 5000 + N, per_sample, hscale=0.8 * amp)
    got = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
    g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
    with _lib.options(no_smallr=1):
        ref = np.asarray(prop.propagate_batch(h0, hks, sig, 0.2, col_ops=col, lindbladian=True, fr_phase=ph)["U"])
        g0 = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.2, col, Ubar, fr_phase=ph))
    assert np.abs(got - ref).max() < 5e-12 * max(1.0, np.abs(ref).max())
    assert np.abs(g - g0).max() < 1e-10 * np.abs(g0).max()
    orc = o.propagate_batch(h0[0] if per_sample else h0, hks[0] if per_sample else hks, sig[:1], 0.2, col_ops=col, lindbladian=True)[0]
    assert np.linalg.norm(got[0] - np.exp(1j * ph[0])[:, None] * orc) < 1e-10 * max(1.0, np.linalg.norm(orc))
    vecI = np.eye(D).reshape(-1)
    assert np.abs(np.einsum("i,bij->bj", vecI, np.exp(-1j * ph)[:, :, None] * got) - vecI).max() < 1e-10
    # the taped pair
    assert prop.lindblad_tape_supported(B, K, N, D) == (N >= 4)  # (the tape is laid out for a multiple of four segments)
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    if N >= 4:
        r = prop.propagate_batch_lindblad_taped(t(h0), t(hks), t(sig), 0.2, t(col), fr_phase=t(ph))
        assert np.abs(r["U"].cpu().numpy() - ref).max() < 5e-12 * max(1.0, np.abs(ref).max())
        gt = r["tape"].vjp(t(Ubar)).cpu().numpy()
        assert np.abs(gt - g0).max() < 1e-10 * np.abs(g0).max()
    if N <= 16 and not per_sample:
        f = lambda sg: float(np.real(np.sum(np.conj(Ubar) * np.exp(1j * ph)[:, :, None] * o.propagate_batch(h0, hks, sg, 0.2, col_ops=col, lindbladian=True))))
        for (b, k, n) in ((0, 0, 0), (B - 1, K - 1, N - 1)):
            e = np.zeros_like(sig)
            e[b, k, n] = 1e-6
            fd = (f(sig + e) - f(sig - e)) / 2e-6
            assert abs(fd - g[b, k, n]) < 2e-6 * max(1.0, abs(fd))
    # a lossy Hamiltonian: the complex kernels, and the taped call says so
    hn = h0 - 0.03j * np.diag(np.arange(D))
    x = np.asarray(prop.propagate_batch(hn, hks, sig, 0.2, col_ops=col, lindbladian=True)["U"])
    with _lib.options(no_smallr=1):
        y = np.asarray(prop.propagate_batch(hn, hks, sig, 0.2, col_ops=col, lindbladian=True)["U"])
    assert np.array_equal(x, y)
    if N >= 4:
        with pytest.raises(Exception, match="Hermitian"):
            prop.propagate_batch_lindblad_taped(t(hn), t(hks), t(sig), 0.2, t(col))
