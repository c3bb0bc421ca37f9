"""Gradient of the propagators w.r.t. the control samples."""
import os

import numpy as np
import pytest

from c3_amd import _lib
from c3_amd.workloads import make_workload
from oracle import c3_oracle as o


@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import propagation, _lib

    _lib.require_gpu()
    return propagation


def test_oracle_gradient_matches_finite_differences():
    w = make_workload(1, B=1, N=5)
    rng = np.random.default_rng(3)
    D = w.D
    Ubar = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    g = o.pwc_signal_gradient(w.h0, w.hks, w.signals[0], w.dt, Ubar, w.fr_phase[0])

    def loss(sig):
        U = o.propagate_batch(w.h0, w.hks, sig[None], w.dt, fr_phase=w.fr_phase[:1])[0]
        return np.real(np.vdot(Ubar, U))

    for k in range(w.K):
        for n in range(5):
            h = 2e3
            sp, sm = w.signals[0].copy(), w.signals[0].copy()
            sp[k, n] += h
            sm[k, n] -= h
            fd = (loss(sp) - loss(sm)) / (2 * h)
            assert abs(fd - g[k, n]) < 1e-6 * np.abs(g).max()


def test_oracle_infid_cotangent():
    rng = np.random.default_rng(4)
    ideal = np.kron(np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2), np.eye(2))
    U = np.linalg.qr(rng.normal(size=(9, 9)) + 1j * rng.normal(size=(9, 9)))[0]
    Ub = o.unitary_infid_cotangent(ideal, U, [0, 1], [3, 3])
    dU = rng.normal(size=(9, 9)) + 1j * rng.normal(size=(9, 9))
    f = lambda V: o.unitary_infid(ideal, V, index=[0, 1], dims=[3, 3])
    eps = 1e-6
    assert abs((f(U + eps * dU) - f(U - eps * dU)) / (2 * eps) - np.real(np.vdot(Ub, dU))) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("cfg,B,N", [(1, 3, 40), (2, 5, 64), (2, 2, 333), (3, 2, 24), (5, 1, 12)])
def test_vjp_vs_oracle(prop, cfg, B, N, generic):
    w = make_workload(cfg, B=B, N=N)
    rng = np.random.default_rng(10 + cfg)
    D = w.D
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    g = np.asarray(prop.propagate_batch_vjp(w.h0, w.hks, w.signals, w.dt, Ubar, fr_phase=w.fr_phase, force_generic=generic))
    for b in range(B):
        want = o.pwc_signal_gradient(w.h0, w.hks, w.signals[b], w.dt, Ubar[b], w.fr_phase[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
def test_vjp_device_resident_infid_gradient(prop):
    """propagate -> unitary_infid cotangent -> vjp, all on the device; checked by a directional finite
    difference of the oracle's infidelity."""
    import torch
    from c3_amd import fidelities

    w = make_workload(2, B=4, N=200)
    ideal = np.kron(np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2), np.eye(2))
    dev = "cuda:0"
    h0, hks = torch.as_tensor(w.h0, device=dev), torch.as_tensor(w.hks, device=dev)
    sig, ph = torch.as_tensor(w.signals, device=dev), torch.as_tensor(w.fr_phase, device=dev)
    U = prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)["U"]
    Ubar, infid = fidelities.unitary_infid_cotangent(ideal, U, [0, 1], [3, 3])
    g = prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar, fr_phase=ph)
    assert g.is_cuda and tuple(g.shape) == (4, 2, 200)
    g = g.cpu().numpy()
    rng = np.random.default_rng(0)
    dirn = rng.normal(size=w.signals.shape[1:])
    for b in range(2):
        def f(sig_b):
            Ub = o.propagate_batch(w.h0, w.hks, sig_b[None], w.dt, fr_phase=w.fr_phase[b : b + 1])[0]
            return o.unitary_infid(ideal, Ub, index=[0, 1], dims=[3, 3])

        assert abs(float(infid[b]) - f(w.signals[b])) < 1e-12
        eps = 1e4
        fd = (f(w.signals[b] + eps * dirn) - f(w.signals[b] - eps * dirn)) / (2 * eps)
        an = float((g[b] * dirn).sum())
        assert abs(fd - an) < 2e-6 * abs(an) + 1e-18


@pytest.mark.gpu
@pytest.mark.parametrize("D", [2, 4, 5, 7, 11, 12])
def test_vjp_small_dims(prop, D):
    """every small-D template instance of the MFMA backward kernel, per-sample Hamiltonians included"""
    rng = np.random.default_rng(D)
    B, K, N = 3, 2, 37
    def herm():
        a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
        return (a + a.conj().T) / 2
    h0 = np.stack([herm() * 3e10 for _ in range(B)])
    hks = np.stack([herm() for _ in range(K)])
    sig = rng.normal(size=(B, K, N)) * 2e9
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar))
    for b in range(B):
        want = o.pwc_signal_gradient(h0[b], hks, sig[b], 1e-11, Ubar[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [13, 16, 20, 24, 27, 32, 36, 40])
def test_vjp_mid_dims(prop, D):
    """every geometry class of the mid-D MFMA backward kernel (strong drive: squarings included)"""
    from c3_amd import _lib

    rng = np.random.default_rng(D)
    B, K, N = 2, 3, 19

    def herm():
        a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
        return (a + a.conj().T) / 2

    h0 = herm() * (6e11 / D)
    hks = np.stack([herm() for _ in range(K)])
    sig = rng.normal(size=(B, K, N)) * 2e9
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    ph = rng.uniform(0, 6, size=(B, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    assert _lib.last_kernel() == "mfma"
    for b in range(B):
        want = o.pwc_signal_gradient(h0, hks, sig[b], 1e-11, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
def test_vjp_errors(prop):
    w = make_workload(1, B=2, N=8)
    with pytest.raises(Exception, match="C3:Error"):
        prop.propagate_batch_vjp(w.h0, w.hks, w.signals, w.dt, np.zeros((1, 3, 3), complex))
    bad = w.h0.copy()
    bad[0, 1] += 1e6
    with pytest.raises(Exception, match="Hermitian"):
        prop.propagate_batch_vjp(bad, w.hks, w.signals, w.dt, np.zeros((2, 3, 3), complex))


@pytest.mark.gpu
def test_goal_run_with_grad_pulse_parameters(prop):
    """envelope rows -> goal and d goal / d (amp, xy_angle, freq_offset, delta, framechange phases) on the
    device, checked against finite differences of the ORACLE pipeline (signal oracle -> propagator oracle ->
    unitary_infid oracle)."""
    from c3_amd import optimal_control as oc, signals as sg

    w = make_workload(2, B=1, N=8)  # operators only
    T, awg_res, sim_res = 6e-9, 2e9, 100e9
    TWO_PI = 2 * np.pi
    B = 3
    rng = np.random.default_rng(5)
    amps = rng.uniform(0.2, 0.5, size=B)
    chans = [
        [dict(shape="gaussian_nonorm", amp=amps, xy_angle=0.2, freq_offset=-53e6 * TWO_PI, delta=-0.6, t_final=T, sigma=T / 4, use_t_before=True, drag=True)],
        [dict(shape="flattop_risefall", amp=0.1, xy_angle=-0.4, freq_offset=10e6 * TWO_PI, delta=0.3, t_final=T, risefall=0.8e-9, drag=True)],
    ]
    env, shapes = sg.pack_components(chans, B=B)
    carrier = np.tile(np.array([[5.05e9 * TWO_PI, 1e9 * TWO_PI], [5.65e9 * TWO_PI, 1e9 * TWO_PI]]), (B, 1, 1))
    phases = np.tile(w.fr_phase[:1] * (T / (8 * w.dt)), (B, 1))
    ideal = np.kron(np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2), np.eye(2))
    r = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], fr_phase=phases)
    goal = r["goal"].cpu().numpy()
    genv = r["grad_env"].cpu().numpy()
    gph = r["grad_fr_phase"].cpu().numpy()

    def oracle_goal(env_b, ph_b, b):
        sigs = []
        for k in range(2):
            c = {name: env_b[k, 0, slot] for name, slot in sg.ENV_SLOTS.items() if name != "flags"}
            fl = int(env_b[k, 0, sg.ENV_SLOTS["flags"]])
            c.update(shape=int(shapes[k, 0]), use_t_before=bool(fl & 1), drag=bool(fl & 2))
            sigs.append(o.generate_signal([c], carrier[b, k, 0], carrier[b, k, 1], 0.0, T, awg_res, sim_res)["values"])
        dt = o.create_ts(0.0, T, sim_res)
        U = o.propagate_batch(w.h0, w.hks, np.stack(sigs)[None], dt[1] - dt[0], fr_phase=ph_b[None])[0]
        return o.unitary_infid(ideal, U, index=[0, 1], dims=[3, 3])

    for b in range(B):
        assert abs(goal[b] - oracle_goal(env[b], phases[b], b)) < 1e-11
        for (k, name, h) in [(0, "amp", 1e-6), (0, "xy_angle", 1e-6), (0, "delta", 1e-5), (0, "freq_offset", 1e3), (1, "amp", 1e-6), (1, "delta", 1e-5)]:
            ep, em = env[b].copy(), env[b].copy()
            ep[k, 0, sg.ENV_SLOTS[name]] += h
            em[k, 0, sg.ENV_SLOTS[name]] -= h
            fd = (oracle_goal(ep, phases[b], b) - oracle_goal(em, phases[b], b)) / (2 * h)
            assert abs(fd - genv[b, k, 0, sg.ENV_SLOTS[name]]) < 2e-6 * abs(fd) + 1e-16, (b, k, name)
        i = 4
        pp, pm = phases[b].copy(), phases[b].copy()
        pp[i] += 1e-6
        pm[i] -= 1e-6
        fd = (oracle_goal(env[b], pp, b) - oracle_goal(env[b], pm, b)) / 2e-6
        assert abs(fd - gph[b, i]) < 1e-6 * abs(fd) + 1e-12
    rr = oc.robust_goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0, 1], [3, 3], fr_phase=phases)
    assert abs(float(rr["goal"]) - goal.mean()) < 1e-14
    assert np.abs(rr["grad_env"].cpu().numpy() - genv.mean(axis=0)).max() < 1e-12 * np.abs(genv).max()


@pytest.mark.gpu
def test_goal_run_with_grad_open_system(prop):
    """The same loop body through the Lindblad path (model.lindbladian: tf_propagation_lind, propagation.py:551-585, under the
    tape of optimizer.py:206-216) with the open-system goal lindbladian_unitary_infid (fidelities.py:221-249): goal against
    the oracle's literal tf_super / tf_project_to_comp(to_super) / tf_superoper_unitary_overlap chain, pulse-parameter
    gradients against finite differences of the oracle pipeline; one qutrit, 9 x 9 superoperators (matrix-core sweep)."""
    from c3_amd import _lib, fidelities as fd_, optimal_control as oc, signals as sg

    w = make_workload(1, B=1, N=8)  # one qutrit: operators only
    T, awg_res, sim_res = 7e-9, 2e9, 100e9
    TWO_PI = 2 * np.pi
    B = 2
    rng = np.random.default_rng(9)
    amps = rng.uniform(0.3, 0.5, size=B)
    chans = [[dict(shape="gaussian_nonorm", amp=amps, xy_angle=0.1, freq_offset=-50e6 * TWO_PI, delta=-0.8, t_final=T, sigma=T / 4, use_t_before=True, drag=True)]]
    env, shapes = sg.pack_components(chans, B=B)
    carrier = np.tile(np.array([[5.0e9 * TWO_PI, 1e9 * TWO_PI]]), (B, 1, 1))
    D = w.D
    a = np.diag(np.sqrt(np.arange(1, D)), 1).astype(complex)
    col = np.stack([np.sqrt(1 / 40e-9) * a, np.sqrt(0.5 / 60e-9) * 2 * a.conj().T @ a])  # T1 / T2* collapse operators (chip.py:216-242)
    p1 = rng.uniform(0, 2 * np.pi, size=(B, D))
    ph_super = (p1[:, :, None] - p1[:, None, :]).reshape(B, D * D)
    ideal = np.array([[1, -1j], [-1j, 1]]) / np.sqrt(2)
    r = oc.goal_run_with_grad(w.h0, w.hks, env, shapes, carrier, 0.0, T, awg_res, sim_res, ideal, [0], [3], fr_phase=ph_super,
                              fid_func="lindbladian_unitary_infid", col_ops=col)
    assert _lib.last_kernel() in ("smalld", "generic_lds")
    goal = r["goal"].cpu().numpy()
    genv = r["grad_env"].cpu().numpy()

    def oracle_goal(env_b, b):
        c = {name: env_b[0, 0, slot] for name, slot in sg.ENV_SLOTS.items() if name != "flags"}
        fl = int(env_b[0, 0, sg.ENV_SLOTS["flags"]])
        c.update(shape=int(shapes[0, 0]), use_t_before=bool(fl & 1), drag=bool(fl & 2))
        vals = o.generate_signal([c], carrier[b, 0, 0], carrier[b, 0, 1], 0.0, T, awg_res, sim_res)["values"]
        ts = o.create_ts(0.0, T, sim_res)
        S = o.propagate_batch(w.h0, w.hks, vals[None, None], ts[1] - ts[0], col_ops=col, lindbladian=True, fr_phase=p1[b][None])[0]
        return o.lindbladian_unitary_infid(ideal, S, index=[0], dims=[3])

    for b in range(B):
        assert abs(goal[b] - oracle_goal(env[b], b)) < 1e-11
        for (name, h) in [("amp", 1e-6), ("xy_angle", 1e-6), ("delta", 1e-5), ("freq_offset", 1e3)]:
            ep, em = env[b].copy(), env[b].copy()
            ep[0, 0, sg.ENV_SLOTS[name]] += h
            em[0, 0, sg.ENV_SLOTS[name]] -= h
            fdv = (oracle_goal(ep, b) - oracle_goal(em, b)) / (2 * h)
            assert abs(fdv - genv[b, 0, 0, sg.ENV_SLOTS[name]]) < 2e-6 * abs(fdv) + 1e-16, (b, name)
    # host-pointer form of the epilogue and its registry entry
    S = r["U"].cpu().numpy()
    assert np.abs(np.asarray(fd_.fidelities["lindbladian_unitary_infid"](ideal, S, [0], [3])) - goal).max() < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,N,generic", [(2, 40, False), (2, 40, True), (3, 16, False)])
def test_vjp_model_gradients(prop, cfg, N, generic):
    """Cotangents of the Hamiltonians themselves (model-parameter fits, modellearning.py:300-341): contraction
    of the per-slice generator cotangents, checked against directional finite differences of the oracle."""
    w = make_workload(cfg, B=2, N=N)
    rng = np.random.default_rng(40 + cfg)
    D, K = w.D, w.K
    Ubar = rng.normal(size=(2, D, D)) + 1j * rng.normal(size=(2, D, D))
    g, g0, gk = prop.propagate_batch_vjp(w.h0, w.hks, w.signals, w.dt, Ubar, fr_phase=w.fr_phase, force_generic=generic, want_model_grads=True)
    g0, gk = np.asarray(g0), np.asarray(gk)
    assert g0.shape == (2, D, D) and gk.shape == (2, K, D, D)

    def loss(h0, hks, b):
        U = o.propagate_batch(h0, hks, w.signals[b : b + 1], w.dt, fr_phase=w.fr_phase[b : b + 1])[0]
        return np.real(np.vdot(Ubar[b], U))

    def herm():
        a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
        return (a + a.conj().T) / 2

    for b in range(2):
        E = herm()
        eps = 2e-6 * np.abs(w.h0).max()
        fd = (loss(w.h0 + eps * E, w.hks, b) - loss(w.h0 - eps * E, w.hks, b)) / (2 * eps)
        an = np.real(np.vdot(g0[b], E))
        assert abs(fd - an) < 2e-6 * abs(an) + 1e-18
        k = b % K
        Ek = herm()
        epk = 2e-4 * np.abs(w.hks).max()
        hp, hm = w.hks.copy(), w.hks.copy()
        hp[k] += epk * Ek
        hm[k] -= epk * Ek
        fd = (loss(w.h0, hp, b) - loss(w.h0, hm, b)) / (2 * epk)
        an = np.real(np.vdot(gk[b, k], Ek))
        assert abs(fd - an) < 2e-6 * abs(an) + 1e-18


def test_oracle_lindblad_gradient_matches_finite_differences():
    """The Lindblad gradient oracle (one Frechet derivative per (k, n), no unitarity assumed) against central finite
    differences of the PINNED Lindblad propagator oracle (propagation.py:551-585)."""
    w = make_workload(4, B=1, N=4)
    rng = np.random.default_rng(5)
    Dm = w.D * w.D
    Ubar = rng.normal(size=(Dm, Dm)) + 1j * rng.normal(size=(Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=Dm)
    g = o.pwc_lindblad_signal_gradient(w.h0, w.hks, w.col_ops, w.signals[0], w.dt, Ubar, ph)

    def loss(sig):
        U = o.propagate_batch(w.h0, w.hks, sig[None], w.dt, col_ops=w.col_ops, lindbladian=True)[0]
        return np.real(np.vdot(Ubar, np.exp(1j * ph)[:, None] * U))

    for k in range(w.K):
        for n in range(4):
            h = 2e3
            sp, sm = w.signals[0].copy(), w.signals[0].copy()
            sp[k, n] += h
            sm[k, n] -= h
            fd = (loss(sp) - loss(sm)) / (2 * h)
            assert abs(fd - g[k, n]) < 1e-6 * np.abs(g).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,N,B,C", [((2,), 9, 3, 1), ((3,), 14, 2, 1), ((2, 2), 7, 2, 2), ((3, 3), 6, 2, 2)])
def test_lindblad_vjp_vs_oracle(prop, dims, N, B, C):
    """c3p_pwc_lindblad_vjp (forward partials in HBM + pair evaluation of T18 on the tiled GEMM) against the FD-pinned
    oracle gradient: D^2 = 4, 9, 16, 81 superoperators, frame-rotation row phases, per-sample cotangents."""
    rng = np.random.default_rng(len(dims) * 10 + N)
    D = int(np.prod(dims))
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    K = 2
    h0, hks = herm(0.8), np.stack([herm(0.5) for _ in range(K)])
    col = np.stack([0.3 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Dm = D * D
    Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm))
    dt = 0.7  # |L| dt ~ 2 - 5: squarings in the pair evaluation
    g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
    for b in range(B):
        want = o.pwc_lindblad_signal_gradient(h0, hks, col, sig[b], dt, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,B,C,per_sample", [(2, 100, 3, 1, False), (3, 64, 2, 2, True), (4, 40, 2, 1, False), (5, 17, 2, 2, False), (6, 9, 2, 1, True)])
def test_lindblad_vjp_small_superoperators_general_sweep(prop, D, N, B, C, per_sample):
    """D^2 <= 36: the sweep for general (non-unitary) generators -- several time segments per sample (prefix and left adjoint
    of every segment from the scan); matrix-core small-D kernel (D <= 3), mid-D kernel (D = 4, 5, 6), and the VALU kernels in
    LDS / global scratch as a second opinion, per-sample operators,
    moderate dissipation -- against the FD-pinned oracle and against the tiled sweep (C3P_TILED_GRAD=1) on the same inputs."""
    from c3_amd import _lib

    rng = np.random.default_rng(100 * D + N)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    K = 2
    nb = B if per_sample else 1
    h0 = np.stack([herm(0.8) for _ in range(nb)])
    hks = np.stack([np.stack([herm(0.5) for _ in range(K)]) for _ in range(nb)])
    if not per_sample:
        h0, hks = h0[0], hks[0]
    col = np.stack([0.25 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Dm = D * D
    Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm))
    dt = 0.3
    g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
    assert _lib.last_kernel() == ("smalld" if D <= 3 else "mfma")
    _lib.set_option("tiled_grad", "1")
    try:
        gt = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
        assert _lib.last_kernel() == "mfma"
    finally:
        _lib.set_option("tiled_grad", None)
    assert np.abs(g - gt).max() < 1e-10 * np.abs(gt).max()
    _lib.set_option("valu_grad", "1")  # the VALU form of the same sweep
    try:
        gv = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar, fr_phase=ph))
        assert _lib.last_kernel() == ("generic_lds" if D <= 4 else "generic_global")
    finally:
        _lib.set_option("valu_grad", None)
    assert np.abs(gv - gt).max() < 1e-10 * np.abs(gt).max()
    for b in range(B):
        want = o.pwc_lindblad_signal_gradient(h0[b] if per_sample else h0, hks[b] if per_sample else hks, col, sig[b], dt, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [2, 4])
def test_lindblad_vjp_sample_chunks(prop, D):
    """Large batches run the general-generator sweep in chunks of samples (its workspace is 2 N D^4 complex per sample):
    C3P_GRAD_CHUNK=2 on five samples with per-sample operators and frame phases must reproduce the single-chunk result."""
    rng = np.random.default_rng(D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    B, K, N, Dm = 5, 2, 24, D * D
    h0 = np.stack([herm(0.8) for _ in range(B)])
    hks = np.stack([np.stack([herm(0.5) for _ in range(K)]) for _ in range(B)])
    col = np.stack([0.25 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    ph = rng.uniform(0, 2 * np.pi, size=(B, Dm))
    one = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.3, col, Ubar, fr_phase=ph))
    _lib.set_option("grad_chunk", "2")
    try:
        many = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.3, col, Ubar, fr_phase=ph))
        _lib.set_option("valu_grad", "1")
        valu = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.3, col, Ubar, fr_phase=ph))
    finally:
        _lib.set_option("grad_chunk", None)
        _lib.set_option("valu_grad", None)
    assert np.abs(one - many).max() < 1e-12 * np.abs(one).max()
    assert np.abs(one - valu).max() < 1e-10 * np.abs(one).max()
    want = o.pwc_lindblad_signal_gradient(h0[4], hks[4], col, sig[4], 0.3, Ubar[4], ph[4])
    assert np.abs(many[4] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N", [(2, 400), (3, 300), (4, 200)])
def test_lindblad_vjp_strong_dissipation(prop, D, N):
    """Strongly damped chains (the smallest eigenvalue of the total superoperator is ~1e-32): the general-generator sweeps never
    invert a slice -- prefixes run forwards, the left adjoint backwards -- so they must stay at rounding level against the oracle."""
    rng = np.random.default_rng(5 + D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    K, B, Dm = 2, 2, D * D
    h0, hks = herm(2.0), np.stack([herm(1.0) for _ in range(K)])
    col = np.stack([np.diag(np.sqrt(np.arange(1, D)), 1).astype(complex), 0.7 * np.diag(np.arange(D)).astype(complex)]) * (1.5 if D == 2 else 1.0)
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Ubar = rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm))
    g = np.asarray(prop.propagate_batch_lindblad_vjp(h0, hks, sig, 0.25, col, Ubar))
    U = np.asarray(prop.propagate_batch(h0, hks, sig, 0.25, col_ops=col, lindbladian=True)["U"])
    assert np.abs(np.linalg.eigvals(U[0])).min() < 1e-25
    for b in range(B):
        want = o.pwc_lindblad_signal_gradient(h0, hks, col, sig[b], 0.25, Ubar[b])
        assert np.abs(g[b] - want).max() < 1e-11 * np.abs(want).max()


@pytest.mark.gpu
def test_lindblad_vjp_cfg4_operators_on_device(prop):
    """cfg4's operators (81 x 81 superoperators), device-resident tensors, against the oracle on both samples."""
    import torch

    w = make_workload(4, B=2, N=10)
    rng = np.random.default_rng(8)
    Dm = w.D * w.D
    Ubar = rng.normal(size=(2, Dm, Dm)) + 1j * rng.normal(size=(2, Dm, Dm))
    t = lambda x: torch.as_tensor(x, device="cuda:0")
    g = prop.propagate_batch_lindblad_vjp(t(w.h0), t(w.hks), t(w.signals), w.dt, t(w.col_ops), t(Ubar))
    assert g.is_cuda
    g = g.cpu().numpy()
    for b in range(2):
        want = o.pwc_lindblad_signal_gradient(w.h0, w.hks, w.col_ops, w.signals[b], w.dt, Ubar[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N", [(41, 9), (48, 12), (64, 6), (81, 5)])
def test_unitary_vjp_above_40_on_the_tiled_sweep(prop, D, N):
    """c3p_pwc_unitary_vjp beyond the on-chip sweeps (D > 40, no cap any more): the tiled backward sweep, complex
    Hermitian operators, frame rotation, norms that need squarings."""
    from c3_amd import _lib

    rng = np.random.default_rng(D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    B, K = 2, 2
    h0, hks = herm(0.25), np.stack([herm(0.15) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    ph = rng.uniform(0, 2 * np.pi, size=(B, D))
    import os

    _lib.set_option("tiled_grad", "1")  # (41 <= D <= 64 takes the VALU sweep at this batch size by default)
    try:
        g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1.0, Ubar, fr_phase=ph))
    finally:
        _lib.set_option("tiled_grad", None)
    assert _lib.last_kernel() == "mfma"
    for b in range(B):
        want = o.pwc_signal_gradient(h0, hks, sig[b], 1.0, Ubar[b], ph[b])
        assert np.abs(g[b] - want).max() < 1e-10 * np.abs(want).max()


def test_oracle_per_slice_cotangents_match_finite_differences():
    rng = np.random.default_rng(12)
    D, N = 4, 5
    herm = lambda: (lambda a: (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    Hs = np.stack([herm() for _ in range(N)])
    Ubar = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    ph = rng.uniform(0, 2 * np.pi, size=D)
    dt = 0.4
    Hb = o.pwc_per_slice_hamiltonian_cotangents(Hs, dt, Ubar, ph)

    def loss(H):
        U = o.propagate_batch(H, None, None, dt) if False else None
        U = np.eye(D, dtype=complex)
        for n in range(N):
            U = o.expm(-1j * dt * H[n]) @ U
        return np.real(np.vdot(Ubar, np.exp(1j * ph)[:, None] * U))

    for trial in range(4):
        dH = np.stack([rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)) for _ in range(N)])  # any direction, not only Hermitian
        eps = 1e-6
        fd = (loss(Hs + eps * dH) - loss(Hs - eps * dH)) / (2 * eps)
        assert abs(fd - np.real(np.sum(np.conj(Hb) * dH))) < 1e-8 * max(1.0, abs(fd))


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,B", [(3, 11, 2), (9, 8, 3), (24, 6, 2), (50, 4, 1)])
def test_per_slice_vjp_vs_oracle(prop, D, N, B):
    """c3p_pwc_unitary_vjp with C3P_PER_SLICE_H (branch B): cotangents of the per-slice Hamiltonians (on-chip general-generator
    sweeps up to D = 40, the tiled sweep above)."""
    rng = np.random.default_rng(D + N)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    Hs = np.stack([np.stack([herm(0.6 / np.sqrt(D)) for _ in range(N)]) for _ in range(B)])
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    ph = rng.uniform(0, 2 * np.pi, size=(B, D))
    dt = 1.3
    got = np.asarray(prop.propagate_per_slice_vjp(Hs, dt, Ubar, fr_phase=ph))
    for b in range(B):
        want = o.pwc_per_slice_hamiltonian_cotangents(Hs[b], dt, Ubar[b], ph[b])
        assert np.abs(got[b] - want).max() < 1e-10 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,B,hermitian", [(2, 90, 3, True), (5, 70, 2, False), (12, 40, 2, True), (13, 33, 2, False), (20, 26, 1, True),
                                             (24, 20, 2, True), (27, 64, 2, True), (32, 18, 1, False), (36, 17, 1, True), (40, 16, 1, True)])
def test_per_slice_vjp_on_chip_sweeps(prop, D, N, B, hermitian):
    """Branch B on the on-chip general-generator sweeps: every small-D dimension class and every mid-D geometry class, several
    time segments, Hermitian and NON-Hermitian slice Hamiltonians (nothing is assumed about them: the squaring plan of the
    backward pass uses the row-sum norm of X_n, i.e. the 1-norm of the X_n^H it exponentiates), frame phases -- against the
    FD-pinned oracle and against the tiled sweep on the same inputs."""
    from c3_amd import _lib

    rng = np.random.default_rng(7 * D + N)
    def mat(s):
        a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
        h = s * (a + a.conj().T) / 2
        if hermitian:
            return h
        # a lossy effective Hamiltonian H - i Gamma (Gamma >= 0: the chain decays instead of blowing up) plus a non-normal
        # part with very different row and column sums
        g = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
        up = np.triu(rng.normal(size=(D, D)), 1) * (1.0 + np.arange(D)[:, None]) / D
        return h - 1j * 0.3 * s * (g @ g.conj().T) / D + 0.5 * s * up
    Hs = np.stack([np.stack([mat(0.5 / np.sqrt(D)) for _ in range(N)]) for _ in range(B)])
    Ubar = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    ph = rng.uniform(0, 2 * np.pi, size=(B, D))
    dt = 1.1
    got = np.asarray(prop.propagate_per_slice_vjp(Hs, dt, Ubar, fr_phase=ph))
    assert _lib.last_kernel() == ("smalld" if D <= 12 else "mfma")
    _lib.set_option("tiled_grad", "1")
    try:
        tiled = np.asarray(prop.propagate_per_slice_vjp(Hs, dt, Ubar, fr_phase=ph))
    finally:
        _lib.set_option("tiled_grad", None)
    assert np.abs(got - tiled).max() < 1e-10 * np.abs(tiled).max()
    for b in range(B):
        want = o.pwc_per_slice_hamiltonian_cotangents(Hs[b], dt, Ubar[b], ph[b])
        assert np.abs(got[b] - want).max() < 1e-10 * np.abs(want).max()
