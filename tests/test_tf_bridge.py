"""c3_amd/tf_bridge.py: the differentiable provider for the reference's @tf.function / GradientTape loop
(optimizers/optimalcontrol.py:200-228, optimizers/optimizer.py:210-215).

NOT RUN AGAINST TENSORFLOW (none in this image).  The TensorFlow calls of the bridge go through tests/tf_standin.py; what is
checked is everything on this side of that line: what the forward callback hands the library and returns, and that the gradient
function registered for the op returns the vector-Jacobian products of the oracle for a hand-fed cotangent.
  * CPU (`-m "not gpu"`): the library entry points are replaced by the oracle (injected compute, as tests/test_dist_gloo.py
    does) -- the wiring of the bridge alone;
  * GPU (`-m gpu`): nothing replaced -- the HIP path under the bridge against the oracle.
"""
import numpy as np
import pytest

from oracle import c3_oracle as o
from c3_amd import workloads
from tf_standin import StandIn


class Instr:
    def __init__(self, name, t_start=0.0, t_end=0.0):
        self.name, self.t_start, self.t_end = name, t_start, t_end

    def get_key(self):
        return self.name


def setup(N=24, lind=False, dims=(3, 3)):
    m = workloads.ChipModel(dims, (5e9, 5.6e9), (-210e6, -240e6), {(0, 1): 20e6}, {"d1": 0, "d2": 1}, t1=(27e-6, 23e-6), t2star=(39e-6, 31e-6))
    ts = (np.arange(N) + 0.5) * 1e-11
    T = N * 1e-11
    env = np.exp(-((ts - T / 2) ** 2) / (2 * (T / 4) ** 2))
    sig = {"g": {"d1": {"values": 2 * np.pi * 4e8 * env * np.cos(2 * np.pi * 5.05e9 * ts), "ts": ts},
                 "d2": {"values": 2 * np.pi * 3e8 * env * np.cos(2 * np.pi * 5.65e9 * ts + 1.0), "ts": ts}}}
    m.set_lindbladian(lind)
    return m, workloads.SignalSource(sig), Instr("g", 0.0, T)


def _oracle_backed(monkeypatch):
    """replace the four library entry points the bridge calls by the oracle (CPU wiring test only)"""
    from c3_amd import propagation as p

    def propagate_batch(h0, hks, signals, dt, *, col_ops=None, lindbladian=False, want_dUs=False, **kw):
        if hks is None:
            dUs = o.tf_batch_propagate(np.asarray(h0), None, None, dt, 1 << 30, col_ops=col_ops, lindbladian=lindbladian)
        else:
            dUs = o.tf_batch_propagate(np.asarray(h0), np.asarray(hks), np.asarray(signals[0]), dt, 1 << 30, col_ops=col_ops, lindbladian=lindbladian)
        return {"U": o.tf_matmul_left(dUs)[None], "dUs": dUs[None] if want_dUs else None}

    def propagate_batch_vjp(h0, hks, signals, dt, U_bar, *, want_model_grads=False, **kw):
        g = o.pwc_signal_gradient(h0, hks, signals[0], dt, U_bar[0])[None]
        if not want_model_grads:
            return g
        Hs = h0[None] + np.einsum("kn,kij->nij", signals[0], hks)
        Hb = o.pwc_per_slice_hamiltonian_cotangents(Hs, dt, U_bar[0])
        return g, Hb.sum(axis=0)[None], np.einsum("kn,nij->kij", signals[0], Hb)[None]

    monkeypatch.setattr(p, "propagate_batch", propagate_batch)
    monkeypatch.setattr(p, "propagate_batch_vjp", propagate_batch_vjp)
    monkeypatch.setattr(p, "propagate_batch_lindblad_vjp", lambda h0, hks, s, dt, col, Ub, **kw: o.pwc_lindblad_signal_gradient(h0, hks, col, s[0], dt, Ub[0])[None])
    monkeypatch.setattr(p, "propagate_per_slice_vjp", lambda hs, dt, Ub, **kw: o.pwc_per_slice_hamiltonian_cotangents(hs, dt, Ub[0])[None])


def _bridge(tf):
    from c3_amd import tf_bridge

    tf_bridge.use_tf_module(tf)
    tf_bridge.options.update(want_dUs=True, model_grads=True)
    return tf_bridge


def _check_closed(tf_bridge, tf):
    m, gen, instr = setup(N=24)
    got = tf_bridge.pwc_tf(m, gen, instr, [], None)
    ref = o.pwc(m, gen, instr, None, None)
    assert np.linalg.norm(np.asarray(got["U"]) - ref["U"]) < 1e-10
    assert np.abs(np.asarray(got["dUs"]) - ref["dUs"]).max() < 1e-12
    assert np.array_equal(np.asarray(got["ts"]), ref["ts"])
    # the gradient function the op registered, fed a cotangent by hand (what tape.gradient does with d goal / d U)
    assert len(tf.grad_fns) == 1
    rng = np.random.default_rng(1)
    Ub = rng.normal(size=(9, 9)) + 1j * rng.normal(size=(9, 9))
    g0, gk, gs = tf.grad_fns[0](Ub, None)
    h0, hc = m.get_Hamiltonians()
    hks = np.stack([hc["d1"], hc["d2"]])
    sig = np.stack([gen.generate_signals(instr)[k]["values"] for k in ("d1", "d2")])
    dt = 1e-11
    want = o.pwc_signal_gradient(h0, hks, sig, dt, Ub)
    assert np.asarray(gs).shape == (2, 24) and np.abs(np.asarray(gs) - want).max() < 1e-9 * np.abs(want).max()
    # operator cotangents against central finite differences of the oracle's propagator (loss = Re sum conj(Ub) U)
    loss = lambda a, b: float(np.real(np.vdot(Ub, o.propagate_batch(a, b, sig[None], dt)[0])))
    for (A, which, idx) in ((g0, "h0", (1, 2)), (g0, "h0", (4, 4)), (gk, "hk", (1, 3, 0)), (gk, "hk", (0, 2, 5))):
        for part in (1.0, 1j):
            eps = 1e3 if which == "h0" else 1e-6  # h0 is ~1e10 rad/s, the control operators are O(1)
            dh0, dhk = np.zeros_like(h0), np.zeros_like(hks)
            (dh0 if which == "h0" else dhk)[idx] = part * eps
            fd = (loss(h0 + dh0, hks + dhk) - loss(h0 - dh0, hks - dhk)) / (2 * eps)
            an = np.asarray(A)[idx]
            an = an.real if part == 1.0 else an.imag  # grad = dL/dRe + i dL/dIm
            assert abs(fd - an) < 1e-5 * max(abs(fd), np.abs(np.asarray(A)).max() * 1e-2), (which, idx, part, fd, an)


def _check_lindblad(tf_bridge, tf):
    m, gen, instr = setup(N=10, lind=True, dims=(2, 2))
    got = tf_bridge.pwc_tf(m, gen, instr, [], None)
    ref = o.pwc(m, gen, instr, None, None)
    assert np.linalg.norm(np.asarray(got["U"]) - ref["U"]) < 1e-10
    rng = np.random.default_rng(2)
    Ub = rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16))
    gs = tf.grad_fns[-1](Ub, None)
    h0, hc = m.get_Hamiltonians()
    hks = np.stack([hc["d1"], hc["d2"]])
    sig = np.stack([gen.generate_signals(instr)[k]["values"] for k in ("d1", "d2")])
    want = o.pwc_lindblad_signal_gradient(h0, hks, np.asarray(m.get_Lindbladians()), sig, 1e-11, Ub)
    assert np.abs(np.asarray(gs) - want).max() < 1e-8 * np.abs(want).max()


def _check_per_slice(tf_bridge, tf):
    m, gen, instr = setup(N=12)
    m.controllability = False
    m.set_max_excitations(2)
    got = tf_bridge.pwc_tf(m, gen, instr, [], 10)
    ref = o.pwc(m, gen, instr, None, 10)
    assert np.asarray(got["U"]).shape == (9, 9) and np.linalg.norm(np.asarray(got["U"]) - ref["U"]) < 1e-10
    assert np.abs(np.asarray(got["dUs"]) - ref["dUs"]).max() < 1e-12
    hs = np.asarray(m.get_Hamiltonian(gen.generate_signals(instr)))
    rng = np.random.default_rng(3)
    Ub = rng.normal(size=hs.shape[1:]) + 1j * rng.normal(size=hs.shape[1:])
    hb = np.asarray(tf.grad_fns[-1](Ub, None))
    want = o.pwc_per_slice_hamiltonian_cotangents(hs, 1e-11, Ub)
    assert hb.shape == hs.shape and np.abs(hb - want).max() < 1e-9 * np.abs(want).max()


def test_bridge_wiring_with_the_oracle_as_compute(monkeypatch):
    _oracle_backed(monkeypatch)
    tf = StandIn()
    b = _bridge(tf)
    _check_closed(b, tf)
    _check_lindblad(b, tf)
    _check_per_slice(b, tf)
    assert "pwc_tf" in __import__("c3_amd.propagation", fromlist=["x"]).unitary_provider
    b.use_tf_module(None)


def test_bridge_options_and_signals_only_gradient(monkeypatch):
    _oracle_backed(monkeypatch)
    tf = StandIn()
    b = _bridge(tf)
    b.options.update(want_dUs=False, model_grads=False)
    try:
        m, gen, instr = setup(N=8)
        got = b.pwc_tf(m, gen, instr, [], None)
        assert got["dUs"] is None
        Ub = np.eye(9, dtype=complex)
        gs = tf.grad_fns[0](Ub, None)  # ONE gradient: the control samples
        assert np.asarray(gs).shape == (2, 8)
    finally:
        b.options.update(want_dUs=True, model_grads=True)
        b.use_tf_module(None)


def test_bridge_keeps_the_reference_guards(monkeypatch):
    """what the reference's pwc refuses, pwc_tf refuses too: a non-uniform time grid on the per-slice branch
    (propagation.py:301-308, same message), control samples with an imaginary part (the reference would use them as complex
    coefficients, the library takes real samples), and an instruction without drive lines (IndexError in the reference)."""
    from c3_amd._lib import C3PropError

    _oracle_backed(monkeypatch)
    tf = StandIn()
    b = _bridge(tf)
    try:
        m, gen, instr = setup(N=12)
        m.controllability = False
        sig = gen.generate_signals(instr)
        bad = {k: dict(v) for k, v in sig.items()}
        first = next(iter(bad))
        bad[first]["ts"] = np.array(bad[first]["ts"], copy=True)
        bad[first]["ts"][5] += 1e-7  # (the reference compares a VARIANCE in s^2 with 1e-5 dt in s: only a gross error trips it)
        with pytest.raises(Exception, match="Something with the times happend"):
            b.pwc_tf(m, workloads.SignalSource({"g": bad}), instr, [], None)
        m.controllability = True
        cplx = {k: dict(v) for k, v in sig.items()}
        cplx[first]["values"] = np.asarray(cplx[first]["values"]) + 1e-3j
        with pytest.raises(Exception, match="real control samples"):
            b.pwc_tf(m, workloads.SignalSource({"g": cplx}), instr, [], None)
        real_as_complex = {k: dict(v, values=np.asarray(v["values"]).astype(complex)) for k, v in sig.items()}
        got = b.pwc_tf(m, workloads.SignalSource({"g": real_as_complex}), instr, [], None)  # zero imaginary part: accepted
        assert np.linalg.norm(np.asarray(got["U"]) - o.pwc(m, gen, instr, None, None)["U"]) < 1e-10
        with pytest.raises(C3PropError, match="drives no line"):
            b.pwc_tf(m, workloads.SignalSource({"g": {}}), instr, [], None)
    finally:
        b.use_tf_module(None)


def test_bridge_without_tensorflow_fails_loudly():
    from c3_amd import tf_bridge
    from c3_amd._lib import C3PropError

    tf_bridge.use_tf_module(None)
    try:
        import tensorflow  # noqa: F401
    except ImportError:
        with pytest.raises(C3PropError, match="needs TensorFlow"):
            tf_bridge.pwc_tf(*setup(N=4), [], None)


@pytest.mark.gpu
def test_bridge_on_the_device(lib):
    """the HIP path under the bridge (host-pointer mode of the C ABI), nothing replaced"""
    from c3_amd import _lib

    _lib.require_gpu()
    tf = StandIn()
    b = _bridge(tf)
    try:
        _check_closed(b, tf)
        _check_lindblad(b, tf)
        _check_per_slice(b, tf)
    finally:
        b.use_tf_module(None)
