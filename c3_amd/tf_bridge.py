"""Differentiable, graph-safe unitary provider for the reference's C1 loop (NOT RUN AGAINST TENSORFLOW HERE: the image has no
TensorFlow; the three TensorFlow calls this file makes are exercised through a stand-in in tests/test_tf_bridge.py).

Why `c3_amd.propagation.pwc` is not enough for every optimiser of the reference:

* `OptimalControl.goal_run` is a `@tf.function` (c3/optimizers/optimalcontrol.py:200-228): while it is traced,
  `gen.generate_signals(instr)` and `model.get_Hamiltonians()` return SYMBOLIC tensors, which `np.asarray` cannot convert;
* gradient-based algorithms (`lbfgs`, `tf_sgd`, ... -- c3/libraries/algorithms.py) take `tape.gradient(goal, params)` from a
  `tf.GradientTape` (c3/optimizers/optimizer.py:210-215): numpy results are constants to the tape and the gradient is None.

`pwc_tf` keeps everything the reference does in TensorFlow in TensorFlow (signal generation, Hamiltonian assembly from model
parameters, frame rotations, fidelity) and replaces exactly the numerical inner boundary of SURVEY 8b --
`tf_batch_propagate` + `tf_matmul_n` (propagation.py:323-336) -- by ONE `tf.py_function` whose gradient is registered with
`tf.custom_gradient`:

    forward   U, dUs = libc3prop (c3p_pwc_unitary / c3p_pwc_lindblad)                       (h0, hks, signals, dt) -> U
    backward  signals_bar, h0_bar, hks_bar = libc3prop (c3p_pwc_unitary_vjp / c3p_pwc_lindblad_vjp) from U_bar

TensorFlow's convention for a real loss L and a complex tensor z is grad = dL/dRe z + i dL/dIm z, which is the library's
(`d loss = Re sum conj(U_bar) dU`), so cotangents pass through unchanged.

Limits, stated plainly:
* `dUs` is returned for the contract (experiment.py:479-481 stores it) but carries NO gradient (`tf.stop_gradient`): a goal
  function that differentiates through the partial propagators needs the reference provider.
* open systems (`model.lindbladian`): the gradient reaches the control samples only; the operators enter through
  `tf.stop_gradient`, so a model-parameter gradient comes back None (loudly) instead of silently zero.
* `model.controllability == False` (one Hamiltonian per slice, propagation.py:295-308): closed systems only, gradient w.r.t.
  every slice Hamiltonian through `propagate_per_slice_vjp`.
* one gate = one library call with a batch of one, as the reference's loop over gates does (experiment.py:448-523).  Batches
  of parameter samples belong to `c3_amd.optimal_control.goal_run_with_grad`, which needs no TensorFlow at all.

Usage:
    import c3_amd.tf_bridge as hip_tf
    exp.set_prop_method(hip_tf.pwc_tf)          # works under @tf.function and tf.GradientTape
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import propagation
from ._lib import C3PropError
from .propagation import unitary_deco

_tf_module = None

# module-level switches (a provider's signature is fixed by experiment.py:472-478).  They are read when `pwc_tf` RUNS, i.e.
# at TRACE time under `@tf.function`: a function traced once keeps the values it was traced with (re-trace to change them).
options = {
    "want_dUs": True,  # False: skip the [N,Dm,Dm] partial propagators (result["dUs"] is None)
    "model_grads": True,  # closed systems: also return h0_bar / hks_bar (costs one [N,D,D] cotangent stack)
}


def use_tf_module(mod) -> None:
    """Inject the TensorFlow module (tests hand a stand-in; None = import tensorflow lazily)."""
    global _tf_module
    _tf_module = mod


def _tf():
    global _tf_module
    if _tf_module is None:
        try:
            import tensorflow as tf  # noqa: PLC0415
        except ImportError as e:  # pragma: no cover  (no TensorFlow in this image)
            raise C3PropError("C3:Error: c3_amd.tf_bridge needs TensorFlow (the reference's own dependency); "
                              "without it use c3_amd.propagation.pwc (eager, gradient-free)") from e
        _tf_module = tf
    return _tf_module


def _np(x):
    return x.numpy() if hasattr(x, "numpy") else np.asarray(x)


# ---- the two host callbacks (numpy in / numpy out through the C ABI's host-pointer mode) --------------------------------


def _forward(h0, hks, signals, dt, col_ops, lindbladian: bool, want_dUs: bool):
    h0, dt = np.asarray(_np(h0), dtype=np.complex128), float(np.real(_np(dt)))
    col = None if col_ops is None else np.asarray(_np(col_ops), dtype=np.complex128)
    if hks is None:  # one Hamiltonian per slice
        r = propagation.propagate_batch(h0, None, None, dt, col_ops=col, lindbladian=lindbladian, want_dUs=want_dUs)
    else:
        sig = np.ascontiguousarray(np.real(_np(signals)), dtype=np.float64)
        r = propagation.propagate_batch(h0, np.asarray(_np(hks), dtype=np.complex128), sig[None], dt, col_ops=col, lindbladian=lindbladian, want_dUs=want_dUs)
    U = np.asarray(r["U"][0])
    dUs = np.asarray(r["dUs"][0]) if want_dUs else np.zeros((0,) + U.shape, dtype=np.complex128)
    return U, dUs


def _backward(h0, hks, signals, dt, col_ops, U_bar, lindbladian: bool, model_grads: bool):
    """(signals_bar [K,N] f64, h0_bar, hks_bar) -- or (hs_bar [N,D,D],) for per-slice Hamiltonians."""
    h0, dt = np.asarray(_np(h0), dtype=np.complex128), float(np.real(_np(dt)))
    U_bar = np.asarray(_np(U_bar), dtype=np.complex128)
    if hks is None:
        return (np.asarray(propagation.propagate_per_slice_vjp(h0, dt, U_bar[None]))[0],)
    hks = np.asarray(_np(hks), dtype=np.complex128)
    sig = np.ascontiguousarray(np.real(_np(signals)), dtype=np.float64)[None]
    if lindbladian:
        g = propagation.propagate_batch_lindblad_vjp(h0, hks, sig, dt, np.asarray(_np(col_ops), dtype=np.complex128), U_bar[None])
        return (np.asarray(g)[0],)
    if model_grads:
        g, g0, gk = propagation.propagate_batch_vjp(h0, hks, sig, dt, U_bar[None], want_model_grads=True)
        return np.asarray(g)[0], np.asarray(g0)[0], np.asarray(gk)[0]
    return (np.asarray(propagation.propagate_batch_vjp(h0, hks, sig, dt, U_bar[None]))[0],)


# ---- the differentiable op -----------------------------------------------------------------------------------------


def hip_propagate(h0, hks, signals, dt, col_ops=None, lindbladian: bool = False, want_dUs: bool = True, model_grads: bool = True):
    """(U [Dm,Dm], dUs [N,Dm,Dm] or None) as TensorFlow tensors, differentiable w.r.t. `signals` [K,N] (and, for closed
    systems, `h0` [D,D] / `hks` [K,D,D]; or the per-slice `h0` [N,D,D] when `hks` is None).  Replaces
    `tf_matmul_n(tf_batch_propagate(h0, hks, signals, dt, ...))` (propagation.py:323-336)."""
    tf = _tf()
    c128, f64 = tf.complex128, tf.float64
    h0 = tf.cast(h0, c128)
    dt = tf.cast(dt, f64)
    D = int(h0.shape[-1])
    Dm = D * D if lindbladian else D
    per_slice = hks is None
    if per_slice and lindbladian:
        raise C3PropError("C3:Error: pwc_tf: per-slice Hamiltonians are served for closed systems only")
    col = None if col_ops is None else tf.stop_gradient(tf.cast(col_ops, c128))
    if lindbladian:
        if col is None:
            raise C3PropError("C3:Error: lindbladian propagation needs collapse operators")
        h0 = tf.stop_gradient(h0)
    extra = [] if col is None else [col]

    def finish(U, dUs):
        U = tf.ensure_shape(U, [Dm, Dm])
        return U, (tf.stop_gradient(tf.ensure_shape(dUs, [None, Dm, Dm])) if want_dUs else None)

    if per_slice:

        @tf.custom_gradient
        def op(hs):
            U, dUs = tf.py_function(lambda a, d: _forward(a, None, None, d, None, False, want_dUs), [hs, dt], [c128, c128])

            def grad(U_bar, *_unused):
                (hs_bar,) = tf.py_function(lambda a, d, ub: _backward(a, None, None, d, None, ub, False, False), [hs, dt, U_bar], [c128])
                return tf.ensure_shape(hs_bar, hs.shape)

            return (U, dUs), grad

        return finish(*op(h0))

    hks = tf.cast(hks, c128)
    if lindbladian:
        hks = tf.stop_gradient(hks)
    signals = tf.cast(tf.math.real(signals), f64)
    K = int(hks.shape[0])

    def fwd(a, b, s, d, *c):
        return _forward(a, b, s, d, c[0] if c else None, lindbladian, want_dUs)

    if lindbladian or not model_grads:

        @tf.custom_gradient
        def op(sig):
            U, dUs = tf.py_function(fwd, [h0, hks, sig, dt] + extra, [c128, c128])

            def grad(U_bar, *_unused):
                (g,) = tf.py_function(lambda a, b, s, d, ub, *c: _backward(a, b, s, d, c[0] if c else None, ub, lindbladian, False),
                                      [h0, hks, sig, dt, U_bar] + extra, [f64])
                return tf.ensure_shape(g, sig.shape)

            return (U, dUs), grad

        return finish(*op(signals))

    @tf.custom_gradient
    def op(h0_, hks_, sig):
        U, dUs = tf.py_function(fwd, [h0_, hks_, sig, dt], [c128, c128])

        def grad(U_bar, *_unused):
            g, g0, gk = tf.py_function(lambda a, b, s, d, ub: _backward(a, b, s, d, None, ub, False, True), [h0_, hks_, sig, dt, U_bar], [f64, c128, c128])
            return tf.ensure_shape(g0, [D, D]), tf.ensure_shape(gk, [K, D, D]), tf.ensure_shape(g, sig.shape)

        return (U, dUs), grad

    return finish(*op(h0, hks, signals))


def _guard(tf, cond_value, bound, message: str):
    """`cond_value < bound` everywhere, as the reference's `if not np.all(... < ...): raise` (propagation.py:301-308).
    Eager: evaluated now, raises the reference's Exception.  While a `@tf.function` is traced the operands are symbolic and
    `np.all` cannot run: a `tf.debugging.assert_less` op is returned instead (InvalidArgumentError at run time) and the
    caller puts the result under `tf.control_dependencies`.  Returns the list of assert ops (empty when eager)."""
    eager = getattr(tf, "executing_eagerly", lambda: True)()
    if eager:
        if not np.all(_np(cond_value) < _np(bound)):
            raise Exception(message)
        return []
    return [tf.debugging.assert_less(cond_value, bound, message=message)]


def _real_signal(tf, values, guards: list):
    """The reference casts the control samples to complex128 (propagation.py:293) and a complex sample would enter the
    Hamiltonian as a complex coefficient; the library takes REAL samples (devices.py:936: i1 i2 + q1 q2 is real).  A sample
    with an imaginary part is therefore refused instead of silently projected."""
    dt = getattr(values, "dtype", None)
    is_cplx = dt is not None and ("complex" in str(dt))
    if is_cplx:
        im = tf.math.reduce_max(tf.math.abs(tf.math.imag(values)))
        guards += _guard(tf, im, 1e-300, "C3:Error: pwc_tf takes real control samples (the signal has an imaginary part)")
    return tf.cast(tf.math.real(values), tf.float64)


class _deps:
    """`tf.control_dependencies(ops)` when there are ops to depend on, a no-op otherwise (eager / stand-in)."""

    def __init__(self, tf, ops):
        self._cm = tf.control_dependencies(ops) if ops else None

    def __enter__(self):
        return self._cm.__enter__() if self._cm is not None else None

    def __exit__(self, *a):
        return self._cm.__exit__(*a) if self._cm is not None else False


@unitary_deco
def pwc_tf(model, gen, instr, folding_stack: list, batch_size=None) -> Dict:
    """The reference's `pwc` (propagation.py:258-341) with the propagation on the MI355X and a registered gradient: usable as
    `Experiment.set_prop_method(pwc_tf)` under `@tf.function` and `tf.GradientTape`.  Returns {"U", "dUs", "ts"} as
    TensorFlow tensors.  `folding_stack` / `batch_size` are accepted for call compatibility."""
    del folding_stack, batch_size
    tf = _tf()
    signal = gen.generate_signals(instr)
    lind = bool(model.lindbladian)
    col_ops = None
    guards: list = []  # graph-mode assert ops (empty when eager: the checks have raised already)
    if lind:
        col_ops = list(model.get_Lindbladians())
        if model.max_excitations:
            cutter = model.ex_cutter
            col_ops = [cutter @ c @ tf.transpose(cutter) for c in col_ops]
        col_ops = tf.stack([tf.cast(c, tf.complex128) for c in col_ops])
    if model.controllability:
        h0, hctrls = model.get_Hamiltonians()
        sigs, hks, ts = [], [], None
        for key in signal:
            sigs.append(_real_signal(tf, signal[key]["values"], guards))
            ts = signal[key]["ts"]
            hks.append(tf.cast(hctrls[key], tf.complex128))
        if not sigs:
            # (the reference fails here too: `ts` stays [] and `ts[1] - ts[0]` raises IndexError, propagation.py:284-310)
            raise C3PropError("C3:Error: pwc_tf: the instruction drives no line, so there is no time grid to propagate on")
        with _deps(tf, guards):
            U, dUs = hip_propagate(h0, tf.stack(hks), tf.stack(sigs), ts[1] - ts[0], col_ops=col_ops, lindbladian=lind,
                                   want_dUs=options["want_dUs"], model_grads=options["model_grads"])
    else:
        hs = model.get_Hamiltonian(signal)
        ts_list = tf.stack([sig["ts"][1:] for sig in signal.values()])
        ts = tf.math.reduce_mean(ts_list, axis=0)
        # the reference's two sanity checks of the time grid (propagation.py:301-308), same message
        tol = 1e-5 * (ts[1] - ts[0])
        guards += _guard(tf, tf.math.reduce_variance(ts_list, axis=0), tol, "C3Error:Something with the times happend.")
        guards += _guard(tf, tf.math.reduce_variance(ts[1:] - ts[:-1]), tol, "C3Error:Something with the times happend.")
        with _deps(tf, guards):
            U, dUs = hip_propagate(hs, None, None, ts[1] - ts[0], col_ops=col_ops, lindbladian=lind, want_dUs=options["want_dUs"])
    if model.max_excitations:
        if lind:
            raise C3PropError("C3:Error: blow-up of a cut Lindblad superoperator is undefined in the reference")
        U = model.blowup_excitations(U)  # TensorFlow ops of the model: differentiable as they are
        if dUs is not None:
            dUs = tf.vectorized_map(model.blowup_excitations, dUs)
    return {"U": U, "dUs": dUs, "ts": ts}
