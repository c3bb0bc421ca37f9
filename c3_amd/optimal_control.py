"""Batched goal-with-gradient for pulse parameters -- the loop body of the reference's optimizers.

Reference: `Optimizer.goal_run_with_grad` tapes the goal (optimizers/optimizer.py:206-216),
`OptimalControl.goal_run` is `fid_func(compute_propagators())` (optimalcontrol.py:200-228) and
`OptimalControlRobust.goal_run_with_grad` averages goal and gradient over noise instances in a
serial Python loop (optimalcontrol_robust.py:49-70).  Here the B instances are one batch:

  envelope rows --synthesize--> signals --propagate--> U --fidelity--> goal[b]
  d goal/d rows <--synth vjp-- d/d signals <--adjoint sweep-- U_bar <--cotangent--

everything resident in HBM.  The mapping between optimizer coordinates and physical parameter values
(`Quantity` scaling, c3objs.py) stays with the caller: it is element-wise host arithmetic.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import fidelities, propagation, signals
from ._lib import C3PropError

_COTANGENTS = {"unitary_infid": fidelities.unitary_infid_cotangent, "average_infid": fidelities.average_infid_cotangent,
               "lindbladian_unitary_infid": fidelities.lindbladian_unitary_infid_cotangent}


def goal_run_with_grad(
    h0,
    hks,
    env_params,
    env_shapes,
    carrier,
    t_start: float,
    t_end: float,
    awg_res: float,
    sim_res: float,
    ideal,
    index,
    dims,
    *,
    fr_phase=None,
    fid_func: str = "unitary_infid",
    col_ops=None,
    device="cuda:0",
    fused: Optional[bool] = None,
) -> Dict:
    """goals [B] and their gradients w.r.t. every envelope row, carrier pair and frame-rotation phase.

    Returns {"goal": [B], "grad_env": [B,K,E,NPAR], "grad_carrier": [B,K,2], "grad_fr_phase": [B,D] or None,
    "U": [B,D,D]} as torch CUDA tensors.

    `col_ops` [C,D,D] switches to the open-system path (model.lindbladian, propagation.py:551-585): U are the D^2 x D^2
    superoperators, `fid_func` an open-system goal ("lindbladian_unitary_infid", fidelities.py:221-249), `fr_phase` [B,D^2]
    the row phases of the superoperator, and the control gradient comes from `propagate_batch_lindblad_vjp`.

    `fused` (closed systems; default: wherever the library serves the shape) takes goal and gradient from ONE pass over the
    chains (`propagation.propagate_batch_goal_vjp`); False runs forward pass, cotangent and vector-Jacobian product as three
    calls -- same numbers, the forward segment products computed twice.
    """
    import torch

    if fid_func not in _COTANGENTS:
        raise C3PropError(f"C3:Error: no cotangent for fidelity '{fid_func}' (have {sorted(_COTANGENTS)})")
    ts = signals.create_ts(t_start, t_end, sim_res)
    if ts.shape[0] < 2:
        raise C3PropError("C3:Error: need at least two time slices")
    dt = float(ts[1] - ts[0])  # propagation.py:310
    as_dev = lambda x, dt_: x.to(device) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=dt_), device=device)
    env = as_dev(env_params, np.float64)
    car = as_dev(carrier, np.float64)
    h0d, hkd = as_dev(h0, np.complex128), as_dev(hks, np.complex128)
    ph = None if fr_phase is None else as_dev(fr_phase, np.float64)
    sig = signals.synthesize_signals(env, env_shapes, car, t_start, t_end, awg_res, sim_res)
    if col_ops is not None:
        if not fid_func.startswith("lindbladian"):
            raise C3PropError(f"C3:Error: '{fid_func}' is a closed-system goal; the Lindblad path needs a lindbladian_* one")
        cold = as_dev(col_ops, np.complex128)
        B, K, N, D = int(sig.shape[0]), int(sig.shape[1]), int(sig.shape[2]), int(h0d.shape[-1])
        tape = None
        if fused is not False and propagation.lindblad_tape_supported(B, K, N, D):
            # ONE forward pass: the superoperators and, on the tape, what their vector-Jacobian product reads (D = 2, 3, 7, 8, 9)
            try:
                r = propagation.propagate_batch_lindblad_taped(h0d, hkd, sig, dt, cold, fr_phase=ph)
                U, tape = r["U"], r["tape"]
            except C3PropError as e:  # a non-Hermitian Hamiltonian: the untaped pair serves it
                if "Hermitian" not in str(e):
                    raise
        if tape is None:
            U = propagation.propagate_batch(h0d, hkd, sig, dt, col_ops=cold, lindbladian=True, fr_phase=ph)["U"]
        U_bar, goal = _COTANGENTS[fid_func](ideal, U, index, dims)
        if tape is not None:
            g_sig = tape.vjp(U_bar)
        else:
            g_sig = propagation.propagate_batch_lindblad_vjp(h0d, hkd, sig, dt, cold, U_bar, fr_phase=ph)
    else:
        if fid_func.startswith("lindbladian"):
            raise C3PropError(f"C3:Error: '{fid_func}' needs collapse operators (col_ops)")
        B, D = int(sig.shape[0]), int(h0d.shape[-1])
        if fused is None:
            fused = propagation.goal_vjp_is_fused(B, D)
        if fused:
            # ONE pass over the chains: the backward pass's scan of the segment products evaluates the goal and starts the
            # adjoint sweep from its cotangent (c3p_pwc_unitary_goal_vjp); no second forward pass, no host-framework ops
            r = propagation.propagate_batch_goal_vjp(h0d, hkd, sig, dt, ideal, index, dims, kind="unitary" if fid_func == "unitary_infid" else "average", fr_phase=ph)
            g_env, g_car = signals.synthesize_signals_vjp(env, env_shapes, car, t_start, t_end, awg_res, sim_res, r["grad_signals"])
            return {"goal": r["goal"], "grad_env": g_env, "grad_carrier": g_car, "grad_fr_phase": r["grad_fr_phase"], "U": r["U"]}
        U = propagation.propagate_batch(h0d, hkd, sig, dt, fr_phase=ph)["U"]
        U_bar, goal = _COTANGENTS[fid_func](ideal, U, index, dims)
        g_sig = propagation.propagate_batch_vjp(h0d, hkd, sig, dt, U_bar, fr_phase=ph)
    g_env, g_car = signals.synthesize_signals_vjp(env, env_shapes, car, t_start, t_end, awg_res, sim_res, g_sig)
    g_ph = None
    if ph is not None:
        # U = diag(e^{i phi}) P  =>  d loss/d phi_i = Re sum_j conj(U_bar_ij) (i U_ij) = -Im sum_j conj(U_bar_ij) U_ij
        g_ph = -(torch.conj(U_bar) * U).sum(dim=-1).imag
    return {"goal": goal, "grad_env": g_env, "grad_carrier": g_car, "grad_fr_phase": g_ph, "U": U}


def robust_goal_run_with_grad(*args, **kwargs) -> Dict:
    """Mean goal and mean gradient over the batch of noise instances (optimalcontrol_robust.py:49-70),
    plus the per-instance values and their standard deviation the reference logs (:64-69)."""
    r = goal_run_with_grad(*args, **kwargs)
    out = {"goal": r["goal"].mean(), "goals_individual": r["goal"], "goal_std": r["goal"].std(unbiased=False),
           "grad_env": r["grad_env"].mean(dim=0), "grad_carrier": r["grad_carrier"].mean(dim=0),
           "gradient_std": r["grad_env"].std(dim=0, unbiased=False)}
    if r["grad_fr_phase"] is not None:
        out["grad_fr_phase"] = r["grad_fr_phase"].mean(dim=0)
    return out
