"""Batched callers of the propagator path for model learning and sensitivity sweeps.

The reference evaluates one control-parameter set at a time: `ModelLearning.goal_run` loops over `ipar`
(optimizers/modellearning.py:300-341), each pass setting the gate-set parameters, calling
`exp.compute_propagators()`, `exp.evaluate(sequences)` and `exp.process(...)` (:218-242), then scoring the
simulated populations against the measured ones with `g_LL_prime` (libraries/estimators.py:155-170);
`Sensitivity` drives the same function along a one-dimensional sweep of a model parameter
(optimizers/sensitivity.py:100-124).  Here all P parameter sets (or sweep points) are ONE batch per gate:

  signals[gate] [P,K,N] (+ per-set model operators)  --propagate_batch-->  U[gate] [P,D,D]
  sequences  --c3p_matmul_chain over P x sequences-->  U_seq [P,S,D,D]  -->  |U_seq psi0|^2  -->  sim_vals [P,S]

Everything up to the populations stays in HBM.  The likelihood is a few flops per value and stays on the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import propagation
from ._lib import C3PropError


def g_LL_prime(exp_values, sim_values, exp_stds, shots):
    """estimators.py:155-157: mean of ((m - s)^2 / (s (1 - s) / shots) - 1) / 2."""
    m = np.asarray(exp_values, dtype=np.float64)
    s = np.asarray(sim_values, dtype=np.float64)
    var = s * (1.0 - s) / np.asarray(shots, dtype=np.float64)
    return np.mean(((m - s) ** 2 / var - 1.0) / 2.0)


def g_LL_prime_combined(gs, weights):
    """estimators.py:168-170."""
    K = np.sum(weights)
    return np.sum(np.array(weights) * np.asarray(gs)) / K


def propagate_parameter_sets(h0, hks, gate_signals: Dict[str, np.ndarray], dt: float, *, fr_phase: Optional[Dict] = None, device=None) -> Dict:
    """U[gate] [P,D,D] for P parameter sets: one `propagate_batch` call per gate (the reference recomputes every
    gate inside the `ipar` loop, modellearning.py:236-237).  `h0` / `hks` may carry a leading P axis (a model
    parameter differs between the sets, as in a sensitivity sweep); `gate_signals[gate]` is [P,K,N]."""
    out = {}
    P = None
    for gate, sig in gate_signals.items():
        if P is None:
            P = int(sig.shape[0])
        if int(sig.shape[0]) != P:
            raise C3PropError(f"C3:Error: gate {gate!r} has {int(sig.shape[0])} parameter sets, expected {P}")
        ph = None if fr_phase is None else fr_phase.get(gate)
        if device is not None:
            import torch

            to = lambda x, dt_: x.to(device) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=dt_), device=device)
            out[gate] = propagation.propagate_batch(to(h0, np.complex128), to(hks, np.complex128), to(sig, np.float64), dt, fr_phase=None if ph is None else to(ph, np.float64))["U"]
        else:
            out[gate] = propagation.propagate_batch(h0, hks, sig, dt, fr_phase=ph)["U"]
    return out


def evaluate_sequences_batch(gate_Us: Dict, sequences: Sequence[Sequence[str]]):
    """U_seq [P,S,D,D]: total propagator of every sequence for every parameter set, `... U2 U1 U0`
    (propagation.py:588-627).  Sequences of equal length are multiplied in one `c3p_matmul_chain` launch over
    P x (sequences of that length) chains; an empty sequence is the identity."""
    first = next(iter(gate_Us.values()))
    is_t = propagation._is_torch(first)
    P, D = int(first.shape[0]), int(first.shape[-1])
    S = len(sequences)
    if is_t:
        import torch

        out = torch.empty((P, S, D, D), dtype=first.dtype, device=first.device)
        eye = torch.eye(D, dtype=first.dtype, device=first.device)
        stack = torch.stack
    else:
        out = np.empty((P, S, D, D), dtype=np.complex128)
        eye = np.eye(D, dtype=np.complex128)
        stack = np.stack
    by_len: Dict[int, List[int]] = {}
    for si, seq in enumerate(sequences):
        for g in seq:
            if g not in gate_Us:
                raise C3PropError(f"C3:Error: sequence uses gate {g!r} without a propagator")
        by_len.setdefault(len(seq), []).append(si)
    for L, idx in by_len.items():
        if L == 0:
            for si in idx:
                out[:, si] = eye
            continue
        if L == 1:
            for si in idx:
                out[:, si] = gate_Us[sequences[si][0]]
            continue
        # [P, n, L, D, D] -> P n chains of L factors, first gate applied first
        M = stack([stack([gate_Us[g] for g in sequences[si]], 1) for si in idx], 1)
        prod = propagation.tf_matmul_left(M.reshape((P * len(idx), L, D, D)))
        prod = prod.reshape((P, len(idx), D, D))
        for j, si in enumerate(idx):
            out[:, si] = prod[:, j]
    return out


def populations_batch(U_seq, psi_init):
    """|U_seq psi0|^2 [P,S,D]  (experiment.py:291-301,603-624, unitary case)."""
    if propagation._is_torch(U_seq):
        import torch

        psi = torch.as_tensor(np.asarray(psi_init).reshape(-1), dtype=U_seq.dtype, device=U_seq.device)
        amp = U_seq @ psi
        return amp.real**2 + amp.imag**2
    amp = U_seq @ np.asarray(psi_init, dtype=np.complex128).reshape(-1)
    return np.abs(amp) ** 2


def process_batch(pops, label_indices: Optional[Sequence[int]] = None):
    """`Experiment.process` without a confusion matrix or rescaling (experiment.py:355-400): the summed
    population of the selected state labels, [P,S]; all populations [P,S,D] when no labels are given."""
    if label_indices is None:
        return pops
    idx = list(label_indices)
    return pops[..., idx].sum(-1)


def goal_run_batched(h0, hks, gate_signals: Dict, dt: float, data_sets: Sequence[Dict], psi_init, label_indices, *, fr_phase: Optional[Dict] = None, device=None) -> Dict:
    """`ModelLearning.goal_run` (modellearning.py:285-360) with the `ipar` loop as one batch.

    `data_sets[p]` = {"seqs": [...], "results": [...], "results_std": [...], "shots": [...]} for parameter set
    p, whose pulses are row p of every `gate_signals[gate]`.  All sets must use the same sequence list (the
    reference's `seqs_per_point`).  Returns {"goal", "goals" [P], "sim_vals" [P,S]}; the per-set goal is
    `g_LL_prime`, combined with the sequence counts as weights.
    """
    P = len(data_sets)
    seqs = data_sets[0]["seqs"]
    for d in data_sets:
        if d["seqs"] != seqs:
            raise C3PropError("C3:Error: batched model learning needs the same sequences for every parameter set")
    Us = propagate_parameter_sets(h0, hks, gate_signals, dt, fr_phase=fr_phase, device=device)
    if int(next(iter(Us.values())).shape[0]) != P:
        raise C3PropError("C3:Error: number of data sets and parameter sets differ")
    sim = process_batch(populations_batch(evaluate_sequences_batch(Us, seqs), psi_init), label_indices)
    sim = sim.cpu().numpy() if propagation._is_torch(sim) else np.asarray(sim)
    goals = np.array([g_LL_prime(d["results"], sim[p], d["results_std"], d["shots"]) for p, d in enumerate(data_sets)])
    weights = [len(seqs)] * P
    return {"goal": g_LL_prime_combined(goals, weights), "goals": goals, "sim_vals": sim}


def sensitivity_sweep(h0_of, hks_of, sweep_values: Sequence[float], gate_signals_one: Dict, dt: float, data_set: Dict, psi_init, label_indices, *, device=None) -> Dict:
    """`Sensitivity.sensitivity` for one swept model parameter (sensitivity.py:100-124): the goal at every sweep
    point, all points in one batch.  `h0_of(v)` / `hks_of(v)` build the (dressed) operators at value v -- the
    model update the reference performs per point (modellearning.py:227-232); the pulses are shared."""
    vals = list(sweep_values)
    P = len(vals)
    h0 = np.stack([np.asarray(h0_of(v), dtype=np.complex128) for v in vals])
    hks = np.stack([np.asarray(hks_of(v), dtype=np.complex128) for v in vals])
    sig = {g: np.broadcast_to(np.asarray(s, dtype=np.float64)[None], (P,) + tuple(np.shape(s))).copy() for g, s in gate_signals_one.items()}
    r = goal_run_batched(h0, hks, sig, dt, [data_set] * P, psi_init, label_indices, device=device)
    return {"values": np.asarray(vals), "goals": r["goals"], "sim_vals": r["sim_vals"]}
