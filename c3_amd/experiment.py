"""Adapter between a C3-style parameter map and the HIP propagators.

A reference `Experiment` does not need this module: it takes `c3_amd.propagation.pwc` through its own plugin slot
(`set_prop_method(callable)`).  This adapter exists so the contract of that slot can be driven
without TensorFlow, and to offer the batched entry the reference lacks.  It is written against the CONTRACT
(SURVEY.md 8b), not against the reference's method bodies:

  * provider selection: None -> "pwc"; a registry name (unitary first, then state providers); or any callable;
  * a unitary provider is called as `f(model, generator, instr, folding_stack[steps], propagate_batch_size)` and
    returns `{"U", "dUs", "ts"}`; a state provider as `f(model, generator, instr, init_state, solver=, step_function=)`
    and returns `{"states", "ts"}`;
  * per gate: frame rotation (a ROW PHASE here -- the number operators are diagonal in the product basis -- applied as
    `exp(i phi)[:, None] * U`, `phi_r - phi_c` on the rows of a superoperator), then the optional dephasing channel
    (Lindblad only, else `ValueError`), results kept in `propagators` / `partial_propagators`
    (replaced or merged according to `overwrite_propagators`);
  * unknown gates raise `Exception("C3:Error: Gate '...' is not defined. ...")`.

The parameter map is duck-typed: `pmap.model`, `pmap.generator`, `pmap.instructions` (name -> object with `t_start`,
`t_end` and either the reference's `comps` or the `carrier_freqs` / `framechanges` dicts of `c3_amd.workloads.Gate`).
"""
from __future__ import annotations

import itertools
import time
from typing import Dict, Iterator, Optional, Tuple

import numpy as np

from . import propagation
from .propagation import state_provider, unitary_provider


def _tf_matmul_n_even(odd, even):
    """One level of the pairwise product tree when the level has an even number of factors."""
    return np.matmul(odd, even)


def _tf_matmul_n_odd(odd, even):
    """... an odd number: the last factor is carried to the next level unchanged."""
    return np.concatenate([np.matmul(odd, even[:-1]), even[-1:]], 0)


def folding_levels(n_steps: int) -> list:
    """The level functions that reduce `n_steps` factors to one (what `pwc` receives as `folding_stack`)."""
    levels = []
    while n_steps > 1:
        levels.append(_tf_matmul_n_odd if n_steps % 2 else _tf_matmul_n_even)
        n_steps = -(-n_steps // 2)
    return levels


def _num(q) -> float:
    return float(np.real(q.get_value())) if hasattr(q, "get_value") else float(np.real(q))


def carrier_frame(instr) -> Tuple[Dict[str, float], Dict[str, float]]:
    """Per drive line: the frequency the frame rotates at (carrier + the offset of the active envelope) and the
    frame change.  Reads the reference's `instr.comps` when present, else the plain dicts of `workloads.Gate`."""
    comps = getattr(instr, "comps", None)
    if not comps:
        freqs = dict(getattr(instr, "carrier_freqs", {}))
        fcs = getattr(instr, "framechanges", {})
        return freqs, {line: fcs.get(line, 0.0) for line in freqs}
    freqs, fcs = {}, {}
    for line, parts in comps.items():
        shift = 0.0
        for part in parts.values():
            par = getattr(part, "params", {})
            if "freq_offset" in par and _num(par["amp"]) != 0.0:
                shift = _num(par["freq_offset"])
        car = parts["carrier"].params
        freqs[line] = _num(car["freq"]) + shift
        fcs[line] = _num(car["framechange"])
    return freqs, fcs


class Experiment:
    def __init__(self, pmap=None, prop_method=None, sim_res=100e9):
        self.pmap = pmap
        self.sim_res = sim_res
        self.opt_gates = None
        self.propagators: Dict[str, np.ndarray] = {}
        self.partial_propagators: Dict = {}
        self.propagate_batch_size = None
        self.use_control_fields = True
        self.overwrite_propagators = True
        self.compute_propagators_timestamp = 0
        self.folding_stack: Dict[int, list] = {}
        self.prop_method = prop_method
        self.set_prop_method(prop_method)

    # ---- provider slot ----------------------------------------------------------------------------------------
    def set_prop_method(self, prop_method=None) -> None:
        if callable(prop_method):
            self.propagation = prop_method
            return
        name = "pwc" if prop_method is None else prop_method
        self.propagation = unitary_provider[name] if name in unitary_provider else state_provider[name]
        if prop_method is None and self.pmap is not None:
            self._compute_folding_stack()

    def _steps(self, instr) -> int:
        return int((instr.t_end - instr.t_start) * self.sim_res)

    def _compute_folding_stack(self) -> None:
        self.folding_stack = {n: folding_levels(n) for n in {self._steps(i) for i in self.pmap.instructions.values()}}

    def set_opt_gates(self, gates) -> None:
        self.opt_gates = [gates] if isinstance(gates, str) else gates

    def set_opt_gates_seq(self, seqs) -> None:
        self.opt_gates = list(set(itertools.chain.from_iterable(seqs)))

    # ---- gates ------------------------------------------------------------------------------------------------
    def _gates(self) -> Iterator[Tuple[str, object]]:
        table = self.pmap.instructions
        for name in (table.keys() if self.opt_gates is None else self.opt_gates):
            if name not in table:
                raise Exception(f"C3:Error: Gate '{name}' is not defined. Available gates are:\n {list(table.keys())}.")
            yield name, table[name]

    def _frame_phases(self, model, instr) -> np.ndarray:
        """phi with FR = diag(exp(i phi)) for this gate, in the (possibly excitation-cut) space the provider used."""
        freqs, fcs = carrier_frame(instr)
        t_gate = instr.t_end - instr.t_start
        if hasattr(model, "frame_rotation_phases"):
            return np.asarray(model.frame_rotation_phases(t_gate, freqs, fcs), dtype=np.float64)
        return np.angle(np.diag(np.asarray(model.get_Frame_Rotation(t_gate, freqs, fcs))))

    def _finish_gate(self, model, generator, instr, U: np.ndarray) -> np.ndarray:
        if model.use_FR:
            row = np.exp(1.0j * self._frame_phases(model, instr))
            if model.lindbladian:
                row = np.kron(row, np.conj(row))  # the diagonal of FR (x) FR*
            self.FR = np.diag(row)
            U = row[:, None] * U
        if model.dephasing_strength != 0.0:
            if not model.lindbladian:
                raise ValueError("Dephasing can only be added when lindblad is on.")
            lines = getattr(instr, "comps", None) or getattr(instr, "carrier_freqs", {})
            amps = {line: complex(generator.devices["awg"].get_average_amp()[0]) for line in lines}
            U = np.asarray(model.get_dephasing_channel(instr.t_end - instr.t_start, amps)) @ U
        return U

    def compute_propagators(self) -> Dict[str, np.ndarray]:
        model, generator = self.pmap.model, self.pmap.generator
        self.set_prop_method(self.prop_method)
        done, partial = {}, {}
        for name, instr in self._gates():
            model.controllability = self.use_control_fields
            res = self.propagation(model, generator, instr, self.folding_stack.get(self._steps(instr), []), self.propagate_batch_size)
            self.ts = res["ts"]
            done[name] = self._finish_gate(model, generator, instr, np.asarray(res["U"]))
            partial[name] = res["dUs"]
        if self.overwrite_propagators:
            self.propagators, self.partial_propagators = done, partial
        else:
            self.propagators.update(done)
            self.partial_propagators.update(partial)
        self.compute_propagators_timestamp = time.time()
        return done

    def compute_propagators_batch(self, gate: str, signals_batch: np.ndarray, fr: Optional[bool] = None) -> np.ndarray:
        """U[b] of `gate` for B parameter samples, signals_batch [B,K,N] (channel order = the gate's signal order), in
        ONE library call -- what the serial sample loops of the optimizers (optimalcontrol_robust.py:54-63,
        modellearning.py:305-318) would call once instead of B times.  The frame rotation rides in the kernel as row
        phases.  With `model.max_excitations` the propagation runs in the cut space and the result is embedded back
        into the full space, as `pwc` does."""
        model = self.pmap.model
        instr = self.pmap.instructions[gate]
        model.controllability = True
        h0, hks, _sig, _ts, dt, col_ops = propagation.gather_pwc_inputs(model, self.pmap.generator, instr)
        cut = bool(getattr(model, "max_excitations", 0))
        if cut and model.lindbladian:
            raise Exception("C3:Error: excitation cut of a Lindblad superoperator is undefined in the reference")
        nb = int(signals_batch.shape[0])
        phase = None
        if model.use_FR if fr is None else fr:
            ph = self._frame_phases(model, instr)
            if cut:
                ph = np.real(np.asarray(model.ex_cutter) @ ph)
            if model.lindbladian:
                ph = (ph[:, None] - ph[None, :]).ravel()
            phase = np.tile(ph, (nb, 1))
        U = np.asarray(propagation.propagate_batch(h0, hks, signals_batch, dt, col_ops=col_ops, lindbladian=bool(model.lindbladian), fr_phase=phase)["U"])
        if cut:
            C = np.asarray(model.ex_cutter)
            U = np.einsum("ri,brs,sj->bij", C, U, C)  # C^T U C per sample (model.py:222-224)
        return U

    # ---- state solvers ----------------------------------------------------------------------------------------
    def _initial_state(self, step_function: str) -> np.ndarray:
        psi = np.asarray(self.pmap.model.get_init_state(), dtype=np.complex128)
        return psi @ psi.conj().T if step_function == "von_neumann" else psi

    def compute_states(self, solver="rk4", step_function="schrodinger"):
        """All intermediate states of the gate sequence `opt_gates` (each gate starts from the last state of the
        previous one; time stamps continue)."""
        state = self._initial_state(step_function)
        states, stamps, t_off = [state[None]], [np.zeros(1, dtype=np.complex128)], 0.0
        self.set_prop_method("ode_solver")
        for _name, instr in self._gates():
            res = self.propagation(self.pmap.model, self.pmap.generator, instr, state, solver=solver, step_function=step_function)
            states.append(res["states"])
            stamps.append(res["ts"] + t_off)
            state, t_off = res["states"][-1], res["ts"][-1]
        return {"states": np.concatenate(states, 0), "ts": np.concatenate(stamps)}

    def compute_final_state(self, solver="rk4", step_function="schrodinger"):
        state = self._initial_state(step_function)
        self.set_prop_method("ode_solver_final_state")
        res = None
        for _name, instr in self._gates():
            res = self.propagation(self.pmap.model, self.pmap.generator, instr, state, solver=solver, step_function=step_function)
            state = res["states"]
        return {"states": res["states"], "ts": res["ts"][-1]}
