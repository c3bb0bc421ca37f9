"""Host-side construction of the inputs the propagator path consumes.

The reference obtains `h0`, `hks`, `col_ops` from `Model` (c3/model.py) and the
control signals from `Generator` (c3/generator/generator.py).  Those subsystems
are out of scope; what the hot path needs from them is
a handful of dense complex128 arrays.  This module builds those arrays with
numpy for (a) the synthetic bench/parity workloads specified in SURVEY.md 8d
and (b) lightweight `ChipModel` / `SignalSource` objects that expose exactly the
methods `pwc` / `ode_solver` call on a reference `Model` / `Generator`
(propagation.py:282-321, 691-704), so the drop-in callables can be exercised
without TensorFlow.

Everything here runs once per parameter set on the host (O(D^3)); it is plumbing,
not the measured path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

TWO_PI = 2.0 * np.pi

# --------------------------------------------------------------------------
# Operators (restating qt_utils.py:58-96, hamiltonians.py:17-54,80-99,126-140,
# chip.py:152-242, model.py:453-534)
# --------------------------------------------------------------------------


def annihilator(dim: int) -> np.ndarray:
    return np.diag(np.sqrt(np.arange(1, dim)), k=1).astype(np.complex128)


def embed(op: np.ndarray, index: int, dims: Sequence[int]) -> np.ndarray:
    """Operator on subsystem `index` extended to the product space (qt_utils.py:68-96)."""
    out = np.eye(1, dtype=np.complex128)
    for j, d in enumerate(dims):
        out = np.kron(out, op if j == index else np.eye(d, dtype=np.complex128))
    return out


def annihilators(dims: Sequence[int]) -> List[np.ndarray]:
    return [embed(annihilator(d), i, dims) for i, d in enumerate(dims)]


def state_labels(dims: Sequence[int]):
    return list(np.ndindex(*dims))


def bare_drift(dims, freqs_hz, anhars_hz, couplings_hz) -> np.ndarray:
    """sum_i 2pi f_i n_i + 2pi a_i n_i(n_i-1)/2 + sum_{i<j} 2pi g (a_i^+ + a_i)(a_j^+ + a_j).

    `couplings_hz` maps (i, j) -> g.  Duffing term only for dim > 2 (chip.py:165-168).
    """
    a = annihilators(dims)
    tot = int(np.prod(dims))
    H = np.zeros((tot, tot), dtype=np.complex128)
    for i, d in enumerate(dims):
        n = a[i].conj().T @ a[i]
        H += TWO_PI * freqs_hz[i] * n
        if d > 2:
            H += TWO_PI * anhars_hz[i] * 0.5 * ((n - np.eye(tot)) @ n)
    for (i, j), g in couplings_hz.items():
        H += TWO_PI * g * ((a[i].conj().T + a[i]) @ (a[j].conj().T + a[j]))
    return H


def dressing_transform(H_bare: np.ndarray) -> np.ndarray:
    """Eigenbasis re-ordered by overlap with the bare states and sign-fixed
    (model.py:453-502, the `max_probabilities > 0.5` branch)."""
    e, v = np.linalg.eigh(H_bare)
    v_sq = (v * v.conj()).real
    if v_sq.max(axis=0).min() <= 0.5:
        raise ValueError("states overly dressed; the greedy recovery branch is not restated")
    reorder = (v_sq > 0.5).astype(np.float64)
    signed = np.sign(v.real) * reorder
    return (v @ signed.T).astype(np.complex128)


def dress(op: np.ndarray, T: np.ndarray) -> np.ndarray:
    return T.conj().T @ op @ T


def qubit_collapse_op(a: np.ndarray, t1: Optional[float], t2star: Optional[float]) -> np.ndarray:
    """One summed collapse operator per subsystem, no temperature term (chip.py:205-242)."""
    L = np.zeros_like(a)
    if t1 is not None:
        L = L + (1.0 / t1) ** 0.5 * a
    if t2star is not None:
        L = L + (0.5 / t2star) ** 0.5 * (2.0 * (a.conj().T @ a))
    return L


def excitation_cutter(dims: Sequence[int], max_excitations: int) -> np.ndarray:
    """Selection matrix of product states with at most `max_excitations` quanta (model.py:198-216)."""
    labels = state_labels(dims)
    keep = [i for i, l in enumerate(labels) if sum(l) <= max_excitations]
    C = np.zeros((len(keep), len(labels)), dtype=np.complex128)
    for r, i in enumerate(keep):
        C[r, i] = 1.0
    return C


def centred_time_grid(t_start: float, t_end: float, resolution: float) -> np.ndarray:
    """ts = linspace(t_start + dt/2, t_end - dt/2, N), N = int(|t_end-t_start| res) (devices.py:72-122)."""
    dt = 1.0 / resolution
    n = int(np.abs(t_start - t_end) * resolution)
    return np.linspace(t_start + dt / 2, t_end - dt / 2, n)


# --------------------------------------------------------------------------
# Duck-typed Model / Generator / Instruction
# --------------------------------------------------------------------------


def transmon_factor(phi: float, phi_0: float, d: float = 0.0) -> float:
    """SQUID tuning factor (cos^2 + d^2 sin^2)^(1/4) of a flux-tunable transmon (chip.py:355-371)."""
    x = np.pi * phi / phi_0
    return float(np.sqrt(np.sqrt(np.cos(x) ** 2 + d**2 * np.sin(x) ** 2)))


def transmon_freq(freq_hz: float, anhar_hz: float, phi: float, phi_0: float, d: float = 0.0) -> float:
    """Biased frequency (freq - anhar) * factor + anhar (chip.py:377-382)."""
    return (freq_hz - anhar_hz) * transmon_factor(phi, phi_0, d) + anhar_hz


def tunable_coupler_problem():
    """Operators of the reference's tunable-coupler integration test (test/test_tunable_coupler.py:30-157).

    Subsystem order [TC, Q1, Q2] (3 levels each, D = 27), XX couplings Q1-TC and Q2-TC, dressed;
    the only driven line in the `crzp` gate is the flux line, whose signal is the TC frequency
    shift in rad/s (FluxTuning, devices.py:497-525) multiplying the dressed number operator
    (z_drive, hamiltonians.py:166-182).  Returns (h0 [27,27], hk_tc [27,27]).
    """
    phi_0 = 10.0
    f_tc = transmon_freq(8.1e9, -235e6, phi_0 * 0.23, phi_0, 0.36)
    dims = [3, 3, 3]
    H = bare_drift(dims, [f_tc, 6.189e9, 5.089e9], [-235e6, -286e6, -310e6], {(1, 0): 142e6, (2, 0): 116e6})
    T = dressing_transform(H)
    a = annihilators(dims)
    return dress(H, T), dress(a[0].conj().T @ a[0], T)


@dataclass
class Gate:
    """The attributes of `Instruction` the path reads (experiment.py:471; gates.py)."""

    name: str
    t_start: float
    t_end: float
    channels: List[str] = field(default_factory=list)
    carrier_freqs: Dict[str, float] = field(default_factory=dict)  # rad/s, per line
    framechanges: Dict[str, float] = field(default_factory=dict)

    def get_key(self) -> str:
        return self.name


class ChipModel:
    """Coupled-transmon model exposing what `pwc`/`ode_solver` call on `Model`.

    get_Hamiltonians (model.py:353-366), get_Hamiltonian (374-412),
    get_Lindbladians (417-421), cut/blowup_excitations (218-224),
    get_Frame_Rotation (536-578), Hs_of_t is done by the solver host code.
    """

    def __init__(self, dims, freqs_hz, anhars_hz, couplings_hz, drive_lines: Dict[str, int], *, t1=None, t2star=None, dressed=True):
        self.dims = list(dims)
        self.tot_dim = int(np.prod(dims))
        self.names = [f"Q{i+1}" for i in range(len(dims))]
        self.ann_opers = annihilators(dims)
        self.drive_lines = dict(drive_lines)
        self.dressed = dressed
        self.lindbladian = False
        self.controllability = True
        self.use_FR = False
        self.dephasing_strength = 0.0
        self.max_excitations = 0
        self.ex_cutter = None
        H = bare_drift(dims, freqs_hz, anhars_hz, couplings_hz)
        self.transform = dressing_transform(H) if dressed else np.eye(self.tot_dim, dtype=np.complex128)
        self.drift_ham = dress(H, self.transform)
        self.control_hams = {
            line: dress(self.ann_opers[q].conj().T + self.ann_opers[q], self.transform)
            for line, q in drive_lines.items()
        }
        self.col_ops = []
        if t1 is not None or t2star is not None:
            for q, a in enumerate(self.ann_opers):
                c = qubit_collapse_op(a, None if t1 is None else t1[q], None if t2star is None else t2star[q])
                self.col_ops.append(dress(c, self.transform))

    # -- option surface -----------------------------------------------------
    def set_lindbladian(self, flag: bool):
        self.lindbladian = bool(flag)

    def set_FR(self, flag: bool):
        self.use_FR = bool(flag)

    def set_max_excitations(self, max_excitations: int):
        if max_excitations:
            self.ex_cutter = excitation_cutter(self.dims, max_excitations)
        self.max_excitations = max_excitations

    # -- accessors ----------------------------------------------------------
    def cut_excitations(self, op):
        C = self.ex_cutter
        return C @ op @ C.T

    def blowup_excitations(self, op):
        C = self.ex_cutter
        return C.T @ op @ C

    def get_Hamiltonians(self):
        drift, controls = self.drift_ham, dict(self.control_hams)
        if self.max_excitations:
            drift = self.cut_excitations(drift)
            controls = {k: self.cut_excitations(v) for k, v in controls.items()}
        return drift, controls

    def get_Hamiltonian(self, signal=None):
        if signal is None:
            H = self.drift_ham
        else:
            H = self.drift_ham[None]
            for key, sig in signal.items():
                if key not in self.control_hams:
                    raise Exception(f"Signal channel {key} not in model systems")
                vals = np.asarray(sig["values"], dtype=np.float64)
                H = H + vals[:, None, None] * self.control_hams[key][None]
        if self.max_excitations:
            H = self.cut_excitations(H)
        return H

    def get_Lindbladians(self):
        return list(self.col_ops)

    def get_init_state(self):
        psi = np.zeros((self.tot_dim, 1), dtype=np.complex128)
        psi[0, 0] = 1.0
        return psi

    def number_operator(self, line: str) -> np.ndarray:
        a = self.ann_opers[self.drive_lines[line]]
        return a.conj().T @ a

    def get_Frame_Rotation(self, t_final, freqs: Dict[str, float], framechanges: Dict[str, float]) -> np.ndarray:
        """FR = expm(i sum_line n_q (w_line T + framechange)) (model.py:536-578); diagonal here."""
        if len(freqs) == 0:
            return np.eye(self.tot_dim, dtype=np.complex128)
        return np.diag(np.exp(1.0j * self.frame_rotation_phases(t_final, freqs, framechanges)))

    def get_dephasing_channel(self, t_final, amps: Dict[str, complex]) -> np.ndarray:
        """Element-wise product of per-line channels (model.py:597-639)."""
        Id = np.ones((self.tot_dim**2, self.tot_dim**2), dtype=np.complex128) * np.kron(np.eye(self.tot_dim), np.eye(self.tot_dim))
        ch = Id
        for line, amp in amps.items():
            z = np.exp(1.0j * np.pi * np.real(np.diag(self.number_operator(line))))
            Z = np.diag(np.kron(z, np.conj(z)))
            p = t_final * amp * self.dephasing_strength
            if np.real(p) > 1 or np.real(p) < 0:
                raise ValueError("Dephasing channel strength {strength} is outside [0,1] range".format(strength=p))
            ch = ch * ((1 - p) * Id + p * Z)
        return ch

    def frame_rotation_phases(self, t_final: float, freqs: Dict[str, float], framechanges: Dict[str, float]) -> np.ndarray:
        """Diagonal of i*exponent of model.py:536-578: FR = diag(exp(i*phase)); the number
        operators are diagonal in the product basis so FR is a row-phase."""
        phase = np.zeros(self.tot_dim)
        for line, f in freqs.items():
            phase = phase + np.real(np.diag(self.number_operator(line))) * (f * t_final + framechanges.get(line, 0.0))
        return phase


class SignalSource:
    """Stands in for `Generator.generate_signals(instr)` -> {chan: {"values","ts"}}
    (generator.py:172-229) with precomputed real f64 waveforms per gate."""

    def __init__(self, signals: Dict[str, Dict[str, Dict[str, np.ndarray]]], resolution: float = 100e9):
        self._signals = signals
        self.resolution = resolution

    def generate_signals(self, instr) -> Dict[str, Dict[str, np.ndarray]]:
        return self._signals[instr.get_key()]


# --------------------------------------------------------------------------
# Synthetic workloads
# --------------------------------------------------------------------------

_FREQS = (5.0e9, 5.6e9, 6.2e9)
_ANHARS = (-210e6, -240e6, -235e6)
_G = 20e6
_T1 = (27e-6, 23e-6, 25e-6)
_T2S = (39e-6, 31e-6, 35e-6)

CONFIGS = {
    # `gpus` = the number of GPUs BASELINE.json quotes the batch on (cfg3 / cfg5: B sharded over 8 GPUs)
    1: dict(name="cfg1 X90 D=3", dims=(3,), N=200, B=1, gpus=1, lindblad=False),
    2: dict(name="cfg2 CR D=9", dims=(3, 3), N=1000, B=256, gpus=1, lindblad=False),
    3: dict(name="cfg3 coupler D=27", dims=(3, 3, 3), N=2000, B=4096, gpus=8, lindblad=False),
    4: dict(name="cfg4 Lindblad D=9 (81x81)", dims=(3, 3), N=1000, B=512, gpus=1, lindblad=True),
    5: dict(name="cfg5 3 qubits D=36", dims=(3, 3, 4), N=5000, B=8192, gpus=8, lindblad=False),
}


@dataclass
class Workload:
    name: str
    dims: tuple
    D: int
    K: int
    N: int
    B: int
    dt: float
    h0: np.ndarray  # c128 [D,D]
    hks: np.ndarray  # c128 [K,D,D]
    signals: np.ndarray  # f64 [B,K,N]
    fr_phase: np.ndarray  # f64 [B,D]   (U <- diag(exp(i phase)) U)
    col_ops: Optional[np.ndarray]  # c128 [C,D,D] or None
    lindblad: bool
    ts: np.ndarray


def make_workload(cfg: int, B: Optional[int] = None, N: Optional[int] = None, seed: Optional[int] = None, b_offset: int = 0) -> Workload:
    """Deterministic synthetic inputs for config `cfg`.

    `B`/`N` override the configured sizes (parity tests run reduced sizes);
    `b_offset` selects a contiguous shard of the sample axis so that ranks of a
    multi-GPU run draw disjoint, reproducible samples.
    """
    c = CONFIGS[cfg]
    dims = c["dims"]
    nq = len(dims)
    B = c["B"] if B is None else B
    N = c["N"] if N is None else N
    seed = 20240 + cfg if seed is None else seed
    couplings = {(i, j): _G for i in range(nq) for j in range(i + 1, nq)}
    H = bare_drift(dims, _FREQS[:nq], _ANHARS[:nq], couplings)
    T = dressing_transform(H)
    a = annihilators(dims)
    h0 = dress(H, T)
    hks = np.stack([dress(a[k].conj().T + a[k], T) for k in range(nq)])
    dt = 1e-11
    ts = (np.arange(N) + 0.5) * dt
    Tg = N * dt
    env = np.exp(-((ts - Tg / 2) ** 2) / (2 * (Tg / 4) ** 2))
    signals = np.empty((B, nq, N), dtype=np.float64)
    # one generator per sample so shards are reproducible independently of B
    for b in range(B):
        rng = np.random.default_rng([seed, b + b_offset])
        A = rng.uniform(0.1, 0.6, size=nq)
        phi = rng.uniform(0.0, TWO_PI, size=nq)
        for k in range(nq):
            w = TWO_PI * (_FREQS[k] + 50e6)
            signals[b, k] = TWO_PI * 1e9 * A[k] * env * np.cos(w * ts + phi[k])
    nums = [np.real(np.diag(x.conj().T @ x)) for x in a]
    phase = np.zeros(h0.shape[0])
    for k in range(nq):
        phase = phase + nums[k] * (TWO_PI * (_FREQS[k] + 50e6) * Tg)
    fr_phase = np.broadcast_to(phase, (B, phase.shape[0])).copy()
    col = None
    if c["lindblad"]:
        col = np.stack([dress(qubit_collapse_op(a[q], _T1[q], _T2S[q]), T) for q in range(nq)])
    return Workload(
        name=f"{c['name']} N={N} B={B}",  # the sizes actually built (B = this rank's samples)
        dims=tuple(dims),
        D=h0.shape[0],
        K=nq,
        N=N,
        B=B,
        dt=dt,
        h0=h0,
        hks=hks,
        signals=signals,
        fr_phase=fr_phase,
        col_ops=col,
        lindblad=c["lindblad"],
        ts=ts,
    )
