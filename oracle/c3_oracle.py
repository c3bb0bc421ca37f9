"""CPU oracle for the C3 propagator hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the arithmetic the reference performs on
the path `Experiment.compute_propagators()` -> `c3/libraries/propagation.py`
(+ the helpers in `c3/utils/tf_utils.py`).  It exists so that the HIP kernels
in `c3_amd/csrc/` can be checked against an independent CPU result.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py`.  Nothing under `c3_amd/` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to this file.

Parity pinning: the oracle is pinned (tests/test_oracle_golden.py) against the
golden arrays the reference's own tests hold for this path
(`test/two_qubit_data.pickle`, `test/transmon_expanded.pickle`,
`test/test_tf_utils.pickle`, `test/tunable_coupler_data.pickle`; extracted to
`tests/golden/*.npz` by `tests/golden/make_golden.py`) and against closed
forms (`test/test_exp.py`).  ODE-solver values are not pinned by any reference
golden array ("ODE parity unpinned by golden values"); they are
pinned here by convergence to the PWC propagator.

Third-party arithmetic: the reference delegates the matrix exponential to
`tf.linalg.expm` (tensorflow>=2.15.0, requirements.txt:17, source not under
/root/reference) and signal interpolation to
`tfp.math.interp_regular_1d_grid` (tensorflow-probability>=0.12.1,
requirements.txt:19).  `expm` below restates the published algorithm TF
implements (Higham 2005, "The scaling and squaring method for the matrix
exponential revisited": Pade orders 3/5/7/9/13 selected per matrix from its
1-norm, then repeated squaring), anchored on the reference's call sites
(propagation.py:378,422,440,456,584) and golden vectors.

All citations are file:line relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------
# Matrix exponential (third-party tf.linalg.expm restated; Higham 2005)
# --------------------------------------------------------------------------

_PADE_B = {
    3: (120.0, 60.0, 12.0, 1.0),
    5: (30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0),
    7: (17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0),
    9: (
        17643225600.0,
        8821612800.0,
        2075673600.0,
        302702400.0,
        30270240.0,
        2162160.0,
        110880.0,
        3960.0,
        90.0,
        1.0,
    ),
    13: (
        64764752532480000.0,
        32382376266240000.0,
        7771770303897600.0,
        1187353796428800.0,
        129060195264000.0,
        10559470521600.0,
        670442572800.0,
        33522128640.0,
        1323241920.0,
        40840800.0,
        960960.0,
        16380.0,
        182.0,
        1.0,
    ),
}
# 1-norm thresholds theta_3, theta_5, theta_7, theta_9, theta_13 (Higham 2005, table 2.3)
PADE_THETA = (
    1.495585217958292e-2,
    2.539398330063230e-1,
    9.504178996162932e-1,
    2.097847961257068e0,
    5.371920351148152e0,
)
PADE_ORDERS = (3, 5, 7, 9, 13)
# number of matrix products each Pade order costs (used for the algorithmic
# flop count)
PADE_PRODUCTS = {3: 2, 5: 3, 7: 4, 9: 5, 13: 6}


def _pade_uv(A: np.ndarray, order: int):
    """U (odd part) and V (even part) of the [m/m] Pade numerator for a batch A[...,D,D]."""
    b = _PADE_B[order]
    eye = np.broadcast_to(np.eye(A.shape[-1], dtype=A.dtype), A.shape)
    A2 = A @ A
    if order == 3:
        U = A @ (b[3] * A2 + b[1] * eye)
        V = b[2] * A2 + b[0] * eye
        return U, V
    A4 = A2 @ A2
    if order == 5:
        U = A @ (b[5] * A4 + b[3] * A2 + b[1] * eye)
        V = b[4] * A4 + b[2] * A2 + b[0] * eye
        return U, V
    A6 = A4 @ A2
    if order == 7:
        U = A @ (b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * eye)
        V = b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * eye
        return U, V
    if order == 9:
        A8 = A6 @ A2
        U = A @ (b[9] * A8 + b[7] * A6 + b[5] * A4 + b[3] * A2 + b[1] * eye)
        V = b[8] * A8 + b[6] * A6 + b[4] * A4 + b[2] * A2 + b[0] * eye
        return U, V
    # order 13
    U = A @ (
        A6 @ (b[13] * A6 + b[11] * A4 + b[9] * A2)
        + b[7] * A6
        + b[5] * A4
        + b[3] * A2
        + b[1] * eye
    )
    V = (
        A6 @ (b[12] * A6 + b[10] * A4 + b[8] * A2)
        + b[6] * A6
        + b[4] * A4
        + b[2] * A2
        + b[0] * eye
    )
    return U, V


def expm_plan(A: np.ndarray):
    """Per-matrix (order, squarings) the reference's expm would pick.

    1-norm = max column sum of |a_ij|; order = first theta_m exceeding it;
    above theta_13 the matrix is scaled by 2^-s, s = floor(log2(norm/theta_13))
    clipped at 0 (TF's formula; scipy uses ceil -- the two agree to rounding
    error in the result, see tests/test_oracle_expm.py).
    """
    A = np.asarray(A)
    norm = np.abs(A).sum(axis=-2).max(axis=-1)
    order = np.full(norm.shape, 13, dtype=np.int64)
    for th, m in zip(PADE_THETA[:4][::-1], PADE_ORDERS[:4][::-1]):
        order = np.where(norm < th, m, order)
    with np.errstate(divide="ignore"):
        s = np.floor(np.log(norm / PADE_THETA[4]) / math.log(2.0))
    s = np.where(np.isfinite(s), np.maximum(s, 0.0), 0.0).astype(np.int64)
    s = np.where(order == 13, s, 0)
    return norm, order, s


def expm(A: np.ndarray) -> np.ndarray:
    """Batched matrix exponential, restating tf.linalg.expm for complex128/float64.

    Call sites in the reference: propagation.py:378,422,440,456,584; model.py:577,626.
    """
    A = np.asarray(A)
    if A.dtype not in (np.float64, np.complex128):
        A = A.astype(np.complex128 if np.iscomplexobj(A) else np.float64)
    batch_shape = A.shape[:-2]
    D = A.shape[-1]
    Af = A.reshape((-1, D, D))
    out = np.empty_like(Af)
    _, order, s = expm_plan(Af)
    for m in PADE_ORDERS:
        idx = np.nonzero(order == m)[0]
        if idx.size == 0:
            continue
        Am = Af[idx]
        sm = s[idx]
        if m == 13:
            Am = Am / (2.0 ** sm)[:, None, None]
        U, V = _pade_uv(Am, m)
        R = np.linalg.solve(V - U, V + U)
        if m == 13:
            for level in range(int(sm.max()) if sm.size else 0):
                sel = sm > level
                R[sel] = R[sel] @ R[sel]
        out[idx] = R
    return out.reshape(batch_shape + (D, D))


def tf_expm(A: np.ndarray, terms: int) -> np.ndarray:
    """Fixed-length Taylor series (propagation.py:630-655): I + A + sum_{k=2}^{terms-1} A^k/k!."""
    A = np.asarray(A, dtype=np.complex128)
    r = np.broadcast_to(np.eye(A.shape[-1], dtype=A.dtype), A.shape).copy()
    P = A.copy()
    r = r + A
    for k in range(2, terms):
        P = (P @ A) / complex(k)
        r = r + P
    return r


def tf_expm_dynamic(A: np.ndarray, acc: float = 1e-5) -> np.ndarray:
    """Taylor series until max|term| <= acc (propagation.py:658-684)."""
    A = np.asarray(A, dtype=np.complex128)
    r = np.eye(A.shape[0], dtype=A.dtype) + A
    P = A.copy()
    k = 2.0
    while np.max(np.abs(P)) > acc:
        P = (P @ A) / k
        k += 1.0
        r = r + P
    return r


# --------------------------------------------------------------------------
# tf_utils matrix helpers
# --------------------------------------------------------------------------


def compute_folding_stack(n_steps: int) -> List[str]:
    """Level kinds of the pairwise tree (experiment.py:93-107): 'even' / 'odd' per level."""
    stack = []
    n = n_steps
    while n > 1:
        stack.append("even" if n % 2 == 0 else "odd")
        n = int(math.ceil(n / 2))
    return stack


def tf_matmul_n(dUs: np.ndarray, folding_stack: Optional[Sequence] = None) -> np.ndarray:
    """Pairwise-tree ordered product (tf_utils.py:144-193).

    Each level forms new[j] = dU[2j+1] @ dU[2j]; an odd tail element is carried
    unchanged.  Result = dU[N-1] ... dU[1] dU[0].  `folding_stack` is accepted
    for signature parity; the level kind is implied by the current length.
    """
    cur = np.asarray(dUs)
    nlev = len(folding_stack) if folding_stack is not None else None
    lev = 0
    while cur.shape[0] > 1 and (nlev is None or lev < nlev):
        even = cur[0::2]
        odd = cur[1::2]
        prod = odd @ even[: odd.shape[0]]
        if even.shape[0] > odd.shape[0]:
            prod = np.concatenate([prod, even[-1:]], axis=0)
        cur = prod
        lev += 1
    return cur[0]


def tf_matmul_left(dUs: np.ndarray) -> np.ndarray:
    """tf.foldr(matmul) (tf_utils.py:120-129): dU[0] is the right-most factor... i.e.
    foldr(f, [a0..aN-1]) = f(a0, f(a1, ... )) with f(a, x) = a... careful:

    tf.foldr(lambda a, x: matmul(a, x), elems) starts with a = elems[-1] and
    walks x = elems[-2], ..., elems[0], so the result is
    elems[-1] @ elems[-2] @ ... @ elems[0]  (later slice on the LEFT).
    """
    dUs = np.asarray(dUs)
    acc = dUs[-1]
    for k in range(dUs.shape[0] - 2, -1, -1):
        acc = acc @ dUs[k]
    return acc


def tf_matmul_right(dUs: np.ndarray) -> np.ndarray:
    """tf.foldl(matmul) (tf_utils.py:132-141): elems[0] @ elems[1] @ ... @ elems[-1]."""
    dUs = np.asarray(dUs)
    acc = dUs[0]
    for k in range(1, dUs.shape[0]):
        acc = acc @ dUs[k]
    return acc


def Id_like(A: np.ndarray) -> np.ndarray:
    """Identity with A's batch shape (tf_utils.py:240-245)."""
    A = np.asarray(A)
    return np.broadcast_to(np.eye(A.shape[-1], dtype=A.dtype), A.shape).copy()


def tf_kron(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Batched Kronecker product (tf_utils.py:257-267)."""
    A = np.asarray(A)
    B = np.asarray(B)
    res = A[..., :, None, :, None] * B[..., None, :, None, :]
    shp = res.shape[:-4] + (A.shape[-2] * B.shape[-2], A.shape[-1] * B.shape[-1])
    return res.reshape(shp)


def tf_spre(A: np.ndarray) -> np.ndarray:
    """A (x) I  (tf_utils.py:271-274)."""
    return tf_kron(A, Id_like(A))


def tf_spost(A: np.ndarray) -> np.ndarray:
    """I (x) A^T  (tf_utils.py:277-280)."""
    A = np.asarray(A)
    return tf_kron(Id_like(A), np.swapaxes(A, -1, -2))


def tf_super(A: np.ndarray) -> np.ndarray:
    """spre(A) @ spost(A^dagger)  (tf_utils.py:284-289)."""
    A = np.asarray(A)
    return tf_spre(A) @ tf_spost(np.conj(np.swapaxes(A, -1, -2)))


def commutator(A, B):
    """tf_utils.py:562."""
    return A @ B - B @ A


def anticommutator(A, B):
    """tf_utils.py:566."""
    return A @ B + B @ A


def interp_regular_1d_grid(x, x_ref_min, x_ref_max, y_ref):
    """tfp.math.interp_regular_1d_grid(..., fill_value='extrapolate') restated.

    y_ref lives on a regular grid of len(y_ref) points spanning
    [x_ref_min, x_ref_max]; linear interpolation inside, linear extrapolation
    with the edge slope outside.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y_ref, dtype=np.float64)
    ny = y.shape[0]
    span = x_ref_max - x_ref_min
    xi = (ny - 1) * (x - x_ref_min) / span
    xi_c = np.clip(xi, 0.0, ny - 1.0)
    lo = np.floor(xi_c).astype(np.int64)
    hi = np.minimum(lo + 1, ny - 1)
    lo = np.maximum(hi - 1, 0)
    t = xi_c - lo
    out = (1.0 - t) * y[lo] + t * y[hi]
    delta = span / (ny - 1)
    above = x > x_ref_max
    below = x < x_ref_min
    if np.any(above):
        slope = (y[-1] - y[-2]) / delta
        out = np.where(above, y[-1] + slope * (x - x_ref_max), out)
    if np.any(below):
        slope = (y[1] - y[0]) / delta
        out = np.where(below, y[0] + slope * (x - x_ref_min), out)
    return out


RK5_NODES = (0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0)
TSIT5_NODES = (0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0)


def interpolation_times(ts: np.ndarray, interpolate_res: int) -> np.ndarray:
    """Stage-time grid built by interpolate_signal (tf_utils.py:521-556)."""
    ts = np.asarray(ts, dtype=np.float64)
    dt = ts[1] - ts[0]
    if interpolate_res == -1:
        grid = np.sort(np.concatenate([ts + c * dt for c in RK5_NODES]))
    elif interpolate_res == -2:
        grid = np.sort(np.concatenate([ts + c * dt for c in TSIT5_NODES]))
    else:
        grid = np.linspace(ts[0], ts[-1] + dt, ts.shape[0] * interpolate_res + 1)
    return grid


def interpolate_signal(ts, sig, interpolate_res):
    """tf_utils.py:521-559."""
    ts = np.asarray(ts, dtype=np.float64)
    grid = interpolation_times(ts, interpolate_res)
    return interp_regular_1d_grid(grid, ts[0], ts[-1], sig)


# --------------------------------------------------------------------------
# Per-slice propagators
# --------------------------------------------------------------------------


def sum_h0_hks(h0, hks, cflds):
    """H[n] = h0 + sum_k c[k,n] hks[k]  (propagation.py:431-436; model.py:691-697)."""
    h0 = np.asarray(h0, dtype=np.complex128)
    hks = np.asarray(hks, dtype=np.complex128)
    c = np.asarray(cflds).astype(np.complex128)
    if h0.ndim < 3:
        h0 = h0[None]
    return h0 + np.einsum("kn,kij->nij", c, hks)


def tf_propagation_vectorized(h0, hks, cflds_t, dt):
    """dU[n] = expm(-i H[n] dt)  (propagation.py:426-440)."""
    if hks is not None and cflds_t is not None:
        h = sum_h0_hks(h0, hks, cflds_t)
    else:
        h = np.asarray(h0, dtype=np.complex128)
    return expm(-1.0j * h * complex(dt))


def lindblad_dissipator(col_ops) -> np.ndarray:
    """Slice-independent part `clp` of the Lindbladian (propagation.py:570-581)."""
    col_ops = np.asarray(col_ops, dtype=np.complex128)
    L = tf_kron(col_ops, Id_like(col_ops))  # C (x) I
    R = tf_kron(Id_like(col_ops), np.swapaxes(col_ops, -1, -2))  # I (x) C^T
    Rh = np.conj(np.swapaxes(R, -1, -2))
    Lh = np.conj(np.swapaxes(L, -1, -2))
    super_clp = L @ Rh
    anti_l = 0.5 * (Lh @ L)
    anti_r = 0.5 * (R @ Rh)
    return np.sum(super_clp - anti_l - anti_r, axis=0)


def lindblad_generator(h, col_ops) -> np.ndarray:
    """L[n] = -i (H(x)I - I(x)H^T) + clp  (propagation.py:565-582)."""
    h = np.asarray(h, dtype=np.complex128)
    if h.ndim < 3:
        h = h[None]
    eye = Id_like(h)
    lind = -1.0j * (tf_kron(h, eye) - tf_kron(eye, np.swapaxes(h, -1, -2)))
    return lind + lindblad_dissipator(col_ops)[None]


def tf_propagation_lind(h0, hks, col_ops, cflds_t, dt):
    """dU[n] = expm(L[n] dt)  (propagation.py:551-585)."""
    if hks is not None and cflds_t is not None:
        h = sum_h0_hks(h0, hks, cflds_t)
    else:
        h = np.asarray(h0, dtype=np.complex128)
    return expm(lindblad_generator(h, col_ops) * complex(dt))


def tf_dU_of_t(h0, hks, cflds_t, dt):
    """Single-slice legacy form (propagation.py:349-379)."""
    h = np.array(h0, dtype=np.complex128)
    for k in range(len(hks)):
        h = h + complex(cflds_t[k]) * np.asarray(hks[k], dtype=np.complex128)
    return expm(-1.0j * h * complex(dt))


def tf_dU_of_t_lind(h0, hks, col_ops, cflds_t, dt):
    """Single-slice legacy Lindblad form (propagation.py:382-423)."""
    h = np.array(h0, dtype=np.complex128)
    for k in range(len(hks)):
        h = h + complex(cflds_t[k]) * np.asarray(hks[k], dtype=np.complex128)
    lind = -1.0j * (tf_spre(h) - tf_spost(h))
    for c in col_ops:
        c = np.asarray(c, dtype=np.complex128)
        ch = np.conj(c.T)
        lind = lind + tf_spre(c) @ tf_spost(ch)
        lind = lind - 0.5 * (tf_spre(ch) @ tf_spre(c))
        lind = lind - 0.5 * (tf_spost(c) @ tf_spost(ch))
    return expm(lind * complex(dt))


def tf_propagation(h0, hks, cflds, dt):
    """Legacy per-slice loop (propagation.py:518-548) -> list of dU."""
    n = len(cflds[0])
    return [tf_dU_of_t(h0, hks, [f[ii] for f in cflds], dt) for ii in range(n)]


def pwc_trott_drift(h0, hks, cflds_t, dt):
    """Trotterised-drift variant (propagation.py:443-457).

    cflds_t broadcasts against hks as in the reference (`cflds_t * hks`), i.e.
    it must be shaped [K,1,1] for a single slice.
    """
    h0 = np.asarray(h0, dtype=np.complex128)
    hks = np.asarray(hks, dtype=np.complex128)
    c = np.asarray(cflds_t).astype(np.complex128)
    e, v = np.linalg.eigh(h0)
    dE = np.exp(-1.0j * e.real * complex(dt))
    dU0 = v @ np.diag(dE) @ v.T
    ht = np.sum(c * hks, axis=0)
    comm = h0 @ ht - ht @ h0
    dh = -1.0j * ht * complex(dt)
    dcomm = -comm * complex(dt) ** 2 / 2.0
    return dU0 @ expm(dh) @ (dU0 - dcomm)


def tf_batch_propagate(hamiltonian, hks, signals, dt, batch_size, col_ops=None, lindbladian=False):
    """Time-axis chunking (propagation.py:460-515); numerically a no-op."""
    batch_size = int(batch_size)
    outs = []
    if signals is not None:
        signals = np.asarray(signals)
        n = signals.shape[1]
        for i in range(int(math.ceil(n / batch_size))):
            x = signals[:, i * batch_size : (i + 1) * batch_size]
            if lindbladian:
                outs.append(tf_propagation_lind(hamiltonian, hks, col_ops, x, dt))
            else:
                outs.append(tf_propagation_vectorized(hamiltonian, hks, x, dt))
    else:
        hamiltonian = np.asarray(hamiltonian)
        n = hamiltonian.shape[0]
        for i in range(int(math.ceil(n / batch_size))):
            x = hamiltonian[i * batch_size : (i + 1) * batch_size]
            if lindbladian:
                outs.append(tf_propagation_lind(x, None, col_ops, None, dt))
            else:
                outs.append(tf_propagation_vectorized(x, None, None, dt))
    return np.concatenate(outs, axis=0)


# --------------------------------------------------------------------------
# Model-side helpers the path consumes (restated; host-side, O(D^3) once)
# --------------------------------------------------------------------------


def excitation_cutter(dims: Sequence[int], max_excitations: int) -> np.ndarray:
    """0/1 selection matrix (model.py:198-216). Rows = kept product-basis labels."""
    labels = list(np.ndindex(*dims))
    rows = []
    for ii, li in enumerate(labels):
        if sum(li) <= max_excitations:
            line = np.zeros(len(labels))
            line[ii] = 1.0
            rows.append(line)
    return np.array(rows, dtype=np.complex128)


def cut_excitations(op, cutter):
    """model.py:218-220."""
    return cutter @ op @ cutter.T


def blowup_excitations(op, cutter):
    """model.py:222-224."""
    return cutter.T @ op @ cutter


def frame_rotation(num_opers: Sequence[np.ndarray], freqs: Sequence[float], framechanges: Sequence[float], t_final: float, tot_dim: int):
    """FR = expm(i sum_line n_q (w_line T + framechange))  (model.py:536-578)."""
    if len(num_opers) == 0:
        return np.eye(tot_dim, dtype=np.complex128)
    exponent = np.zeros((tot_dim, tot_dim), dtype=np.complex128)
    for n_op, f, fc in zip(num_opers, freqs, framechanges):
        exponent = exponent + 1.0j * np.asarray(n_op, dtype=np.complex128) * (complex(f) * complex(t_final) + complex(fc))
    return expm(exponent)


def dephasing_channel(num_opers: Sequence[np.ndarray], amps: Sequence[float], t_final: float, strength: float, tot_dim: int):
    """Element-wise product of per-line channels (model.py:597-639)."""
    Id = tf_super(np.eye(tot_dim, dtype=np.complex128))
    ch = Id
    for n_op, amp in zip(num_opers, amps):
        Z = tf_super(expm(1.0j * np.asarray(n_op, dtype=np.complex128) * np.pi))
        p = t_final * amp * strength
        if np.real(p) > 1 or np.real(p) < 0:
            raise ValueError(f"Dephasing channel strength {p} is outside [0,1] range")
        ch = ch * ((1 - p) * Id + p * Z)
    return ch


# --------------------------------------------------------------------------
# pwc (propagation.py:258-341) on plain arrays, and through duck-typed objects
# --------------------------------------------------------------------------


def pwc_arrays(
    h0,
    hks,
    signals,
    dt,
    *,
    col_ops=None,
    lindbladian=False,
    batch_size=None,
    cutter=None,
    folding_stack=None,
):
    """The numerical core of `pwc` for one gate: returns {"U", "dUs"}.

    Branch A: h0[D,D], hks[K,D,D], signals[K,N].  Branch B: h0[N,D,D], hks=signals=None.
    Inputs are assumed already cut to the excitation subspace when `cutter` is given
    (model.get_Hamiltonians does the cut, model.py:353-366); col_ops are cut here
    as in propagation.py:319-321.
    """
    if signals is not None:
        n = np.asarray(signals).shape[1]
    else:
        n = np.asarray(h0).shape[0]
    if batch_size is None:
        batch_size = n
    if lindbladian:
        if cutter is not None:
            col_ops = [cutter @ np.asarray(c) @ cutter.T for c in col_ops]
        dUs = tf_batch_propagate(h0, hks, signals, dt, batch_size, col_ops=col_ops, lindbladian=True)
    else:
        dUs = tf_batch_propagate(h0, hks, signals, dt, batch_size)
    U = tf_matmul_n(dUs, folding_stack)
    if cutter is not None:
        U = blowup_excitations(tf_matmul_left(dUs), cutter)
        dUs = np.stack([blowup_excitations(d, cutter) for d in dUs])
    return {"U": U, "dUs": dUs}


def uniform_dt_check(ts_list):
    """The two variance checks of propagation.py:301-308; raises the reference's message."""
    ts_list = np.asarray(ts_list, dtype=np.float64)
    ts = ts_list.mean(axis=0)
    step = ts[1] - ts[0]
    if not np.all(ts_list.var(axis=0) < 1e-5 * step):
        raise Exception("C3Error:Something with the times happend.")
    if not np.all(np.var(ts[1:] - ts[:-1]) < 1e-5 * step):
        raise Exception("C3Error:Something with the times happend.")
    return ts


def pwc(model, gen, instr, folding_stack, batch_size=None) -> Dict:
    """`pwc` (propagation.py:258-341) over duck-typed model/generator objects.

    model: .controllability, .lindbladian, .max_excitations, .ex_cutter,
           .get_Hamiltonians(), .get_Hamiltonian(signal), .get_Lindbladians()
    gen:   .generate_signals(instr) -> {chan: {"values", "ts"}}
    """
    signal = gen.generate_signals(instr)
    if model.controllability:
        h0, hctrls = model.get_Hamiltonians()
        signals, hks, ts = [], [], None
        for key in signal:
            signals.append(np.asarray(signal[key]["values"], dtype=np.float64))
            ts = np.asarray(signal[key]["ts"])
            hks.append(np.asarray(hctrls[key]))
        signals = np.asarray(signals)
        hks = np.asarray(hks, dtype=np.complex128)
    else:
        h0 = model.get_Hamiltonian(signal)
        ts = uniform_dt_check([np.asarray(sig["ts"])[1:] for sig in signal.values()])
        hks, signals = None, None
    dt = ts[1] - ts[0]
    cutter = np.asarray(model.ex_cutter) if model.max_excitations else None
    col_ops = model.get_Lindbladians() if model.lindbladian else None
    res = pwc_arrays(
        h0,
        hks,
        signals,
        dt,
        col_ops=col_ops,
        lindbladian=bool(model.lindbladian),
        batch_size=batch_size,
        cutter=cutter,
        folding_stack=folding_stack,
    )
    res["ts"] = ts
    return res


# --------------------------------------------------------------------------
# ODE state solver (propagation.py:687-904)
# --------------------------------------------------------------------------

solver_slicing = {  # propagation.py:27-32
    "rk4": [2, 3, 2],
    "rk38": [3, 4, 3],
    "rk5": [6, 6, -1],
    "tsit5": [6, 6, -2],
}


def step_schrodinger(psi, h, dt, col=None):
    """propagation.py:897-899."""
    return -1.0j * (h @ psi) * dt


def step_von_neumann(rho, h, dt, col=None):
    """propagation.py:902-904."""
    return -1.0j * commutator(h, rho) * dt


def step_lindblad(rho, h, dt, col):
    """propagation.py:886-894."""
    d = -1.0j * commutator(h, rho)
    for c in col:
        c = np.asarray(c, dtype=np.complex128)
        ch = np.conj(c.T)
        d = d + (c @ rho) @ ch
        d = d - 0.5 * anticommutator(ch @ c, rho)
    return d * dt


step_dict = {"schrodinger": step_schrodinger, "von_neumann": step_von_neumann, "lindblad": step_lindblad}


def rk4(func, rho, h, dt, col=None):
    """propagation.py:755-762."""
    k1 = func(rho, h[0], dt, col)
    k2 = func(rho + k1 / 2.0, h[1], dt, col)
    k3 = func(rho + k2 / 2.0, h[1], dt, col)
    k4 = func(rho + k3, h[2], dt, col)
    return rho + (k1 + 2 * k2 + 2 * k3 + k4) / 6.0


def rk38(func, rho, h, dt, col=None):
    """propagation.py:765-772."""
    k1 = func(rho, h[0], dt, col)
    k2 = func(rho + k1 / 3.0, h[1], dt, col)
    k3 = func(rho + (-k1 / 3.0) + k2, h[2], dt, col)
    k4 = func(rho + k1 - k2 + k3, h[3], dt, col)
    return rho + (k1 + 3 * k2 + 3 * k3 + k4) / 8.0


# Butcher rows as data (Dormand-Prince 5(4), propagation.py:775-823).  Stage 7
# re-uses h[5] and the update weights are the 4th-order embedded row, exactly
# as the reference writes them.
RK5_A = (
    (),
    (1.0 / 5,),
    (3.0 / 40, 9.0 / 40),
    (44.0 / 45, -56.0 / 15, 32.0 / 9),
    (19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729),
    (9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656),
    (35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84),
)
RK5_B = (5179.0 / 57600, 0.0, 7571.0 / 16695, 393.0 / 640, -92097.0 / 339200, 187.0 / 2100, 1.0 / 40)
RK5_H = (0, 1, 2, 3, 4, 5, 5)

# Tsitouras 5(4) rows as the reference writes them (propagation.py:826-883).
TSIT5_A = (
    (),
    (0.161,),
    (-0.008480655492356989, 0.335480655492357),
    (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
    (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
    (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383),
    (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774),
)
TSIT5_B = (
    0.09468075576583945,
    0.009183565540343254,
    0.4877705284247616,
    1.234297566930479,
    -2.7077123499835256,
    1.866628418170587,
    1.0 / 66,
)
TSIT5_H = (0, 1, 2, 3, 4, 5, 5)


def _rk_tableau(func, rho, h, dt, col, A, B, Hidx):
    ks = []
    for stage in range(len(A)):
        y = rho
        for a, k in zip(A[stage], ks):
            if a != 0.0:
                y = y + a * k
        ks.append(func(y, h[Hidx[stage]], dt, col))
    out = rho
    for b, k in zip(B, ks):
        if b != 0.0:
            out = out + b * k
    return out


def rk5(func, rho, h, dt, col=None):
    """propagation.py:775-823."""
    return _rk_tableau(func, rho, h, dt, col, RK5_A, RK5_B, RK5_H)


def tsit5(func, rho, h, dt, col=None):
    """propagation.py:826-883."""
    return _rk_tableau(func, rho, h, dt, col, TSIT5_A, TSIT5_B, TSIT5_H)


solver_dict = {"rk4": rk4, "rk38": rk38, "rk5": rk5, "tsit5": tsit5}


def Hs_of_t_arrays(h0, hks, signals, ts, interpolate_res):
    """Model.Hs_of_t on arrays (model.py:641-697): returns (Hs[r*N(+1),D,D], dt)."""
    ts = np.asarray(ts, dtype=np.float64)
    dt = ts[1] - ts[0]
    sig_i = np.stack([interpolate_signal(ts, s, interpolate_res) for s in signals])
    Hs = sum_h0_hks(h0, hks, sig_i)
    return Hs, dt


def ode_solver_arrays(h0, hks, signals, ts, init_state, solver, step_function, col=None, final_only=False):
    """ode_solver / ode_solver_final_state on arrays (propagation.py:687-752)."""
    if col is not None:
        step_function = "lindblad"
    start, stop, interp = solver_slicing[solver]
    Hs, dt = Hs_of_t_arrays(h0, hks, signals, ts, interp)
    dt = complex(dt)
    fn = solver_dict[solver]
    step = step_dict[step_function]
    state = np.asarray(init_state, dtype=np.complex128)
    n = np.asarray(ts).shape[0]
    states = []
    for i in range(n):
        h = Hs[start * i : start * i + stop]
        state = fn(step, state, h, dt, col=col)
        if not final_only:
            states.append(state)
    if final_only:
        return {"states": state, "ts": np.asarray(ts).astype(np.complex128)}
    return {"states": np.stack(states), "ts": np.asarray(ts).astype(np.complex128)}


def ode_solver(model, gen, instr, init_state, solver, step_function) -> Dict:
    """propagation.py:687-721 over duck-typed objects."""
    return _ode_solver_obj(model, gen, instr, init_state, solver, step_function, False)


def ode_solver_final_state(model, gen, instr, init_state, solver, step_function) -> Dict:
    """propagation.py:724-752."""
    return _ode_solver_obj(model, gen, instr, init_state, solver, step_function, True)


def _ode_solver_obj(model, gen, instr, init_state, solver, step_function, final_only):
    signal = gen.generate_signals(instr)
    col = model.get_Lindbladians() if model.lindbladian else None
    h0, hctrls = model.get_Hamiltonians()
    ts_list, signals, hks = [], [], []
    for key in signal:
        ts_list.append(np.asarray(signal[key]["ts"], dtype=np.float64))
        signals.append(np.asarray(signal[key]["values"], dtype=np.float64))
        hks.append(np.asarray(hctrls[key]))
    ts = uniform_dt_check(ts_list)
    return ode_solver_arrays(h0, np.asarray(hks), signals, ts, init_state, solver, step_function, col=col, final_only=final_only)


# --------------------------------------------------------------------------
# rk4_unitary family (propagation.py:71-101, 221-255)
# --------------------------------------------------------------------------


def rk4_step(h, psi, dt):
    """propagation.py:95-101 (psi is a vector; step = -i dt H psi, :35-36)."""
    f = lambda p, hh: -1.0j * dt * (hh @ p)
    k1 = f(psi, h[0])
    k2 = f(psi + k1 / 2.0, h[1])
    k3 = f(psi + k2 / 2.0, h[1])
    k4 = f(psi + k3, h[2])
    return psi + (k1 + 2 * k2 + 2 * k3 + k4) / 6.0


def gen_du_rk4(h, dt, dim):
    """propagation.py:85-92: rows are the propagated basis vectors (NOT transposed)."""
    rows = []
    for ii in range(dim):
        psi = np.zeros(dim, dtype=np.complex128)
        psi[ii] = 1.0
        rows.append(rk4_step(h, psi, dt))
    return np.stack(rows)


def gen_dus_rk4(h, dt, dim=None):
    """propagation.py:71-82."""
    h = np.asarray(h)
    if dim is None:
        dim = h.shape[1]
    return [gen_du_rk4(h[jj : jj + 3], dt, dim) for jj in range(0, len(h) - 2, 2)]


def gen_u_rk4(h, dt, dim):
    """propagation.py:246-255: columns are the propagated basis vectors."""
    h = np.asarray(h)
    cols = []
    for ii in range(dim):
        psi = np.zeros(dim, dtype=np.complex128)
        psi[ii] = 1.0
        for jj in range(0, len(h) - 2, 2):
            psi = rk4_step(h[jj : jj + 3], psi, dt)
        cols.append(psi)
    return np.stack(cols).T


def rk4_unitary_arrays(Hs, dt, dim, cutter=None):
    """Numerical core of rk4_unitary (propagation.py:221-243) given Hs at prop_res=2."""
    dUs = np.stack(gen_dus_rk4(Hs, dt, dim))
    U = gen_u_rk4(Hs, dt, dim)
    if cutter is not None:
        U = blowup_excitations(U, cutter)
        dUs = np.stack([blowup_excitations(d, cutter) for d in dUs])
    return {"U": U, "dUs": dUs}


# --------------------------------------------------------------------------
# evaluate_sequences (propagation.py:588-627)
# --------------------------------------------------------------------------


def evaluate_sequences(propagators: Dict[str, np.ndarray], sequences: list):
    gates = propagators
    first = list(gates.values())[0]
    dim = first.shape[0]
    out = []
    for seq in sequences:
        if len(seq) == 0:
            out.append(np.eye(dim, dtype=first.dtype))
        else:
            out.append(tf_matmul_left(np.asarray([gates[g] for g in seq], dtype=np.complex128)))
    return out


# --------------------------------------------------------------------------
# Batched conveniences used by tests/bench (loop over independent samples)
# --------------------------------------------------------------------------


def propagate_batch(h0, hks, signals_b, dt, *, col_ops=None, lindbladian=False, fr_phase=None):
    """U[b] for independent samples: signals_b[B,K,N]; the reference has B=1 and
    loops in Python (optimalcontrol_robust.py:54-63).  Optional frame-rotation
    row phases fr_phase[B,D] (U <- diag(exp(i*phase)) U, experiment.py:482-509)."""
    signals_b = np.asarray(signals_b)
    outs = []
    for b in range(signals_b.shape[0]):
        r = pwc_arrays(h0, hks, signals_b[b], dt, col_ops=col_ops, lindbladian=lindbladian)
        U = r["U"]
        if fr_phase is not None:
            ph = np.exp(1.0j * np.asarray(fr_phase[b]))
            if lindbladian:
                ph = np.kron(ph, np.conj(ph))
            U = ph[:, None] * U
        outs.append(U)
    return np.stack(outs)


def algorithmic_flops_per_slice(D: int, K: int, order: int, squarings: int) -> float:
    """F_slice = 8 D^3 (pi_m + s + 1) + (32/3) D^3 + 4 K D^2."""
    return 8.0 * D**3 * (PADE_PRODUCTS[order] + squarings + 1) + (32.0 / 3.0) * D**3 + 4.0 * K * D**2


# --------------------------------------------------------------------------
# Fidelity epilogue: c3/libraries/fidelities.py:154-218,290-347 and the
# helpers tf_project_to_comp / tf_unitary_overlap / tf_average_fidelity (tf_utils.py:330-436),
# projector / pauli_basis (qt_utils.py:10-56,178-193) -- restated literally.
# --------------------------------------------------------------------------

_PAULIS = (
    np.array([[1, 0], [0, 1]], dtype=np.complex128),
    np.array([[0, 1], [1, 0]], dtype=np.complex128),
    np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    np.array([[1, 0], [0, -1]], dtype=np.complex128),
)


def _kron_n(mats):
    out = np.eye(1)
    for m in mats:
        out = np.kron(out, m)
    return out


def projector(dims, indices, outdims=None):
    """qt_utils.py:178-193: selected subspaces keep their lowest two states, the rest the lowest one."""
    if outdims is None:
        outdims = [2] * len(dims)
    ids = []
    for index, dim in enumerate(dims):
        ids.append(np.eye(dim, outdims[index]) if index in indices else np.eye(dim, 1))
    return _kron_n(ids)


def tf_project_to_comp(A, dims, index=None, to_super=False):
    """tf_utils.py:428-436: P^T A P."""
    if not index:
        index = list(range(len(dims)))
    proj = projector(dims, index)
    if to_super:
        proj = np.kron(proj, proj)
    P = proj.astype(np.asarray(A).dtype)
    return P.T @ np.asarray(A) @ P


def tf_unitary_overlap(A, B, lvls=None):
    """tf_utils.py:330-366: |tr(A B^+)/lvls|^2."""
    if lvls is None:
        lvls = B.shape[0]
    return np.abs(np.trace(A @ np.conj(B.T)) / lvls) ** 2


def unitary_infid(ideal, actual, index=None, dims=None):
    """fidelities.py:154-184."""
    if index is None:
        index = list(range(len(dims)))
    actual_comp = tf_project_to_comp(actual, dims=dims, index=index)
    return 1 - tf_unitary_overlap(actual_comp, ideal, lvls=2 ** len(index))


def tf_superoper_unitary_overlap(A, B, lvls=None):
    """tf_utils.py:369-377: |sqrt(tr(A B^+)) / lvls|^2."""
    if lvls is None:
        lvls = np.sqrt(B.shape[0])
    return np.abs(np.sqrt(np.trace(A @ np.conj(B.T)) + 0j) / lvls) ** 2


def lindbladian_unitary_infid(ideal, actual, index=(0,), dims=(2,)):
    """fidelities.py:221-249: the unitary overlap of the projected superoperator with tf_super(ideal)."""
    index = list(index)
    U_ideal = tf_super(np.asarray(ideal))
    actual_comp = tf_project_to_comp(actual, dims=dims, index=index, to_super=True)
    return 1 - tf_superoper_unitary_overlap(actual_comp, U_ideal, lvls=2 ** len(index))


def pauli_basis(dims=(2,)):
    """qt_utils.py:10-44."""
    paulis = []
    for dim in dims:
        padded = []
        for P in _PAULIS:
            o_ = np.zeros((dim, dim), dtype=np.complex128)
            o_[:2, :2] = P
            padded.append(o_)
        paulis.append(padded)
    result = [[]]
    for pauli_set in paulis:
        result = [x + [y] for x in result for y in pauli_set]
    size = int(np.prod(np.array(dims) ** 2))
    B = np.zeros((size, size), dtype=complex)
    for idx, op_tuple in enumerate(result):
        op = _kron_n(op_tuple)
        vec = np.reshape(np.transpose(op), [-1, 1])
        B[:, idx] = vec.T.conj()
    return B


def super_to_choi(A):
    """tf_utils.py:416-425."""
    n = int(np.sqrt(A.shape[0]))
    return np.reshape(np.transpose(np.reshape(A, [n] * 4), (3, 1, 2, 0)), A.shape)


def tf_choi_to_chi(U, dims=None):
    """tf_utils.py:404-412."""
    if dims is None:
        dims = [int(np.sqrt(U.shape[0]))]
    B = pauli_basis([2] * len(dims))
    return np.conj(B.T) @ U @ B


def tf_super_to_fid(err, lvls):
    """tf_utils.py:396-401."""
    lambda_chi = tf_choi_to_chi(super_to_choi(err), dims=lvls)
    d = 2 ** len(lvls)
    return np.abs((lambda_chi[0, 0] / d + 1) / (d + 1))


def tf_average_fidelity(A, B, lvls=None):
    """tf_utils.py:380-385."""
    if lvls is None:
        lvls = [B.shape[0]]
    Lambda = np.conj(A.T) @ B
    return tf_super_to_fid(tf_super(Lambda), lvls)


def average_infid(ideal, actual, index=(0,), dims=(2,)):
    """fidelities.py:290-313."""
    index = list(index)
    actual_comp = tf_project_to_comp(actual, dims=list(dims), index=index)
    return 1 - tf_average_fidelity(actual_comp, ideal, lvls=[2] * len(index))


def unitary_infid_set(propagators: Dict, ideals: Dict, index, dims):
    """fidelities.py:187-218 with the ideal gates given directly (instructions[gate].get_ideal_gate)."""
    return float(np.mean([unitary_infid(ideals[g], U, index, dims) for g, U in propagators.items()]))


def average_infid_set(propagators: Dict, ideals: Dict, index, dims):
    """fidelities.py:316-347."""
    return float(np.mean([average_infid(ideals[g], U, index, dims) for g, U in propagators.items()]))


# --------------------------------------------------------------------------
# Signal synthesis for the standard line LO + AWG -> DAC -> Mixer -> VoltsToHertz
# (the step immediately before the propagator path)
# --------------------------------------------------------------------------

ENV_NO_DRIVE = 0  # envelopes.py:26-29
ENV_RECT = 1  # envelopes.py:195-198
ENV_GAUSSIAN_NONORM = 2  # envelopes.py:470-487
ENV_FLATTOP = 3  # envelopes.py:254-279
ENV_FLATTOP_RISEFALL = 4  # envelopes.py:228-251
ENV_COSINE = 5  # envelopes.py:421-438
ENV_GAUSSIAN_SIGMA = 6  # envelopes.py:374-398 (area-normalised, offset removed)
ENV_GAUSSIAN = 7  # envelopes.py:401-418 (gaussian_sigma with sigma = t_final / 6)
ENV_TRAPEZOID = 8  # envelopes.py:201-225


def create_ts(t_start: float, t_end: float, resolution: float) -> np.ndarray:
    """Centred sample times of a device (devices.py:72-122): `int(|t_end - t_start| * res)` samples,
    `linspace(t_start + dt/2, t_end - dt/2, num)`."""
    num = int(np.abs(t_start - t_end) * resolution)
    dt = 1.0 / resolution
    return np.linspace(t_start + dt / 2, t_end - dt / 2, num)


def _sigmoid(x):
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-x))


def envelope_shape(shape: int, t, p: Dict) -> np.ndarray:
    """Real part of the envelope library functions named above."""
    t = np.asarray(t, dtype=np.float64)
    if shape == ENV_NO_DRIVE:
        return np.zeros_like(t)
    if shape == ENV_RECT:
        return np.ones_like(t)
    if shape == ENV_GAUSSIAN_NONORM:
        return np.exp(-((t - p["t_final"] / 2) ** 2) / (2 * p["sigma"] ** 2))
    if shape in (ENV_FLATTOP, ENV_FLATTOP_RISEFALL):
        from scipy.special import erf

        rf = p["risefall"]
        t_up, t_down = (p["t_up"], p["t_down"]) if shape == ENV_FLATTOP else (rf, p["t_final"] - rf)
        return (1 + erf((t - t_up) / rf)) / 2 * (1 + erf((-t + t_down) / rf)) / 2
    if shape == ENV_COSINE:
        return 0.5 * (1 - np.cos(2 * np.pi * t / p["t_final"]))
    if shape in (ENV_GAUSSIAN_SIGMA, ENV_GAUSSIAN):
        from scipy.special import erf

        T = p["t_final"]
        sigma = p["sigma"] if shape == ENV_GAUSSIAN_SIGMA else T / 6
        gauss = np.exp(-((t - T / 2) ** 2) / (2 * sigma**2))
        offset = np.exp(-(T**2) / (8 * sigma**2))
        norm = np.sqrt(2 * np.pi * sigma**2) * erf(T / (np.sqrt(8) * sigma)) - T * offset
        return (gauss - offset) / norm
    if shape == ENV_TRAPEZOID:
        rf, T = p["risefall"], p["t_final"]
        env = np.ones_like(t)
        env = np.where(t <= rf * 2.5, t / (rf * 2.5), env)
        env = np.where(t >= T - rf * 2.5, (T - t) / (rf * 2.5), env)
        return env
    raise ValueError(f"unknown envelope shape {shape}")


def envelope_shape_der(shape: int, t, p: Dict) -> np.ndarray:
    """d(shape)/dt -- what the reference's GradientTape yields for these closed forms (pulse.py:171-180)."""
    t = np.asarray(t, dtype=np.float64)
    if shape in (ENV_NO_DRIVE, ENV_RECT):
        return np.zeros_like(t)
    if shape == ENV_GAUSSIAN_NONORM:
        return -(t - p["t_final"] / 2) / p["sigma"] ** 2 * envelope_shape(shape, t, p)
    if shape in (ENV_FLATTOP, ENV_FLATTOP_RISEFALL):
        from scipy.special import erf

        rf = p["risefall"]
        t_up, t_down = (p["t_up"], p["t_down"]) if shape == ENV_FLATTOP else (rf, p["t_final"] - rf)
        u, d = (t - t_up) / rf, (-t + t_down) / rf
        g = lambda x: 2 / np.sqrt(np.pi) * np.exp(-x * x) / rf
        return (g(u) * (1 + erf(d)) - (1 + erf(u)) * g(d)) / 4
    if shape == ENV_COSINE:
        w = 2 * np.pi / p["t_final"]
        return 0.5 * w * np.sin(w * t)
    if shape in (ENV_GAUSSIAN_SIGMA, ENV_GAUSSIAN):
        from scipy.special import erf

        T = p["t_final"]
        sigma = p["sigma"] if shape == ENV_GAUSSIAN_SIGMA else T / 6
        offset = np.exp(-(T**2) / (8 * sigma**2))
        norm = np.sqrt(2 * np.pi * sigma**2) * erf(T / (np.sqrt(8) * sigma)) - T * offset
        return -(t - T / 2) / sigma**2 * np.exp(-((t - T / 2) ** 2) / (2 * sigma**2)) / norm
    if shape == ENV_TRAPEZOID:
        rf, T = p["risefall"], p["t_final"]
        d = np.zeros_like(t)
        d = np.where(t <= rf * 2.5, 1.0 / (rf * 2.5), d)
        d = np.where(t >= T - rf * 2.5, -1.0 / (rf * 2.5), d)
        return d
    raise ValueError(f"unknown envelope shape {shape}")


def envelope_mask(ts, t_final: float, t_window: float) -> np.ndarray:
    """Envelope.compute_mask (pulse.py:88-115): 1 inside [0, 0.999 min(t_final, window)), 0 outside,
    built from two saturated sigmoids."""
    tf_ = min(t_final, t_window)
    dt = ts[1] - ts[0]
    return _sigmoid((ts / dt + 0.001) * 1e6) * _sigmoid((0.999 * tf_ - ts) / dt * 1e6)


def envelope_values(comp: Dict, ts_off: np.ndarray, t_window: float) -> np.ndarray:
    """Complex envelope samples of one component (pulse.py:117-143 `_get_shape_values_before/_just`;
    pulse.py:171-180 for the DRAG variant: imag = -delta * dt * d env/dt by autodiff, which also
    propagates through `t_before = 2 ts[0] - ts[1]` into samples 0 and 1)."""
    shape = comp["shape"]
    mask = envelope_mask(ts_off, comp.get("t_final", t_window), t_window)
    offset = 0.0
    if comp.get("use_t_before", False):
        t_before = 2 * ts_off[0] - ts_off[1]
        offset = envelope_shape(shape, t_before, comp)
    env = mask * (envelope_shape(shape, ts_off, comp) - offset)
    if not comp.get("drag", False):
        return env.astype(np.complex128)
    dt = ts_off[1] - ts_off[0]
    # d(sum env)/d ts: the mask is saturated (zero slope); the offset term couples samples 0 and 1
    denv = mask * envelope_shape_der(shape, ts_off, comp)
    if comp.get("use_t_before", False):
        doff = float(envelope_shape_der(shape, t_before, comp)) * mask.sum()
        denv[0] -= 2 * doff
        denv[1] += doff
    return env - 1j * denv * dt * comp.get("delta", 0.0)


def awg_iq(components: Sequence[Dict], ts: np.ndarray, t_start: float):
    """Instruction.get_awg_signal (gates.py:341-370): sum_e amp_e env_e exp(i (xy_e - freq_offset_e t))."""
    sig = np.zeros(ts.shape, dtype=np.complex128)
    for comp in components:
        t0 = t_start + comp.get("delay", 0.0)
        t_window = comp["t_final"] if "t_final" in comp else np.inf
        ts_off = ts - t0
        phase = comp.get("xy_angle", 0.0) - comp.get("freq_offset", 0.0) * ts_off
        sig = sig + comp["amp"] * envelope_values(comp, ts_off, t_window) * np.exp(1j * phase)
    return sig.real, sig.imag


def dac_nearest(x: np.ndarray, new_dim: int) -> np.ndarray:
    """DigitalToAnalog (devices.py:306-351): nearest-neighbour upsampling with half-pixel centres
    (`tf.image.resize(..., method="nearest")`: source index floor((i + 0.5) old/new)).  The reference's
    current code path passes through float32 inside `tf.image.resize`; its stored golden signals
    (test/two_qubit_data.pickle, test/tunable_coupler_data.pickle) were produced in float64 and are
    matched to 1e-15 by this float64 restatement (the float32 path differs by 4e-8 relative, inside
    the rtol=1e-7 of test/test_two_qubits.py:22-33)."""
    old = x.shape[-1]
    idx = np.minimum(np.floor((np.arange(new_dim) + 0.5) * (old / new_dim)).astype(np.int64), old - 1)
    return x[..., idx]


def generate_signal(components: Sequence[Dict], lo_freq: float, v_to_hz: float, t_start: float, t_end: float, awg_res: float, sim_res: float):
    """One drive line: LO (devices.py:1073-1130, noiseless branch), AWG (devices.py:1157-1195), DAC,
    Mixer `I cos + Q sin` (devices.py:914-939), VoltsToHertz (devices.py:203-221).
    Returns {"values" [N], "ts" [N], "inphase"/"quadrature" at AWG resolution}."""
    ts_awg = create_ts(t_start, t_end, awg_res)
    ts = create_ts(t_start, t_end, sim_res)
    inph, quad = awg_iq(components, ts_awg, t_start)
    I, Q = dac_nearest(inph, ts.shape[0]), dac_nearest(quad, ts.shape[0])
    values = (np.cos(lo_freq * ts) * I + np.sin(lo_freq * ts) * Q) * v_to_hz
    return {"values": values, "ts": ts, "inphase": inph, "quadrature": quad, "ts_awg": ts_awg}


# --------------------------------------------------------------------------
# Gradient of the PWC propagator w.r.t. the control samples
# --------------------------------------------------------------------------


def expm_frechet(X: np.ndarray, E: np.ndarray) -> np.ndarray:
    """L(X, E) = d/de exp(X + e E)|_0 as the (0,1) block of exp([[X, E], [0, X]]) -- the exact derivative
    the reference obtains by differentiating `tf.linalg.expm` with a GradientTape
    (c3/optimizers/optimizer.py:206-216 wraps the goal function; propagation.py:440 is on the tape)."""
    D = X.shape[-1]
    aug = np.zeros((2 * D, 2 * D), dtype=np.complex128)
    aug[:D, :D] = X
    aug[D:, D:] = X
    aug[:D, D:] = E
    return expm(aug)[:D, D:]


def pwc_signal_gradient(h0, hks, signals, dt, Ubar, fr_phase=None) -> np.ndarray:
    """d loss / d signals[k, n] for one sample, where d loss = Re sum_ij conj(Ubar_ij) dU_ij and
    U = FR . dU_{N-1} ... dU_0, dU_n = exp(-i dt (h0 + sum_k c_k(n) hk)) (propagation.py:426-440,
    tf_utils.py:144-193, experiment.py:482-509).  Direct evaluation: prefix / suffix products and one
    Frechet derivative per (k, n)."""
    K, N = signals.shape
    D = h0.shape[-1]
    Xs = [-1j * dt * (h0 + sum(signals[k, n] * hks[k] for k in range(K))) for n in range(N)]
    dUs = [expm(X) for X in Xs]
    P = [np.eye(D, dtype=np.complex128)]
    for n in range(N):
        P.append(dUs[n] @ P[-1])
    lam = Ubar.astype(np.complex128)
    if fr_phase is not None:
        lam = np.exp(-1j * np.asarray(fr_phase))[:, None] * lam  # FR^H Ubar
    g = np.zeros((K, N))
    for n in range(N - 1, -1, -1):
        W = lam @ P[n].conj().T  # cotangent of dU_n
        for k in range(K):
            g[k, n] = np.real(np.vdot(W, expm_frechet(Xs[n], -1j * dt * hks[k])))
        lam = dUs[n].conj().T @ lam
    return g


def pwc_per_slice_hamiltonian_cotangents(Hs, dt, Ubar, fr_phase=None) -> np.ndarray:
    """Branch B (propagation.py:295-308): cotangents of the per-slice Hamiltonians, H_bar[n] with
    d loss = Re sum conj(H_bar[n]) dH[n] for d loss = Re sum conj(Ubar) dU, U = diag(e^{i phase}) prod_n exp(-i H_n dt).
    H_bar[n] = conj(-i dt) L(X_n^H)[W_n], W_n the cotangent of the slice exponential (L = Frechet derivative of exp).
    Pinned by finite differences of the pinned propagator oracle (tests/test_gradient.py)."""
    Hs = np.asarray(Hs, dtype=np.complex128)
    N, D = Hs.shape[0], Hs.shape[-1]
    Xs = [-1j * dt * Hs[n] for n in range(N)]
    Es = [expm(X) for X in Xs]
    pre = [np.eye(D, dtype=np.complex128)]
    for n in range(N):
        pre.append(Es[n] @ pre[-1])
    post = [None] * N
    acc = np.eye(D, dtype=np.complex128)
    for n in range(N - 1, -1, -1):
        post[n] = acc
        acc = acc @ Es[n]
    ph = np.exp(1j * np.asarray(fr_phase)) if fr_phase is not None else np.ones(D)
    out = np.zeros((N, D, D), dtype=np.complex128)
    for n in range(N):
        W = (post[n].conj().T * np.conj(ph)[None, :]) @ np.asarray(Ubar) @ pre[n].conj().T
        out[n] = (1j * dt) * expm_frechet(Xs[n].conj().T, W)
    return out


def pwc_lindblad_signal_gradient(h0, hks, col_ops, signals, dt, Ubar, fr_phase=None) -> np.ndarray:
    """d loss / d signals[k, n] for loss with d loss = Re sum conj(Ubar) dU, U the Lindblad superoperator of
    propagation.py:551-585 (what the reference obtains by taping tf_propagation_lind, optimizer.py:206-216): one
    Frechet derivative of the slice exponential per (k, n), no unitarity assumed.  `fr_phase` [D^2]: row phases applied
    to U.  Pinned by central finite differences of the pinned propagator oracle (tests/test_gradient.py)."""
    h0 = np.asarray(h0, dtype=np.complex128)
    hks = np.asarray(hks, dtype=np.complex128)
    signals = np.asarray(signals, dtype=np.float64)
    K, N = signals.shape
    D = h0.shape[-1]
    I = np.eye(D)
    clp = lindblad_dissipator(col_ops)
    sup = lambda h: -1j * (np.kron(h, I) - np.kron(I, h.T))
    Gk = [sup(hks[k]) * dt for k in range(K)]
    Xs = [(sup(h0 + sum(signals[k, n] * hks[k] for k in range(K))) + clp) * dt for n in range(N)]
    Es = [expm(X) for X in Xs]
    Dm = D * D
    pre = [np.eye(Dm, dtype=np.complex128)]  # pre[n] = E_{n-1} ... E_0
    for n in range(N):
        pre.append(Es[n] @ pre[-1])
    post = [None] * N  # post[n] = E_{N-1} ... E_{n+1}
    acc = np.eye(Dm, dtype=np.complex128)
    for n in range(N - 1, -1, -1):
        post[n] = acc
        acc = acc @ Es[n]
    ph = np.exp(1j * np.asarray(fr_phase)) if fr_phase is not None else np.ones(Dm)
    grad = np.zeros((K, N))
    for n in range(N):
        W = (post[n].conj().T * np.conj(ph)[None, :]) @ np.asarray(Ubar) @ pre[n].conj().T  # cotangent of E_n
        for k in range(K):
            grad[k, n] = np.real(np.sum(np.conj(W) * expm_frechet(Xs[n], Gk[k])))
    return grad


def unitary_infid_cotangent(ideal, U, index, dims):
    """Ubar for loss = unitary_infid (fidelities.py:154-184): 1 - |tr(G^+ P^T U P) / L|^2."""
    Pm = projector(dims, index)
    L = Pm.shape[1]
    s = np.trace(ideal.conj().T @ (Pm.T @ U @ Pm))
    return -(2.0 / L**2) * s * (Pm @ ideal @ Pm.T)


def generate_signal_vjp(components: Sequence[Dict], lo_freq: float, v_to_hz: float, t_start: float, t_end: float, awg_res: float, sim_res: float, gsig: np.ndarray):
    """Vector-Jacobian product of `generate_signal` for the commonly optimised pulse parameters
    (opt_map entries of the reference's examples: amp, xy_angle, freq_offset, delta) and the carrier.
    `gsig` = d loss / d values [N].  Returns (list of dicts per component, {"lo_freq", "v_to_hz"}).
    The reference gets these from the same GradientTape that covers the propagation
    (optimizers/optimizer.py:206-216; Instruction.get_awg_signal gates.py:341-370 is on the tape)."""
    ts_awg = create_ts(t_start, t_end, awg_res)
    ts = create_ts(t_start, t_end, sim_res)
    N, Na = ts.shape[0], ts_awg.shape[0]
    idx = np.minimum(np.floor((np.arange(N) + 0.5) * (Na / N)).astype(np.int64), Na - 1)
    cs, sn = np.cos(lo_freq * ts), np.sin(lo_freq * ts)
    inph, quad = awg_iq(components, ts_awg, t_start)
    gI = np.zeros(Na)
    gQ = np.zeros(Na)
    np.add.at(gI, idx, gsig * cs * v_to_hz)
    np.add.at(gQ, idx, gsig * sn * v_to_hz)
    I, Q = inph[idx], quad[idx]
    gcar = {"lo_freq": float(np.sum(gsig * v_to_hz * ts * (-sn * I + cs * Q))), "v_to_hz": float(np.sum(gsig * (cs * I + sn * Q)))}
    out = []
    for comp in components:
        t0 = t_start + comp.get("delay", 0.0)
        ts_off = ts_awg - t0
        ph = np.exp(1j * (comp.get("xy_angle", 0.0) - comp.get("freq_offset", 0.0) * ts_off))
        env = envelope_values(comp, ts_off, comp["t_final"] if "t_final" in comp else np.inf)
        z = comp["amp"] * env * ph
        g = {"amp": float(np.sum(gI * (env * ph).real + gQ * (env * ph).imag)), "xy_angle": float(np.sum(-gI * z.imag + gQ * z.real)),
             "freq_offset": float(np.sum(ts_off * (gI * z.imag - gQ * z.real))), "delta": 0.0}
        if comp.get("drag", False):
            env0 = envelope_values(dict(comp, delta=0.0), ts_off, comp["t_final"])
            env1 = envelope_values(dict(comp, delta=1.0), ts_off, comp["t_final"])
            dz = comp["amp"] * (env1 - env0) * ph  # the envelope is affine in delta
            g["delta"] = float(np.sum(gI * dz.real + gQ * dz.imag))
        out.append(g)
    return out, gcar
