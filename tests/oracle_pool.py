"""Process-parallel oracle for the full-size spot checks: the GPU box has hundreds of host threads, one oracle
propagator of cfg4 / cfg5 takes ~1 s on one of them (test infrastructure only)."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init():
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)


def _one(job):
    h0, hks, sig, dt, col, lind, ph = job
    from oracle import c3_oracle as o

    return o.propagate_batch(h0, hks, sig[None], dt, col_ops=col, lindbladian=lind, fr_phase=None if ph is None else ph[None])[0]


def propagate_samples(h0, hks, signals, dt, *, col_ops=None, lindbladian=False, fr_phase=None, workers=None):
    """oracle.propagate_batch over the samples of `signals` [n,K,N], one process per sample (spawn: the parent may hold a
    HIP context)."""
    import numpy as np

    n = int(signals.shape[0])
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    workers = max(1, min(n, workers or 32, avail))
    jobs = [(h0, hks, signals[i], dt, col_ops, lindbladian, None if fr_phase is None else fr_phase[i]) for i in range(n)]
    if workers == 1:
        _init()
        return np.stack([_one(j) for j in jobs])
    with mp.get_context("spawn").Pool(workers, initializer=_init) as pool:
        return np.stack(pool.map(_one, jobs, chunksize=1))


def _one_ode(job):
    h0, hks, sig, ts, init, solver, step, col = job
    from oracle import c3_oracle as o

    return o.ode_solver_arrays(h0, hks, sig, ts, init, solver, step, col=col, final_only=True)["states"]


def ode_final_states(h0, hks, signals, ts, init, solver, step, *, col_ops=None, workers=None):
    """oracle.ode_solver_arrays(final_only) over the samples of `signals` [n,K,N], one process per sample."""
    import numpy as np

    n = int(signals.shape[0])
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    workers = max(1, min(n, workers or 32, avail))
    jobs = [(h0, hks, signals[i], ts, init, solver, step, col_ops) for i in range(n)]
    if workers == 1:
        _init()
        return np.stack([_one_ode(j) for j in jobs])
    with mp.get_context("spawn").Pool(workers, initializer=_init) as pool:
        return np.stack(pool.map(_one_ode, jobs, chunksize=1))
