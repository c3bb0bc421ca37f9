"""Round-3 GPU parity tests (through the C ABI): regressions for the round-2 advisor findings, the lane-row ODE
kernels of c3p_ode_row.hip, and the wider full-size spot checks."""
import os

import numpy as np
import pytest

from c3_amd import _lib, workloads
from oracle import c3_oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import _lib, propagation

    _lib.require_gpu()
    return propagation


def fro_max(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return max(np.linalg.norm(a[i] - b[i]) for i in range(a.shape[0]))


# --------------------------------------------------------------------------
# the ordered combine of the register-resident / arena kernels' segment products must not
# write into the buffer it reads (uneven generic segments: the last one holds a single matrix)
# --------------------------------------------------------------------------


@pytest.mark.parametrize("B,N,no_regd", [(1, 800, False), (3, 410, False), (16, 130, False), (2, 400, True)])
def test_segment_combine_does_not_alias_its_input(prop, B, N, no_regd):
    """propagation.py:551-585 + tf_utils.py:144-193 at cfg4's operators with few samples and many time segments."""
    wl = workloads.make_workload(4, B=B, N=N)
    if no_regd:
        _lib.set_option("no_regd", "1")
    try:
        r = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
        again = prop.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    finally:
        _lib.set_option("no_regd", None)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    assert fro_max(r["U"], ref) < TOL
    assert np.array_equal(np.asarray(r["U"]), np.asarray(again["U"]))


# --------------------------------------------------------------------------
# lane-row ODE kernels (c3p_ode_row.hip): SURVEY 8a rows a11 - a14
# --------------------------------------------------------------------------


def _rand_herm(rng, D, scale, real=False):
    a = rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))
    return (scale * (a + a.conj().T) / 2).astype(complex)


def _ode_problem(D, K, B, N, real, seed):
    rng = np.random.default_rng(seed)
    h0 = _rand_herm(rng, D, 0.3, real)
    hks = np.stack([_rand_herm(rng, D, 0.2, real) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ts = (np.arange(N) + 0.5) * 0.05
    return h0, hks, sig, ts


@pytest.mark.parametrize("D,K", [(2, 1), (3, 2), (4, 1), (5, 3), (6, 2), (7, 4), (9, 2), (11, 1), (12, 3), (13, 2), (16, 4)])
@pytest.mark.parametrize("real", [False, True])
def test_ode_row_schrodinger_dimensions(prop, D, K, real):
    """propagation.py:687-752 + :897-899 on every padded-dimension class of the lane-row kernel, B not a multiple of the
    four samples per wavefront, N not a multiple of the 14-step signal chunk, trajectory and final state."""
    from c3_amd import _lib

    B, N = 7, 33
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 100 * D + K)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "tsit5"):
        out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert _lib.last_kernel() == "ode_row"
        fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
        for b in range(B):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger")["states"]
            assert np.abs(out[b] - ref).max() < 1e-11
            assert np.abs(fin[b] - ref[-1]).max() < 1e-11


@pytest.mark.parametrize("solver", ["rk4", "rk38", "rk5", "tsit5"])
@pytest.mark.parametrize("D,K,C", [(3, 1, 1), (4, 2, 2), (6, 3, 0), (9, 2, 2), (12, 4, 3), (16, 2, 1)])
def test_ode_row_density_matrices(prop, solver, D, K, C):
    """von_neumann (:902-904) for C = 0 / real operators, lindblad (:886-894) with C collapse operators."""
    from c3_amd import _lib

    B, N = 5, 19
    h0, hks, sig, ts = _ode_problem(D, K, B, N, C == 0 and D % 2 == 0, 7 * D + C)
    rng = np.random.default_rng(D + C)
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    rho = a @ a.conj().T
    rho /= np.trace(rho)
    col = np.stack([0.4 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)]) if C else None
    step = "lindblad" if C else "von_neumann"
    out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col))
    assert _lib.last_kernel() == "ode_row"
    fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col, final_only=True))
    for b in range(B):
        ref = o.ode_solver_arrays(h0, hks, sig[b], ts, rho, solver, step, col=col)["states"]
        assert np.abs(out[b] - ref).max() < 1e-11
        assert np.abs(fin[b] - ref[-1]).max() < 1e-11


def test_ode_row_matches_workgroup_kernel_and_cfg2(prop):
    """Same arithmetic as the round-1 workgroup-per-sample kernel (C3P_ODE_WG=1 selects it) on cfg2's operators at
    N = 1000: identical to rounding; trajectory against the oracle on three samples."""
    from c3_amd import _lib

    wl = workloads.make_workload(2, B=37)
    psi = np.zeros((wl.D, 1), complex)
    psi[0, 0] = 1.0
    new = {s: np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, psi, s, "schrodinger")) for s in ("rk4", "rk5")}
    assert _lib.last_kernel() == "ode_row"
    _lib.set_option("ode_wg", "1")
    try:
        old = {s: np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, psi, s, "schrodinger")) for s in ("rk4", "rk5")}
        assert _lib.last_kernel() == "ode_wg"
    finally:
        _lib.set_option("ode_wg", None)
    for s in new:
        assert np.abs(new[s] - old[s]).max() < 1e-12
    for b in (0, 17, 36):
        ref = o.ode_solver_arrays(wl.h0, wl.hks, wl.signals[b], wl.ts, psi, "rk4", "schrodinger")["states"]
        assert np.abs(new["rk4"][b] - ref).max() < 1e-11


def test_ode_row_rk4_unitary(prop):
    """rk4_unitary (propagation.py:71-101,221-255): the columns of the propagator as B x D vector problems."""
    from c3_amd import _lib

    D, K, B = 6, 2, 3
    Ns = 41
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, False, 5)
    lib = _lib.load()
    import ctypes

    U = np.zeros((B, D, D), complex)
    dUs = np.zeros((B, (Ns - 1) // 2, D, D), complex)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.1, B, K, Ns, D, 1, p(U), p(dUs), None)
    assert rc == 0 and _lib.last_kernel() == "ode_row"
    for b in range(B):
        Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
        ref = o.rk4_unitary_arrays(Hs, 0.1, D)
        assert np.abs(U[b] - ref["U"]).max() < 1e-12
        assert np.abs(dUs[b] - ref["dUs"]).max() < 1e-12


# --------------------------------------------------------------------------
# multi-GPU readiness without the 8-GPU node: the RCCL paths with one rank under torch.distributed.run
# --------------------------------------------------------------------------
import json
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_rccl_one_rank_sharded_propagation(lib):
    """c3_amd.dist over backend "nccl" (= RCCL): communicator, all_gather_into_tensor of the slabs, all-reduce of the goal."""
    from c3_amd import _lib

    _lib.require_gpu()
    out = _torchrun([os.path.join(ROOT, "tests", "checks", "dist_nccl_check.py")], 29541)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "DIST_NCCL_OK world=1" in out.stdout


@pytest.mark.parametrize("extra", [[], ["--gather-every", "32"], ["--exchange", "goal"], ["--scaling", "strong", "--batch", "64"], ["--overlap-gather", "--steps", "7"]])
def test_bench_under_torchrun_one_rank(lib, extra):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, RCCL), with --check.  The DEFAULT
    schedule is north_star's: ONE all-gather of U per step; the amortised schedule (one all-gather per 32 steps) and the
    gather-free goal exchange are timed beside it and arrive as extra keys of the same line.  stdout is exactly one JSON
    line: RCCL's version banner (printed from C at communicator set-up) goes to stderr."""
    from c3_amd import _lib

    _lib.require_gpu()
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--check", "--no-cpu-baseline",
                     "--no-e2e", "--ramp-ms", "0"] + extra, 29543 + len(extra))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "RCCL world size 1" in out.stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["max_fro_err_vs_oracle"] < 1e-10
    assert d["roofline"]["frac"] is None or d["roofline"]["frac"] <= 1.0
    alt = d.get("other_exchange_schedules", {})
    if "--exchange" in extra:
        assert d["config"]["exchange"] == "goal" and 0.0 <= d["goal"]["mean_unitary_infidelity_vs_identity"] <= 1.0
        assert "no gather" in d["config"]["parallelism"]
    elif "--gather-every" in extra:
        assert "per 32 steps" in d["config"]["parallelism"] and "all_gather_every_step" in alt
    else:
        assert "one RCCL all-gather of U per step" in d["config"]["parallelism"]
        assert ("issued asynchronously" in d["config"]["parallelism"]) == ("--overlap-gather" in extra)
        if "--overlap-gather" not in extra:  # (round 4) the same schedule with the collective under the next step, timed beside it
            assert alt["all_gather_every_step_overlapped"]["value"] > 0
        assert "all_gather_every_32_steps" in alt and alt["all_gather_every_32_steps"]["value"] > 0
        if "--scaling" not in extra:
            assert alt["goal_all_reduce_every_step"]["value"] > 0


# --------------------------------------------------------------------------
# full-size parity on 32 samples of every BASELINE configuration, real and complex Hamiltonians
# --------------------------------------------------------------------------


def _complexify(hks):
    hk = np.array(hks)
    k = min(1, hk.shape[0] - 1)
    up = np.triu(hk[k].real, 1)
    hk[k] = hk[k] + 0.3j * (up - up.T)
    return hk


@pytest.mark.parametrize("cfg,complex_ops", [(4, False), (5, False), (2, True), (3, True), (4, True), (5, True)])
def test_full_size_parity_32_samples(prop, cfg, complex_ops):
    """BASELINE.json configs at their per-GPU batch and full slice count: 32 samples spread over the batch against the
    oracle (one host process per sample); complex_ops gives a control operator an imaginary part, i.e. the general
    instances instead of the real-Hamiltonian fast path (the reference makes no such distinction, propagation.py:426-440)."""
    import torch

    from tests.oracle_pool import propagate_samples

    per_gpu = {2: 256, 3: 512, 4: 512, 5: 1024}[cfg]
    wl = workloads.make_workload(cfg, B=per_gpu)
    hks = _complexify(wl.hks) if complex_ops else wl.hks
    dev = "cuda:0"
    fr = wl.fr_phase
    if wl.lindblad:
        fr = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
    t = lambda x: torch.as_tensor(x, device=dev)
    r = prop.propagate_batch(t(wl.h0), t(hks), t(wl.signals), wl.dt, col_ops=t(wl.col_ops) if wl.lindblad else None,
                             lindbladian=wl.lindblad, fr_phase=t(fr))
    U = r["U"]
    idx = np.unique(np.linspace(0, per_gpu - 1, 32).astype(int))
    ref = propagate_samples(wl.h0, hks, wl.signals[idx], wl.dt, col_ops=wl.col_ops, lindbladian=wl.lindblad, fr_phase=wl.fr_phase[idx])
    got = U[torch.as_tensor(idx, device=dev)].cpu().numpy()
    assert fro_max(got, ref) < TOL
    if not wl.lindblad:
        Uh = U.cpu().numpy()
        assert np.abs(Uh @ Uh.conj().transpose(0, 2, 1) - np.eye(wl.D)).max() < 1e-10  # every sample of the batch


# --------------------------------------------------------------------------
# library hygiene: pre-sized workspace, graph capture of one call, per-device profiling switch
# --------------------------------------------------------------------------


def test_reserve_then_steady_state_and_graph_capture(prop):
    """c3p_reserve sizes the workspace for a call shape; afterwards calls do not (re)allocate, and one c3p_pwc_unitary
    call is capturable into a hipGraph (through torch.cuda.graph) and replays to the eager result."""
    import torch

    from c3_amd import _lib

    lib = _lib.load()
    lib.c3p_shutdown()  # start from an empty workspace
    wl = workloads.make_workload(2, B=256, N=200)
    dev = torch.device("cuda:0")
    _lib.check(lib.c3p_reserve(0, wl.B, wl.K, wl.N, wl.D, 0, 0))
    gen0 = lib.c3p_workspace_generation()
    assert gen0 > 0
    bp = prop.BatchPropagator(*(torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals)), wl.dt,
                              fr_phase=torch.as_tensor(wl.fr_phase, device=dev))
    eager = bp.run().clone()
    torch.cuda.synchronize()
    assert lib.c3p_workspace_generation() == gen0  # no allocation after the reservation
    out = torch.zeros_like(eager)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        bp.run(out=out)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            bp.run(out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    assert lib.c3p_workspace_generation() == gen0
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals[:2], wl.dt, fr_phase=wl.fr_phase[:2])
    assert fro_max(out[:2].cpu().numpy(), ref) < TOL
    # a capture that would have to grow the workspace is refused with an error, not a crash
    big = workloads.make_workload(3, B=8, N=40)
    bp2 = prop.BatchPropagator(*(torch.as_tensor(x, device=dev) for x in (big.h0, big.hks, big.signals)), big.dt)
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        failed = False
        try:
            with torch.cuda.graph(g2, stream=s):
                bp2.run()
        except Exception as e:
            failed = "stream capture" in str(e)
    torch.cuda.synchronize()
    assert failed
    bp2.run()  # and the library is still usable afterwards
    torch.cuda.synchronize()


def test_reserve_lindblad_and_profiling_switch(prop):
    import torch

    from c3_amd import _lib

    lib = _lib.load()
    wl = workloads.make_workload(4, B=3, N=12)
    _lib.check(lib.c3p_reserve(1, wl.B, wl.K, wl.N, wl.D, int(wl.col_ops.shape[0]), 0))
    gen = lib.c3p_workspace_generation()
    dev = "cuda:0"
    t = lambda x: torch.as_tensor(x, device=dev)
    lib.c3p_set_profiling(1)
    try:
        r = prop.propagate_batch(t(wl.h0), t(wl.hks), t(wl.signals), wl.dt, col_ops=t(wl.col_ops), lindbladian=True)
        torch.cuda.synchronize()
        assert lib.c3p_last_kernel_ms() > 0.0
    finally:
        lib.c3p_set_profiling(0)
    assert lib.c3p_workspace_generation() == gen
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True)
    assert fro_max(r["U"].cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("D,K", [(17, 1), (20, 2), (24, 3), (27, 3), (28, 2), (32, 4), (33, 1), (36, 3), (41, 2), (48, 2)])
@pytest.mark.parametrize("real", [False, True])
def test_ode_rowq_mid_dimensions(prop, D, K, real):
    """Schroedinger steps at 17 <= D <= 48 (c3p_ode_rowq.hip: several DPP rows per sample, H advanced along the linear
    pieces of the control amplitudes): every column-group class, all four solvers, trajectory and final state."""
    from c3_amd import _lib

    B, N = 5, 31
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 31 * D + K)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    psi /= np.linalg.norm(psi, axis=1, keepdims=True)
    for solver in ("rk4", "rk38", "rk5", "tsit5"):
        out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
        assert _lib.last_kernel() == "ode_row"
        for b in range(B):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger")["states"]
            assert np.abs(out[b] - ref).max() < 1e-11
    fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, "rk4", "schrodinger", final_only=True))
    ref = np.stack([o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], "rk4", "schrodinger", final_only=True)["states"] for b in range(B)])
    assert np.abs(fin - ref).max() < 1e-11


def test_ode_rowq_cfg3_long_and_rk4_unitary(prop):
    """cfg3's operators (D = 27, K = 3) over 300 steps (re-anchoring of H every 14 steps, chunk boundaries), and the
    rk4_unitary columns (stride-2 sample windows: two linear pieces per step) at D = 20."""
    import ctypes

    from c3_amd import _lib

    wl = workloads.make_workload(3, B=6, N=300)
    psi = np.zeros((wl.D, 1), complex)
    psi[2, 0] = 1.0
    out = np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, psi, "rk4", "schrodinger"))
    assert _lib.last_kernel() == "ode_row"
    for b in (0, 5):
        ref = o.ode_solver_arrays(wl.h0, wl.hks, wl.signals[b], wl.ts, psi, "rk4", "schrodinger")["states"]
        assert np.abs(out[b] - ref).max() < 1e-11
    D, K, B, Ns = 20, 2, 2, 33
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, False, 77)
    lib = _lib.load()
    U = np.zeros((B, D, D), complex)
    dUs = np.zeros((B, (Ns - 1) // 2, D, D), complex)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    for env, kern in ((None, "ode_mfma"), ("C3P_ODE_PROP_ROWS", "ode_row")):  # matrix-core kernel, lane-row column kernel
        if env:
            _lib.set_option(env[4:].lower(), "1")
        try:
            U[:] = 0
            dUs[:] = 0
            rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.1, B, K, Ns, D, 1, p(U), p(dUs), None)
        finally:
            if env:
                _lib.set_option(env[4:].lower(), None)
        assert rc == 0 and _lib.last_kernel() == kern
        for b in range(B):
            Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
            ref = o.rk4_unitary_arrays(Hs, 0.1, D)
            assert np.abs(U[b] - ref["U"]).max() < 1e-12
            assert np.abs(dUs[b] - ref["dUs"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("D,K,real", [(17, 1, True), (27, 3, True), (32, 2, False), (36, 3, True), (48, 4, False)])
def test_rk4_unitary_matrix_core(prop, D, K, real):
    """rk4_unitary (propagation.py:71-101,221-255) at 17 <= D <= 48 on the matrix-core kernel (one product per stage, stride-2
    sample windows, per-step propagators stored transposed from a state that is reset every step) against the oracle."""
    import ctypes

    from c3_amd import _lib

    B, Ns = 3, 21
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, real, 300 + D)
    lib = _lib.load()
    U = np.zeros((B, D, D), complex)
    dUs = np.zeros((B, (Ns - 1) // 2, D, D), complex)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.05, B, K, Ns, D, 1, p(U), p(dUs), None)
    assert rc == 0 and _lib.last_kernel() == "ode_mfma"
    for b in range(B):
        Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
        ref = o.rk4_unitary_arrays(Hs, 0.05, D)
        assert np.abs(U[b] - ref["U"]).max() < 1e-12
        assert np.abs(dUs[b] - ref["dUs"]).max() < 1e-12


def test_fused_infidelity_sum(prop):
    """c3p_gate_infid: per-sample unitary_infid / average_infid and their batch sum in one launch, against the oracle
    (fidelities.py:154-184,290-313) and the unfused device path."""
    import torch

    from c3_amd import fidelities as fid

    wl = workloads.make_workload(2, B=37, N=20)
    U = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, fr_phase=wl.fr_phase)
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
    ideal, _ = np.linalg.qr(a)
    Ud = torch.as_tensor(U, device="cuda:0")
    for kind, ref_fn, dev_fn in (("unitary", o.unitary_infid, fid.unitary_infid), ("average", o.average_infid, fid.average_infid)):
        r = fid.infid_sum(torch.as_tensor(ideal, device="cuda:0"), Ud, [0, 1], [3, 3], kind=kind, want_each=True)
        each = r["each"].cpu().numpy()
        ref = np.array([ref_fn(ideal, U[b], index=[0, 1], dims=[3, 3]) for b in range(wl.B)]).real
        assert np.abs(each - ref).max() < 1e-13
        s = r["sum"].cpu().numpy()
        assert abs(s[0] - ref.sum()) < 1e-11 and s[1] == wl.B
        unfused = np.asarray(dev_fn(ideal, U, index=[0, 1], dims=[3, 3])).real
        assert np.abs(each - unfused).max() < 1e-13
    # host-pointer route
    r = fid.infid_sum(ideal, U, [0, 1], [3, 3], kind="unitary", want_each=True)
    assert np.abs(r["each"] - np.array([o.unitary_infid(ideal, U[b], index=[0, 1], dims=[3, 3]) for b in range(wl.B)]).real).max() < 1e-13


# --------------------------------------------------------------------------
# edge cases of the lane-row ODE kernels (the reference's loops accept them: propagation.py:687-752)
# --------------------------------------------------------------------------


def test_ode_row_edge_cases(prop):
    """One sample, two time samples (the minimum the interpolation needs), a one-level system, a sample count that is not
    a multiple of the samples per wavefront, an empty batch."""
    from c3_amd import _lib

    # N = 2, B = 1, D = 1 .. 3
    for D in (1, 2, 3):
        h0, hks, sig, ts = _ode_problem(D, 1, 1, 2, False, D)
        psi = np.ones((D, 1), complex) / np.sqrt(D)
        for solver in ("rk4", "rk5"):
            out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger"))
            ref = o.ode_solver_arrays(h0, hks, sig[0], ts, psi, solver, "schrodinger")["states"]
            assert out.shape == (1, 2, D, 1) and np.abs(out[0] - ref).max() < 1e-13
        rho = psi @ psi.conj().T
        out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, "rk4", "von_neumann"))
        ref = o.ode_solver_arrays(h0, hks, sig[0], ts, rho, "rk4", "von_neumann")["states"]
        assert np.abs(out[0] - ref).max() < 1e-13
    assert _lib.last_kernel() == "ode_row"
    # empty batch
    h0, hks, sig, ts = _ode_problem(4, 2, 3, 10, False, 9)
    out = prop.ode_solve_batch(h0, hks, sig[:0], ts[1] - ts[0], np.ones((4, 1), complex) / 2)
    assert tuple(np.asarray(out).shape) == (0, 10, 4, 1)
    # more control lines than the lane-row kernels hold in registers (since round 4: H assembled per sample index first, the
    # lane-row kernels interpolate it; tests/test_gpu_round4.py), same results
    h0, hks, sig, ts = _ode_problem(5, 6, 3, 12, False, 21)
    psi = np.zeros((5, 1), complex)
    psi[1, 0] = 1.0
    out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, "rk4", "schrodinger"))
    assert _lib.last_kernel() == "ode_row"
    for b in range(3):
        assert np.abs(out[b] - o.ode_solver_arrays(h0, hks, sig[b], ts, psi, "rk4", "schrodinger")["states"]).max() < 1e-12


# --------------------------------------------------------------------------
# register-resident kernel, zero-padded classes (VERDICT r2 missing #4): Dm = 41..48 -> 49, 56..64 -> 65, 70..80 -> 81
# --------------------------------------------------------------------------


@pytest.mark.parametrize("D", [41, 45, 48, 56, 60, 64, 70, 77, 80])
def test_regd_padded_classes_unitary(prop, D):
    """Matrix dimensions below a kernel class run zero padded (the padding stays decoupled through every product);
    complex Hermitian operators, partial propagators, several time segments, frame rotation."""
    from c3_amd import _lib

    rng = np.random.default_rng(D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    B, N, K = 3, 19, 2
    h0, hks = herm(0.08), np.stack([herm(0.05) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    ph = rng.uniform(0, 2 * np.pi, size=(B, D))
    r = prop.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph, want_dUs=True)
    assert _lib.last_kernel() == "mfma"
    for b in range(B):
        ref = o.pwc_arrays(h0, hks, sig[b], 1.0)
        assert np.linalg.norm(np.asarray(r["U"][b]) - np.exp(1j * ph[b])[:, None] * ref["U"]) < TOL
        assert np.abs(np.asarray(r["dUs"][b]) - ref["dUs"]).max() < 1e-12
    _lib.set_option("regd_pad", "0")  # the arena kernel these dimensions ran on before
    try:
        old = prop.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph)
    finally:
        _lib.set_option("regd_pad", None)
    assert fro_max(r["U"], old["U"]) < 1e-11


def test_regd_padded_lindblad_64(prop):
    """D = 8 -> 64 x 64 Lindblad superoperators in the 65 class (propagation.py:551-585)."""
    rng = np.random.default_rng(64)
    D, B, N, K = 8, 2, 14, 2
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0, hks = herm(0.3), np.stack([herm(0.2) for _ in range(K)])
    col = np.stack([0.1 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(2)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    r = prop.propagate_batch(h0, hks, sig, 1.0, col_ops=col, lindbladian=True, want_dUs=True)
    ref = o.propagate_batch(h0, hks, sig, 1.0, col_ops=col, lindbladian=True)
    assert fro_max(r["U"], ref) < TOL
    assert np.abs(np.asarray(r["dUs"][0]) - o.tf_propagation_lind(h0, hks, col, sig[0], 1.0)).max() < 1e-12


@pytest.mark.parametrize("D,K,real", [(3, 1, True), (9, 2, True), (9, 2, False), (12, 3, False)])
def test_ode_row_time_segments_small_batches(prop, D, K, real):
    """Final state of a SMALL batch: the interval is cut into time segments whose step maps (D columns each) are integrated
    in parallel and multiplied in order (the equations are linear) -- same numbers as the direct integration to rounding,
    and as the oracle's solver; uneven last segment, all four solvers."""
    B, N = 5, 203
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 900 + D)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "rk38", "rk5", "tsit5"):
        seg = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
        _lib.set_option("ode_no_seg", "1")
        try:
            direct = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
        finally:
            _lib.set_option("ode_no_seg", None)
        assert np.abs(seg - direct).max() < 1e-12 * max(1.0, np.abs(direct).max())
        for b in (0, B - 1):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger", final_only=True)["states"]
            assert np.abs(seg[b] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_rk4_unitary_time_segments(prop):
    import ctypes

    from c3_amd import _lib

    D, K, B, Ns = 6, 2, 2, 401
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, False, 55)
    lib = _lib.load()
    U = np.zeros((B, D, D), complex)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.02, B, K, Ns, D, 1, p(U), None, None)
    assert rc == 0 and _lib.last_kernel() == "ode_row"
    for b in range(B):
        Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
        assert np.abs(U[b] - o.gen_u_rk4(Hs, 0.02, D)).max() < 1e-12


# --------------------------------------------------------------------------
# matrix-core rho-valued ODE kernel (c3p_ode_rhoq.hip): von Neumann / Lindblad states at 17 <= D <= 48
# --------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["rk4", "rk38", "rk5", "tsit5"])
@pytest.mark.parametrize("D,K,C,real", [(17, 1, 0, True), (27, 3, 0, True), (27, 3, 0, False), (32, 4, 0, False), (33, 2, 0, True),
                                        (36, 3, 0, False), (48, 1, 0, True), (17, 2, 1, False), (27, 3, 2, True), (32, 1, 3, False)])
def test_ode_rhoq_density_matrices(prop, solver, D, K, C, real):
    """von_neumann (propagation.py:902-904) and lindblad (:886-894) on 16 x 16 register tiles + fp64 MFMA products:
    trajectory and final state against the oracle's solver, real and complex operators, every tile-count class, dimensions
    that are not multiples of 4 or 16."""
    from c3_amd import _lib

    B, N = 3, 11
    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 11 * D + C)
    rng = np.random.default_rng(D + C)
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    rho = a @ a.conj().T
    rho /= np.trace(rho)
    col = np.stack([0.2 * (rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))) for _ in range(C)]) if C else None
    step = "lindblad" if C else "von_neumann"
    out = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col))
    assert _lib.last_kernel() == "ode_mfma"
    fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, solver, step, col_ops=col, final_only=True))
    for b in range(B):
        ref = o.ode_solver_arrays(h0, hks, sig[b], ts, rho, solver, step, col=col)["states"]
        assert np.abs(out[b] - ref).max() < 1e-11
        assert np.abs(fin[b] - ref[-1]).max() < 1e-11


@pytest.mark.gpu
def test_ode_rhoq_matches_workgroup_kernel_per_sample_states(prop):
    """Per-sample initial states and the round-1 kernel as a second opinion (C3P_ODE_WG=1) at cfg3's dimension."""
    from c3_amd import _lib

    D, K, B, N = 27, 3, 6, 40
    h0, hks, sig, ts = _ode_problem(D, K, B, N, True, 5)
    rng = np.random.default_rng(1)
    a = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    rho = a @ a.conj().transpose(0, 2, 1)
    rho /= np.trace(rho, axis1=1, axis2=2)[:, None, None]
    new = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, "tsit5", "von_neumann", final_only=True))
    assert _lib.last_kernel() == "ode_mfma"
    _lib.set_option("ode_wg", "1")
    try:
        old = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], rho, "tsit5", "von_neumann", final_only=True))
        assert _lib.last_kernel() == "ode_wg"
    finally:
        _lib.set_option("ode_wg", None)
    assert np.abs(new - old).max() < 1e-12
    assert np.abs(np.trace(new, axis1=1, axis2=2) - 1.0).max() < 1e-12


@pytest.mark.gpu
def test_ode_rhoq_general_inputs_take_the_two_product_form(prop):
    """A non-Hermitian 'state' and a non-Hermitian control operator are legal inputs of ode_solver (propagation.py:687-752
    makes no assumption): the per-sample symmetry check must send them to the general instance, and a batch may mix both."""
    D, K, B, N = 20, 2, 4, 9
    h0, hks, sig, ts = _ode_problem(D, K, B, N, False, 41)
    rng = np.random.default_rng(3)
    a = rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D))
    rho = a @ a.conj().transpose(0, 2, 1)
    rho[1] = a[1]  # not Hermitian
    rho[3] = a[3] + 0.5 * a[3].T
    for hk_general in (False, True):
        hk = hks.copy()
        if hk_general:
            hk[1] = hk[1] + 0.3 * rng.normal(size=(D, D))  # not Hermitian either
        out = np.asarray(prop.ode_solve_batch(h0, hk, sig, ts[1] - ts[0], rho, "rk38", "von_neumann", final_only=True))
        for b in range(B):
            ref = o.ode_solver_arrays(h0, hk, sig[b], ts, rho[b], "rk38", "von_neumann", final_only=True)["states"]
            assert np.abs(out[b] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,step", [(3, "von_neumann"), (3, "lindblad"), (5, "von_neumann")])
def test_ode_rhoq_full_size_properties(prop, cfg, step):
    """BASELINE cfg3 / cfg5 operators and slice counts (N = 2000 / 5000) on the matrix-core kernel: the RK step preserves the
    trace and Hermiticity exactly (rounding), the one-product and the two-product forms agree, two samples against the
    oracle's solver."""
    wl = workloads.make_workload(cfg, B=16)
    D = wl.D
    psi = np.zeros((D, 1), complex)
    psi[1, 0] = 0.6
    psi[2, 0] = 0.8j
    rho = psi @ psi.conj().T
    col = None
    if step == "lindblad":
        col = np.stack([0.05 * np.diag(np.sqrt(np.arange(1, D) % 3 + 1.0), 1), 0.03 * np.diag(np.arange(D) % 3).astype(float)]).astype(complex)
    run = lambda: np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, rho, "rk4", step, col_ops=col, final_only=True))
    out = run()
    from c3_amd import _lib

    assert _lib.last_kernel() == "ode_mfma"
    assert np.abs(np.trace(out, axis1=1, axis2=2) - 1.0).max() < 1e-11
    assert np.abs(out - out.conj().transpose(0, 2, 1)).max() < 1e-13
    _lib.set_option("ode_rho_general", "1")
    try:
        gen = run()
    finally:
        _lib.set_option("ode_rho_general", None)
    assert np.abs(out - gen).max() < 1e-11
    for b in (0, 15):
        ref = o.ode_solver_arrays(wl.h0, wl.hks, wl.signals[b], wl.ts, rho, "rk4", step, col=col, final_only=True)["states"]
        assert np.abs(out[b] - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,step,solver,n", [(2, "schrodinger", "rk4", 32), (2, "schrodinger", "tsit5", 32), (2, "von_neumann", "rk4", 32),
                                               (2, "lindblad", "rk5", 16), (3, "schrodinger", "rk4", 16), (3, "von_neumann", "rk4", 8),
                                               (3, "lindblad", "rk4", 8), (5, "von_neumann", "rk38", 4)])
def test_ode_full_size_parity_many_samples(prop, cfg, step, solver, n):
    """The ODE solver variant at BASELINE's operators and slice counts (N = 1000 / 2000 / 5000), every kernel family (lane-row
    vector / matrix, mid-D lane-row, matrix core), n samples each against the oracle's solver of the same tableau (one oracle
    process per sample on the host cores): relative 1e-11 of the largest state element."""
    from tests.oracle_pool import ode_final_states

    wl = workloads.make_workload(cfg, B=n)
    D = wl.D
    psi = np.zeros((D, 1), complex)
    psi[0, 0] = 0.6
    psi[1, 0] = 0.8j
    init = psi if step == "schrodinger" else psi @ psi.conj().T
    col = None
    if step == "lindblad":
        col = np.stack([0.05 * np.diag(np.sqrt(np.arange(1, D) % 3 + 1.0), 1), 0.03 * np.diag(np.arange(D) % 3).astype(float)]).astype(complex)
    out = np.asarray(prop.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, init, solver, step, col_ops=col, final_only=True))
    ref = ode_final_states(wl.h0, wl.hks, wl.signals, wl.ts, init, solver, step, col_ops=col)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() < 1e-11 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("D,K,B,N,real", [(20, 2, 5, 200, False), (27, 3, 2, 160, True), (36, 1, 3, 130, False), (40, 2, 1, 100, True)])
def test_ode_mid_dimension_time_segments(prop, D, K, B, N, real):
    """Few states at 17 <= D <= 40, final state only: the time axis is cut into segments whose step maps are integrated in
    parallel on the matrix-core kernel (one workgroup per sample and segment), multiplied in order on the mid-D chain kernel and
    applied to the initial state.  Same RK steps, re-associated products: against the oracle's sequential integration and
    against the direct integration (C3P_ODE_NO_SEG=1), per-sample initial states."""
    from c3_amd import _lib

    h0, hks, sig, ts = _ode_problem(D, K, B, N, real, 900 + D)
    rng = np.random.default_rng(D)
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    for solver in ("rk4", "rk5"):
        fin = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
        assert _lib.last_kernel() == "ode_mfma"
        _lib.set_option("ode_no_seg", "1")
        try:
            direct = np.asarray(prop.ode_solve_batch(h0, hks, sig, ts[1] - ts[0], psi, solver, "schrodinger", final_only=True))
            assert _lib.last_kernel() == "ode_row"
        finally:
            _lib.set_option("ode_no_seg", None)
        assert np.abs(fin - direct).max() < 1e-12
        for b in range(B):
            ref = o.ode_solver_arrays(h0, hks, sig[b], ts, psi[b], solver, "schrodinger", final_only=True)["states"]
            assert np.abs(fin[b] - ref).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("D,B,Ns", [(27, 3, 257), (40, 1, 161)])
def test_rk4_unitary_mid_dimension_time_segments(prop, D, B, Ns):
    """rk4_unitary for a handful of gates at 17 <= D <= 40: one workgroup per (gate, time segment), ordered product of the
    segment maps; against the oracle and the unsegmented launch."""
    import ctypes

    from c3_amd import _lib

    K = 2
    h0, hks, sig, _ = _ode_problem(D, K, B, Ns, False, 70 + D)
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h0c, hkc, sgc = np.ascontiguousarray(h0), np.ascontiguousarray(hks), np.ascontiguousarray(sig)
    res = {}
    for env in (None, "C3P_ODE_NO_SEG"):
        if env:
            _lib.set_option(env[4:].lower(), "1")
        try:
            U = np.zeros((B, D, D), complex)
            dUs = np.zeros((B, (Ns - 1) // 2, D, D), complex)
            rc = lib.c3p_rk4_unitary(p(h0c), p(hkc), p(sgc), None, 0, 0.02, B, K, Ns, D, 1, p(U), p(dUs), None)
            assert rc == 0 and _lib.last_kernel() == "ode_mfma"
            res[env] = (U, dUs)
        finally:
            if env:
                _lib.set_option(env[4:].lower(), None)
    assert np.abs(res[None][0] - res["C3P_ODE_NO_SEG"][0]).max() < 1e-12
    assert np.abs(res[None][1] - res["C3P_ODE_NO_SEG"][1]).max() == 0.0
    for b in range(B):
        Hs = h0[None] + np.einsum("kn,kij->nij", sig[b], hks)
        ref = o.rk4_unitary_arrays(Hs, 0.02, D)
        assert np.abs(res[None][0][b] - ref["U"]).max() < 1e-12
        assert np.abs(res[None][1][b] - ref["dUs"]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,B", [(2, 50, 3), (3, 40, 2), (4, 21, 2), (5, 12, 1), (6, 9, 2)])
def test_lindblad_with_per_slice_hamiltonians_on_matrix_cores(prop, D, N, B):
    """Branch B together with model.lindbladian (propagation.py:295-308 into :551-585): the superoperator generator of every
    slice Hamiltonian is formed densely and propagated by the supplied-generator mode of the matrix-core chain kernels (before:
    the generic LDS kernel); U and the per-slice propagators against the oracle, per-sample Hamiltonians."""
    from c3_amd import _lib

    rng = np.random.default_rng(31 * D + N)
    H = np.stack([np.stack([_rand_herm(rng, D, 0.4) for _ in range(N)]) for _ in range(B)])
    col = np.stack([0.15 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(2)])
    r = prop.propagate_batch(H, None, None, 0.7, col_ops=col, lindbladian=True, want_dUs=True)
    assert _lib.last_kernel() == ("smalld" if D * D <= 12 else "mfma")
    U, dUs = np.asarray(r["U"]), np.asarray(r["dUs"])
    for b in range(B):
        ref = o.pwc_arrays(H[b], None, None, 0.7, col_ops=col, lindbladian=True)
        assert np.linalg.norm(U[b] - ref["U"]) < TOL
        assert np.abs(dUs[b] - ref["dUs"]).max() < 1e-12
