"""C-ABI surface and host logic, no GPU compute calls."""
import os
import re

import numpy as np
import pytest

from c3_amd import _lib, propagation, workloads, dist as c3dist
from oracle import c3_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "c3prop.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c3p_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 12
    for name in syms:
        assert hasattr(lib, name), f"libc3prop.so lacks {name}"
        assert name in _lib.SIGNATURES, f"ctypes binding lacks {name}"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.c3p_version() >= 1


def test_dispatch_table_in_integration_md_is_the_committed_measurement():
    """INTEGRATION.md section 8 is the rendering of profiles/r06/dispatch_table.json (launch logs measured on the GPU box,
    VERDICT r5 item 9); tests/test_gpu_round6.py re-measures sample rows."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dispatch_table.py"), "--check", os.path.join(ROOT, "profiles", "r06", "dispatch_table.json")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_launch_log_is_empty_before_any_compute_call(lib):
    buf = __import__("ctypes").create_string_buffer(16)
    assert lib.c3p_last_kernel_detail(buf, 16) >= 0  # callable without a device; no compute call on this thread yet or a short log


def test_no_gpu_fails_loudly(lib):
    """The product path has no CPU fallback: without a device every entry point raises."""
    if lib.c3p_device_count() > 0:
        pytest.skip("a GPU is visible")
    wl = workloads.make_workload(1, B=1, N=4)
    with pytest.raises(_lib.C3PropError, match="C3:Error"):
        propagation.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt)
    with pytest.raises(_lib.C3PropError):
        propagation.tf_matmul_left(np.zeros((2, 3, 3), complex))
    with pytest.raises(_lib.C3PropError):
        propagation.ode_solve_batch(wl.h0, wl.hks, wl.signals, wl.dt, np.zeros((3, 1), complex))


def test_product_code_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "c3_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


def test_registries_mirror_reference():
    """propagation.py:18-21,39-68: provider names the reference registers on this path."""
    assert set(propagation.unitary_provider) >= {"pwc", "tf_propagation", "rk4_unitary", "gen_dus_rk4"}
    assert set(propagation.state_provider) >= {"ode_solver", "ode_solver_final_state"}
    assert set(propagation.solver_dict) == {"rk4", "rk38", "rk5", "tsit5"}
    assert set(propagation.step_dict) == {"lindblad", "schrodinger", "von_neumann"}
    assert propagation.solver_slicing == o.solver_slicing


def test_workload_is_deterministic_and_shardable():
    a = workloads.make_workload(2, B=6, N=16)
    b = workloads.make_workload(2, B=6, N=16)
    assert np.array_equal(a.signals, b.signals)
    lo = workloads.make_workload(2, B=3, N=16, b_offset=0)
    hi = workloads.make_workload(2, B=3, N=16, b_offset=3)
    assert np.array_equal(np.concatenate([lo.signals, hi.signals]), a.signals)
    assert np.abs(a.h0 - a.h0.conj().T).max() < 1e-12 * np.abs(a.h0).max()
    assert a.hks.shape == (2, 9, 9) and a.fr_phase.shape == (6, 9)
    assert abs(a.ts[1] - a.ts[0] - a.dt) < 1e-25


def test_time_grid_check_raises_like_reference():
    """propagation.py:301-308: non-uniform time grids raise the reference's message."""
    ts = np.linspace(0, 1, 10) ** 2
    with pytest.raises(Exception, match="Something with the times happend"):
        propagation._uniform_ts([ts, ts])
    good = np.linspace(0.5e-11, 9.5e-11, 10)
    assert np.allclose(propagation._uniform_ts([good, good]), good)


class _Instr:
    def __init__(self, name):
        self.name = name

    def get_key(self):
        return self.name


def test_gather_pwc_inputs_branches():
    m = workloads.ChipModel((3, 3), (5e9, 5.6e9), (-210e6, -240e6), {(0, 1): 20e6}, {"d1": 0, "d2": 1}, t1=(27e-6, 23e-6), t2star=(39e-6, 31e-6))
    ts = workloads.centred_time_grid(0.0, 2e-10, 100e9)
    sig = {"g": {"d1": {"values": np.sin(ts * 1e10), "ts": ts}, "d2": {"values": np.cos(ts * 1e10), "ts": ts}}}
    gen = workloads.SignalSource(sig)
    h0, hks, signals, ts_out, dt, col = propagation.gather_pwc_inputs(m, gen, _Instr("g"))
    assert h0.shape == (9, 9) and hks.shape == (2, 9, 9) and signals.shape == (2, 20) and col is None
    assert abs(dt - 1e-11) < 1e-24
    m.controllability = False
    h0b, hksb, sigb, ts_b, dt_b, _ = propagation.gather_pwc_inputs(m, gen, _Instr("g"))
    assert h0b.shape == (20, 9, 9) and hksb is None and sigb is None and ts_b.shape == (19,)
    want = o.sum_h0_hks(h0, hks, signals)
    assert np.abs(h0b - want).max() < 1e-6 * np.abs(want).max()
    m.controllability = True
    m.set_lindbladian(True)
    m.set_max_excitations(2)
    h0c, hksc, _, _, _, colc = propagation.gather_pwc_inputs(m, gen, _Instr("g"))
    assert h0c.shape == (6, 6) and hksc.shape == (2, 6, 6) and len(colc) == 2 and colc[0].shape == (6, 6)


def test_experiment_folding_stack_and_slot():
    """experiment.py:76-107 on the host mirror (no GPU needed)."""
    from c3_amd.experiment import Experiment, _tf_matmul_n_even, _tf_matmul_n_odd

    class PM:
        model = None
        generator = None
        instructions = {"a": workloads.Gate("a", 0.0, 7e-9), "b": workloads.Gate("b", 0.0, 1e-10)}

    exp = Experiment(PM(), sim_res=100e9)
    assert exp.propagation is propagation.unitary_provider["pwc"]
    assert set(exp.folding_stack) == {700, 10}
    kinds = ["even" if f is _tf_matmul_n_even else "odd" for f in exp.folding_stack[700]]
    assert kinds == o.compute_folding_stack(700)
    exp.set_prop_method("ode_solver")
    assert exp.propagation is propagation.state_provider["ode_solver"]
    f = lambda *a, **k: None
    exp.set_prop_method(f)
    assert exp.propagation is f
    exp.set_opt_gates("a")
    assert exp.opt_gates == ["a"]
    exp.set_opt_gates_seq([["a", "b"], ["b"]])
    assert sorted(exp.opt_gates) == ["a", "b"]
    # the stack functions multiply like tf_matmul_n's levels
    rng = np.random.default_rng(0)
    M = rng.normal(size=(5, 2, 2))
    cur = M
    for fn in Experiment(PM(), sim_res=5 / 1e-10 ).folding_stack.get(5, []):
        cur = fn(cur[1::2], cur[0::2])
    # 5 steps at that resolution: ((M4)(M3 M2))(M1 M0) ordering
    if cur.shape[0] == 1:
        assert np.allclose(cur[0], M[4] @ M[3] @ M[2] @ M[1] @ M[0])


def test_shard_bounds_cover_batch():
    for B in (1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            spans = [c3dist.shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == c3dist.max_shard(B, world)


def test_algorithmic_flops_formula():
    # SURVEY.md 8d worked value: cfg2, Pade-9, s=0 -> 43 416 flop/slice
    assert abs(o.algorithmic_flops_per_slice(9, 2, 9, 0) - 43416.0) < 1e-6
    assert abs(o.algorithmic_flops_per_slice(3, 1, 7, 0) - 1404.0) < 1e-6


def test_taylor_thresholds_are_safe():
    """The device kernels replace Pade+solve by a scaled Taylor polynomial (c3p_common.h):
    check on the CPU that degree m at its threshold reproduces expm to rounding level."""
    import math

    th = {4: 4.0e-4, 8: 5.45e-2, 12: 3.18e-1, 16: 8.16e-1, 20: 1.49}
    rng = np.random.default_rng(0)
    for m, theta in th.items():
        A = rng.normal(size=(9, 9)) + 1j * rng.normal(size=(9, 9))
        A = A - A.conj().T
        A *= theta / np.abs(A).sum(0).max()
        T = sum(np.linalg.matrix_power(A, k) / math.factorial(k) for k in range(m + 1))
        assert np.abs(T - o.expm(A)).max() < 1e-15 * max(1.0, 10 * theta)


def test_t18_coefficients_reproduce_taylor_series():
    """The Bader-Blanes-Casas T18 constants in c3p_common.h, expanded as a scalar polynomial,
    must equal sum_{k<=18} x^k/k! (the kernels rely on it for degree-18 accuracy)."""
    import math
    from numpy.polynomial import polynomial as Pn

    text = open(os.path.join(ROOT, "c3_amd", "csrc", "c3p_common.h")).read()
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"#define C3P_T18_([AB]\d\d) \((-?[0-9.]+)\)", text)}
    assert len(c) == 20

    def poly(c0, c1, c2, c3, c6):
        p = np.zeros(7)
        p[0], p[1], p[2], p[3], p[6] = c0, c1, c2, c3, c6
        return p

    B1 = poly(0, c["A11"], c["A21"], c["A31"], 0)
    B2 = poly(0, c["B11"], c["B21"], c["B31"], c["B61"])
    B3 = poly(c["B02"], c["B12"], c["B22"], c["B32"], c["B62"])
    B4 = poly(c["B03"], c["B13"], c["B23"], c["B33"], c["B63"])
    B5 = poly(0, 0, c["B24"], c["B34"], c["B64"])
    A9 = Pn.polyadd(Pn.polymul(B1, B5), B4)
    T = Pn.polyadd(B2, Pn.polymul(Pn.polyadd(B3, A9), A9))
    assert len(T) == 19
    for k in range(19):
        assert abs(T[k] * math.factorial(k) - 1.0) < 5e-15, (k, T[k])
    # and as a matrix function at its threshold
    rng = np.random.default_rng(2)
    A = rng.normal(size=(9, 9)) + 1j * rng.normal(size=(9, 9))
    A = A - A.conj().T
    A *= 1.13 / np.abs(A).sum(0).max()
    M = lambda p: sum(p[k] * np.linalg.matrix_power(A, k) for k in range(len(p)) if p[k] != 0)
    A9m = M(B1) @ M(B5) + M(B4)
    T18 = M(B2) + (M(B3) + A9m) @ A9m
    assert np.abs(T18 - o.expm(A)).max() < 2e-14


def test_computational_rows_match_projector():
    from c3_amd import fidelities

    for dims, index in (([3, 3], [0, 1]), ([3, 3], [1]), ([3, 4, 2], [0, 2]), ([2, 2], None)):
        P = o.projector(dims, index if index else list(range(len(dims))))
        want = [int(np.argmax(P[:, a])) for a in range(P.shape[1])]
        assert list(fidelities.computational_rows(dims, index)) == want
    assert set(fidelities.fidelities) >= {"unitary_infid", "unitary_infid_set", "average_infid", "average_infid_set"}


def test_dpp_blocks_are_hazard_guarded():
    """The ODE lane-row kernels feed `v_fmac_f64_dpp` from inline asm; the ISA the compiler produced must keep every DPP
    run behind its `s_nop` and free of VALU EXEC writes (tools/check_dpp_hazards.py; the mid-D source compiles in seconds,
    the D <= 16 source -- 140 kernels -- is checked by `python tools/check_dpp_hazards.py` and in the closing script)."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), "c3p_ode_rowq.hip"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_lindblad_tape_layout_is_host_arithmetic(lib):
    """c3p_pwc_lindblad_tape_bytes needs no device: the tape of the open-system evaluation (include/c3prop.h) is sized from the
    shape alone -- two table sets, their flags, B x S complex segment slots, B x N real D^2 x D^2 prefixes -- and shapes neither the
    Hermitian-basis kernels (D = 7, 8, 9) nor the small-D kernels (D = 2, 3) serve report 0."""
    import ctypes

    seg = ctypes.c_int(-1)
    n = lib.c3p_pwc_lindblad_tape_bytes(64, 2, 1000, 9, ctypes.byref(seg))
    assert seg.value == 4  # 64 samples x 4 segments = one round of 256 workgroups
    assert n >= 64 * 1000 * 81 * 81 * 8 + 64 * 4 * 81 * 81 * 16
    assert n < 1.02 * (64 * 1000 * 81 * 81 * 8 + 64 * 4 * 81 * 81 * 16) + (1 << 22)
    for D in (5, 6, 10):
        assert lib.c3p_pwc_lindblad_tape_bytes(4, 2, 100, D, ctypes.byref(seg)) == 0 and seg.value == 0
    # D = 4 (two qubits, real Hermitian-basis kernels): two real table sets, B x S real segment products, B x N real slice propagators
    n = lib.c3p_pwc_lindblad_tape_bytes(64, 2, 1000, 4, ctypes.byref(seg))
    assert seg.value >= 4 and seg.value % 4 == 0 and 64 * (1000 + seg.value) * 256 * 8 <= n < 64 * (1000 + seg.value) * 256 * 8 + (1 << 21)
    # D = 2, 3 (small-D kernels): tables + B x S segment products + B x N slice propagators, complex D^2 x D^2
    for D in (2, 3):
        n = lib.c3p_pwc_lindblad_tape_bytes(64, 1, 1000, D, ctypes.byref(seg))
        S, m = seg.value, D**4 * 16
        assert S >= 1 and S % 4 == 0 and 64 * S <= 4096
        assert 64 * (1000 + S) * m <= n < 64 * (1000 + S) * m + (1 << 20)
    assert lib.c3p_pwc_lindblad_tape_bytes(4, 9, 100, 3, None) == 0
    assert lib.c3p_pwc_lindblad_tape_bytes(4, 17, 100, 9, None) == 0  # more control lines than the kernels hold


def _build_abi_client(tmp_path):
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = os.path.join(str(tmp_path), "abi_client")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "checks", "abi_client.c"), "-o", exe, "-ldl", "-lm"], check=True)
    return exe


def test_header_is_plain_c_and_a_c_client_binds_the_library(lib, tmp_path):
    """include/c3prop.h compiles as strict C99 and a C program that only knows the header (dlopen + dlsym, what a foreign
    FFI does) drives the option table, the tape arithmetic and the no-device error path of the in-tree libc3prop.so."""
    import subprocess

    exe = _build_abi_client(tmp_path)
    out = subprocess.run([exe, _lib.LIB_PATH, "host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "ABI_CLIENT_HOST_OK" in out.stdout


def test_economised_cos_sin_tables_are_the_generator_output_and_accurate():
    """c3p_common.h's c3p_mm{6,7,8}_{cos,sinc} (round 6): (1) equal to what tools/gen_minimax_cossin.py derives in exact rational
    arithmetic, (2) accurate on their whole interval: |p(w) - cos(sqrt w)| and |p(w) - sin(sqrt w)/sqrt w| on [0, theta^2] in
    exact arithmetic against long Taylor sums, below 2.4e-16 / 4e-17 (economisation + rounding of the coefficients to double) -- for a real symmetric Y that scalar error IS the matrix error."""
    import importlib.util
    from fractions import Fraction as F
    from math import factorial

    spec = importlib.util.spec_from_file_location("gen_mm", os.path.join(ROOT, "tools", "gen_minimax_cossin.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    text = open(os.path.join(ROOT, "c3_amd", "csrc", "c3p_common.h")).read()
    for name, theta, deg in (("mm6", F(83, 100), 6), ("mm7", F(130, 100), 7), ("mm8", F(185, 100), 8)):
        assert f"#define C3P_{name.upper()}_THETA {float(theta)!r}" in text
        pc, ps, dc, ds = g.tables(theta, deg)
        for kind, tab, bound in (("cos", pc, 2.4e-16), ("sinc", ps, 4.0e-17)):
            m = re.search(r"c3p_%s_%s\[%d\] = \{([^}]*)\}" % (name, kind, deg + 1), text)
            vals = [float.fromhex(x.strip()) for x in m.group(1).split(",")]
            assert vals == [float(x) for x in tab], (name, kind)
            assert vals[0] == 1.0  # exp(0) = I exactly
            # exact evaluation of the DOUBLE coefficients against a degree-30 Taylor sum at 400 points of [0, theta^2]
            worst = F(0)
            L = theta * theta
            for i in range(401):
                w = L * i / 400
                p = sum(F(v) * w**j for j, v in enumerate(vals))
                f = sum(F((-1) ** j, factorial(2 * j + (0 if kind == "cos" else 1))) * w**j for j in range(31))
                worst = max(worst, abs(p - f))
            assert float(worst) < bound, (name, kind, float(worst))
        assert dc < 2.2e-16 and ds < 2e-17



def test_four_product_scheme_for_normal_generators():
    """c3p_common.h's c3p_e4n (round 6): (1) the values tools/gen_t16n4.py solves for (unit constant term forced), (2) the scheme with
    those DOUBLE parameters reproduces e^{iy} on [-1.35, 1.35] to 8e-16 (exact expansion of the 4-product scheme), (3) on matrices:
    skew-Hermitian generators, and real skew-symmetric ones with a dissipator-like symmetric part up to the accepted 0.25, at norms up to
    the radius are as close to scipy's expm as the published 5-product T18 (whose radius 1.13 they exceed: T18 takes one squaring there),
    (4) the plan rule: the four products are taken exactly when they save a product over T18N / Paterson-Stockmeyer."""
    import importlib.util
    import math
    from decimal import Decimal as Dc
    from fractions import Fraction as F

    import scipy.linalg as sl

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    g, t18 = load("gen_t16n4"), load("gen_t18_normal")
    text = open(os.path.join(ROOT, "c3_amd", "csrc", "c3p_common.h")).read()
    assert "#define C3P_E4N_THETA 1.35" in text
    e = [float.fromhex(x.strip()) for x in re.search(r"c3p_e4n\[16\] = \{([^}]*)\};", text).group(1).split(",")]
    u, resid, dc, ds, r0, r1 = g.solve(F(135, 100))
    assert float(resid) < 1e-40 and dc < 1e-16 and ds < 2e-17
    want = [float(x) for x in u]
    assert abs(want[15] - 1.0) < 2.3e-16
    want[15] = 1.0
    assert e == want
    coeff = g.scheme([Dc(x) for x in e])
    worst = 0.0
    for i in range(-100, 101):
        y = Dc("1.35") * i / 100
        re_, im_, yk = Dc(0), Dc(0), Dc(1)
        for k in range(17):
            term = coeff[k] * yk
            if k % 4 == 0: re_ += term
            elif k % 4 == 1: im_ += term
            elif k % 4 == 2: re_ -= term
            else: im_ -= term
            yk *= y
        worst = max(worst, abs(complex(float(re_) - math.cos(float(y)), float(im_) - math.sin(float(y)))))
    assert worst < 8e-16, worst

    def E4(A):
        I = np.eye(A.shape[0]); A2 = A @ A
        y0 = A2 @ (e[0] * A2 + e[1] * A)
        y1 = (y0 + e[2] * A2 + e[3] * A) @ (y0 + e[4] * A2) + e[5] * y0 + e[6] * A2
        return (y1 + e[7] * A2 + e[8] * A) @ (y1 + e[9] * y0 + e[10] * A) + e[11] * y1 + e[12] * y0 + e[13] * A2 + e[14] * A + e[15] * I

    taylor = {k: float(x) for k, x in t18.TAYLOR.items()}

    def T18(A, p=taylor):
        I = np.eye(A.shape[0]); A2 = A @ A; A3 = A2 @ A; A6 = A3 @ A3
        B1 = p["a11"] * A + p["a21"] * A2 + p["a31"] * A3; B5 = p["b24"] * A2 + p["b34"] * A3 + p["b64"] * A6
        B4 = p["b03"] * I + p["b13"] * A + p["b23"] * A2 + p["b33"] * A3 + p["b63"] * A6; A9 = B1 @ B5 + B4
        B3 = p["b02"] * I + p["b12"] * A + p["b22"] * A2 + p["b32"] * A3 + p["b62"] * A6
        B2 = p["b11"] * A + p["b21"] * A2 + p["b31"] * A3 + p["b61"] * A6
        return B2 + (B3 + A9) @ A9

    rng = np.random.default_rng(6)
    for n, s in ((9, 1.35), (9, 0.8), (27, 1.3), (36, 1.0)):
        M = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        H = (M + M.conj().T) / 2
        X = -1j * H * (s / np.linalg.norm(H, 2))
        ref = sl.expm(X)
        half = T18(X / 2)
        other = half @ half if s > 1.13 else T18(X)
        assert np.linalg.norm(E4(X) - ref, 2) < 1.5 * np.linalg.norm(other - ref, 2) + 1e-15, (n, s)
    for eps in (1e-6, 1e-2, 0.25):
        A = rng.normal(size=(81, 81)); A = (A - A.T) / 2; A *= 1.1 / np.linalg.norm(A, 2)
        E = rng.normal(size=(81, 81)); E = -(E @ E.T); E *= eps / np.abs(E).sum(axis=0).max()
        X = A + E
        ref = sl.expm(X)
        assert np.linalg.norm(E4(X) - ref, 2) < 1.5 * np.linalg.norm(T18(X) - ref, 2) + 1e-15, eps
    assert np.array_equal(E4(np.zeros((5, 5))), np.eye(5))

    # (4) the plan rule of c3p_pick_plan_mfma(nrm, 2.0, normal = true), restated: products of each candidate
    def squarings(theta, nrm):
        s = 0
        while theta * 2.0**s < nrm:
            s += 1
        return s

    for nrm, four in ((0.5, True), (1.35, True), (1.36, False), (2.0, False), (2.1, True), (2.7, True), (2.71, False), (4.0, False), (5.0, True)):
        assert (4 + squarings(1.35, nrm) < 5 + squarings(2.0, nrm)) == four, nrm


def test_t18_for_normal_generators():
    """c3p_common.h's c3p_t18_tab row 1 (round 6): (1) the values tools/gen_t18_normal.py solves for, (2) the scheme with those
    DOUBLE parameters reproduces e^{iy} on [-2, 2] to 5e-16 (exact expansion of the 5-product scheme), (3) on matrices: skew-Hermitian
    generators and real skew-symmetric ones with a dissipator-like symmetric part up to the accepted 0.25, without squaring, are as
    close to scipy's expm as the published Taylor parameters after one squaring."""
    import importlib.util
    import math
    from decimal import Decimal as Dc
    from fractions import Fraction as F

    import scipy.linalg as sl

    spec = importlib.util.spec_from_file_location("gen_t18n", os.path.join(ROOT, "tools", "gen_t18_normal.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    text = open(os.path.join(ROOT, "c3_amd", "csrc", "c3p_common.h")).read()
    assert "#define C3P_T18N_THETA 2.0" in text
    rows = re.search(r"c3p_t18_tab\[2\]\[20\] = \{\s*\{([^}]*)\},\s*\{([^}]*)\},", text)
    econ = [float.fromhex(x.strip()) for x in rows.group(2).split(",")]
    v, resid, dc, ds = t.solve(F(2))
    assert float(resid) < 1e-40 and dc < 1e-16 and ds < 1e-16
    assert econ == [float(x) for x in v]
    # the enum order of the header is the generator's parameter order
    assert re.search(r"enum \{ " + ", ".join("C3P_I_" + n.upper() for n in t.NAMES) + r" \};", text)
    coeff = t.t18_coeffs([Dc(x) for x in econ])
    worst = 0.0
    for i in range(-100, 101):
        y = Dc(2) * i / 100
        re_, im_, yk = Dc(0), Dc(0), Dc(1)
        for k in range(19):
            term = coeff[k] * yk
            if k % 4 == 0: re_ += term
            elif k % 4 == 1: im_ += term
            elif k % 4 == 2: re_ -= term
            else: im_ -= term
            yk *= y
        worst = max(worst, abs(complex(float(re_) - math.cos(float(y)), float(im_) - math.sin(float(y)))))
    assert worst < 5e-16, worst
    taylor = {k: float(x) for k, x in t.TAYLOR.items()}
    pe = dict(zip(t.NAMES, econ))

    def T18(A, p):
        I = np.eye(A.shape[0]); A2 = A @ A; A3 = A2 @ A; A6 = A3 @ A3
        B1 = p["a11"] * A + p["a21"] * A2 + p["a31"] * A3; B5 = p["b24"] * A2 + p["b34"] * A3 + p["b64"] * A6
        B4 = p["b03"] * I + p["b13"] * A + p["b23"] * A2 + p["b33"] * A3 + p["b63"] * A6; A9 = B1 @ B5 + B4
        B3 = p["b02"] * I + p["b12"] * A + p["b22"] * A2 + p["b32"] * A3 + p["b62"] * A6
        B2 = p["b11"] * A + p["b21"] * A2 + p["b31"] * A3 + p["b61"] * A6
        return B2 + (B3 + A9) @ A9

    rng = np.random.default_rng(5)
    for n, s in ((9, 2.0), (27, 1.6), (81, 2.0)):
        M = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        H = (M + M.conj().T) / 2
        X = -1j * H * (s / np.linalg.norm(H, 2))
        ref = sl.expm(X)
        half = T18(X / 2, taylor)
        assert np.linalg.norm(T18(X, pe) - ref, 2) < 1.5 * np.linalg.norm(half @ half - ref, 2) + 1e-15
    for eps in (1e-6, 1e-2, 0.25):
        A = rng.normal(size=(81, 81)); A = (A - A.T) / 2; A *= 1.7 / np.linalg.norm(A, 2)
        E = rng.normal(size=(81, 81)); E = -(E @ E.T); E *= eps / np.abs(E).sum(axis=0).max()
        X = A + E
        ref = sl.expm(X)
        half = T18(X / 2, taylor)
        assert np.linalg.norm(T18(X, pe) - ref, 2) < 1.5 * np.linalg.norm(half @ half - ref, 2) + 1e-15, eps
