"""ODE shapes that ran on the round-1 workgroup kernel until round 4: more than four control lines (D <= 16) and supplied
per-sample-index Hamiltonians (rk4_unitary, branch B) -- lane-row kernels against the workgroup kernel (option ode_wg).
    python tools/bench_ode_fallbacks.py --out gpurun_out/ode_fallbacks.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import _lib, propagation as prop

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(x, device=dev)
rng = np.random.default_rng(5)


def herm(D, s, real=False):
    m = rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D))
    return (s * (m + m.conj().T) / 2).astype(complex)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


rows = []
for D, K, B, N, solver, step in [(9, 6, 256, 1000, "rk4", "schrodinger"), (9, 6, 2048, 1000, "rk4", "schrodinger"), (9, 6, 256, 1000, "tsit5", "schrodinger"),
                                 (16, 8, 256, 1000, "rk4", "schrodinger"), (16, 8, 2048, 1000, "rk4", "schrodinger"),
                                 (9, 6, 256, 1000, "rk4", "von_neumann"), (9, 6, 2048, 1000, "rk4", "von_neumann"), (9, 6, 2048, 1000, "tsit5", "von_neumann"), (9, 6, 16384, 200, "rk4", "von_neumann"), (9, 4, 2048, 1000, "rk4", "von_neumann"), (9, 4, 16384, 200, "rk4", "von_neumann"),
                                 (9, 6, 256, 1000, "rk4", "von_neumann_real"), (9, 6, 2048, 1000, "rk4", "von_neumann_real"), (9, 4, 2048, 1000, "rk4", "von_neumann_real")]:
    real = step.endswith("_real")
    step = step.replace("_real", "")
    h0, hks = t(herm(D, 0.3, real)), t(np.stack([herm(D, 0.2, real) for _ in range(K)]))
    sig = t(rng.uniform(-1, 1, size=(B, K, N)))
    psi = rng.normal(size=(B, D, 1)) + 1j * rng.normal(size=(B, D, 1))
    init = t(psi if step == "schrodinger" else np.einsum("bik,bjk->bij", psi, psi.conj()))
    f = lambda: prop.ode_solve_batch(h0, hks, sig, 0.05, init, solver, step, final_only=True)
    row = {"case": f"{step} {solver} D={D} K={K}" + (" real operators" if real else ""), "B": B, "N": N}
    row["lane_row_ms"] = timed(f)
    row["kernel"] = _lib.last_kernel()
    x = f().cpu().numpy()
    with _lib.options(ode_wg=1):
        row["workgroup_ms"] = timed(f, 3)
        y = f().cpu().numpy()
    row["speedup"] = row["workgroup_ms"] / row["lane_row_ms"]
    row["max_dev"] = float(np.abs(x - y).max())
    row["rk_steps_per_s"] = B * N / row["lane_row_ms"] * 1e3
    rows.append(row)
    print(json.dumps(row), flush=True)
# Lindblad steps at 33 <= D <= 48 (round 5): matrix-core kernel with the collapse operators read from memory against the
# workgroup kernel these shapes ran on until then (option ode_lind_wg)
for D, C, B, N, solver in [(36, 2, 16, 400, "rk4"), (36, 2, 256, 400, "rk4"), (36, 2, 256, 400, "tsit5"), (48, 1, 256, 400, "rk4"), (33, 3, 256, 400, "rk4")]:
    h0, hks = t(herm(D, 0.3)), t(np.stack([herm(D, 0.2) for _ in range(2)]))
    col = t(np.stack([0.05 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)]))
    sig = t(rng.uniform(-1, 1, size=(B, 2, N)))
    a_ = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    rho = a_ @ a_.conj().T
    init = t(rho / np.trace(rho))
    f = lambda: prop.ode_solve_batch(h0, hks, sig, 0.02, init, solver, "lindblad", col_ops=col, final_only=True)
    row = {"case": f"lindblad {solver} D={D} K=2 C={C}", "B": B, "N": N}
    row["lane_row_ms"] = timed(f)
    row["kernel"] = _lib.last_kernel()
    x = f().cpu().numpy()
    with _lib.options(ode_lind_wg=1):
        row["workgroup_ms"] = timed(f, 3)
        y = f().cpu().numpy()
    row["speedup"] = row["workgroup_ms"] / row["lane_row_ms"]
    row["max_dev"] = float(np.abs(x - y).max())
    row["rk_steps_per_s"] = B * N / row["lane_row_ms"] * 1e3
    rows.append(row)
    print(json.dumps(row), flush=True)
for D, Ns in [(9, 2001), (16, 2001)]:
    Hs = rng.normal(size=(Ns, D, D)) + 1j * rng.normal(size=(Ns, D, D))
    Hs = t(0.2 * (Hs + Hs.conj().transpose(0, 2, 1)))
    f = lambda: prop._rk4_unitary_device(Hs=Hs, dt=0.05, want_dUs=False)
    row = {"case": f"rk4_unitary, supplied Hamiltonians D={D}", "B": 1, "N": (Ns - 1) // 2}
    row["lane_row_ms"] = timed(f)
    row["kernel"] = _lib.last_kernel()
    x = f()[0].cpu().numpy()
    with _lib.options(ode_wg=1):
        row["workgroup_ms"] = timed(f, 3)
        y = f()[0].cpu().numpy()
    row["speedup"] = row["workgroup_ms"] / row["lane_row_ms"]
    row["max_dev"] = float(np.abs(x - y).max())
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump({"what": "ODE calls with K > 4 control lines / supplied Hamiltonians: lane-row kernels (round 4) against the workgroup kernel (ode_wg = 1); ms per call, median", "rows": rows}, open(a.out, "w"), indent=1)
