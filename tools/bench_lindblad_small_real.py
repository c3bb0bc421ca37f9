"""Lindblad chains of one qubit / qutrit: the real Hermitian-basis kernels (c3p_smallr.hip, default for Hermitian Hamiltonians)
against the complex small-D kernels (option no_smallr).    python tools/bench_lindblad_small_real.py --out gpurun_out/lindblad_small_real.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import _lib, propagation as prop, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(x, device=dev)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return 1e3 * float(np.median(ts))


rows = []
for D, B, N in [(3, 64, 1000), (3, 256, 1000), (3, 1024, 1000), (2, 256, 1000), (2, 1024, 1000), (4, 64, 1000), (4, 256, 1000), (4, 1024, 1000)]:
    rng = np.random.default_rng(D * 100 + B)
    if D == 3:
        wl = workloads.make_workload(1, B=B, N=N)  # cfg1's qutrit
        h0, hks, sig, dt = wl.h0, wl.hks, wl.signals, wl.dt
        col = workloads.qubit_collapse_op(workloads.annihilator(3).astype(complex), 27e-6, 39e-6)[None]
    elif D == 4:
        # two coupled qubits, one drive each (lab frame), T1 on both
        sz, sx, sm, id2 = np.diag([0.0, 1.0]), np.array([[0, 1], [1, 0]], dtype=float), np.array([[0, 1], [0, 0]], dtype=float), np.eye(2)
        w1, w2, g = 5.0e9 * 2 * np.pi, 5.6e9 * 2 * np.pi, 20e6 * 2 * np.pi
        h0 = (w1 * np.kron(sz, id2) + w2 * np.kron(id2, sz) + g * np.kron(sx, sx)).astype(complex)
        hks = np.stack([np.kron(sx, id2), np.kron(id2, sx)]).astype(complex)
        sig = rng.normal(size=(B, 2, N)) * 2e8
        dt = 1e-11
        col = np.stack([np.sqrt(1 / 27e-6) * np.kron(sm, id2), np.sqrt(1 / 23e-6) * np.kron(id2, sm)]).astype(complex)
    else:
        h0 = np.diag([0.0, 5e9 * 2 * np.pi]).astype(complex)
        hks = np.array([[[0, 1], [1, 0]]], dtype=complex)
        sig = rng.normal(size=(B, 1, N)) * 2e8
        dt = 1e-11
        col = (np.sqrt(1 / 27e-6) * np.array([[0, 1], [0, 0]], dtype=complex))[None]
    K = hks.shape[0]
    ph = rng.uniform(0, 6, size=(B, D * D))
    h0d, hkd, sgd, cold, phd = t(h0), t(hks), t(sig), t(col), t(ph)
    f = lambda: prop.propagate_batch(h0d, hkd, sgd, dt, col_ops=cold, lindbladian=True, fr_phase=phd)
    row = {"case": f"Lindblad D={D} ({D*D}x{D*D})", "B": B, "N": N, "K": K}
    row["real_ms"] = timed(f)
    x = f()["U"].cpu().numpy()
    with _lib.options(no_smallr=1):
        row["complex_ms"] = timed(f)
        y = f()["U"].cpu().numpy()
    row["speedup"] = row["complex_ms"] / row["real_ms"]
    row["max_dev"] = float(np.abs(x - y).max())
    # the gradient (c3p_pwc_lindblad_vjp) the same way
    Ub = t(rng.normal(size=(B, D * D, D * D)) + 1j * rng.normal(size=(B, D * D, D * D)))
    gfn = lambda: prop.propagate_batch_lindblad_vjp(h0d, hkd, sgd, dt, cold, Ub, fr_phase=phd)
    row["vjp_real_ms"] = timed(gfn, 5)
    gx = gfn().cpu().numpy()
    with _lib.options(no_smallr=1):
        row["vjp_complex_ms"] = timed(gfn, 5)
        gy = gfn().cpu().numpy()
    row["vjp_speedup"] = row["vjp_complex_ms"] / row["vjp_real_ms"]
    row["vjp_max_rel_dev"] = float(np.abs(gx - gy).max() / np.abs(gy).max())
    row["propagators_per_s"] = B / row["real_ms"] * 1e3
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump({"what": "Lindblad chains of one qubit / qutrit, forward pass on resident inputs: real Hermitian-basis kernels against the complex small-D kernels (no_smallr = 1); ms per batch, median", "rows": rows}, open(a.out, "w"), indent=1)
