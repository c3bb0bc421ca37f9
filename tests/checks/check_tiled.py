"""GPU check + timing of the tiled large-matrix path (c3p_tiled.hip) against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from c3_amd import propagation, workloads, _lib
from oracle import c3_oracle as o

def rand_herm(rng, D, scale):
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    return scale * (a + a.conj().T) / 2

worst = 0.0
rng = np.random.default_rng(5)
for D, B, N, K, sc in [(100, 3, 7, 2, 0.02), (128, 2, 5, 1, 0.05), (93, 2, 4, 2, 0.01), (200, 2, 3, 2, 0.01), (300, 1, 2, 1, 0.004)]:
    h0 = rand_herm(rng, D, sc); hks = np.stack([rand_herm(rng, D, sc / 2) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N)); ph = rng.uniform(0, 6, size=(B, D))
    r = propagation.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph, want_dUs=True)
    ref = o.propagate_batch(h0, hks, sig, 1.0, fr_phase=ph)
    err = max(np.linalg.norm(r["U"][b] - ref[b]) for b in range(B))
    d = o.tf_propagation_vectorized(h0, hks, sig[0], 1.0)
    e2 = np.abs(np.asarray(r["dUs"][0]) - d).max()
    print(f"unitary D={D} B={B} N={N}: kernel={_lib.last_kernel()} err={err:.2e} dUs={e2:.2e}", flush=True)
    worst = max(worst, err, e2)
# per-slice Hamiltonians (branch B) at D = 45 and 120
for D in (45, 120):
    H = np.stack([rand_herm(rng, D, 0.03) for _ in range(5)])
    r = propagation.propagate_batch(H, None, None, 1.0)
    ref = o.pwc_arrays(H, None, None, 1.0)["U"] if False else None
    U = np.eye(D, dtype=complex)
    for n in range(5):
        U = o.expm(-1j * H[n]) @ U
    err = np.linalg.norm(np.asarray(r["U"][0]) - U)
    print(f"per-slice D={D}: kernel={_lib.last_kernel()} err={err:.2e}", flush=True)
    worst = max(worst, err)
# Lindblad D = 27 -> 729 x 729 (the reference's skipped case), tiny N
D, B, N, K = 27, 2, 2, 2
wl3 = workloads.make_workload(3, B=B, N=N)
a = workloads.annihilators((3, 3, 3))
T = workloads.dressing_transform(workloads.bare_drift((3, 3, 3), (5.0e9, 5.6e9, 6.2e9), (-210e6, -240e6, -235e6), {(0, 1): 20e6, (0, 2): 20e6, (1, 2): 20e6}))
col = np.stack([workloads.dress(workloads.qubit_collapse_op(a[q], (27e-6, 23e-6, 25e-6)[q], (39e-6, 31e-6, 35e-6)[q]), T) for q in range(3)])
hks2 = wl3.hks[:K]
t0 = time.perf_counter()
r = propagation.propagate_batch(wl3.h0, hks2, wl3.signals[:, :K], wl3.dt, col_ops=col, lindbladian=True)
t1 = time.perf_counter()
ref = o.propagate_batch(wl3.h0, hks2, wl3.signals[:, :K], wl3.dt, col_ops=col, lindbladian=True)
err = max(np.linalg.norm(r["U"][b] - ref[b]) for b in range(B))
print(f"lindblad 729 B={B} N={N}: kernel={_lib.last_kernel()} err={err:.2e} ({t1 - t0:.2f} s incl. first-call setup)", flush=True)
worst = max(worst, err)
print("WORST", worst)
if "--time" in sys.argv:
    dev = torch.device("cuda:0")
    B, N = 8, 40
    wl3 = workloads.make_workload(3, B=B, N=N)
    args = [torch.as_tensor(x, device=dev) for x in (wl3.h0, wl3.hks[:K], wl3.signals[:, :K], col)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        propagation.propagate_batch(args[0], args[1], args[2], wl3.dt, col_ops=args[3], lindbladian=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        flop = B * N * 7 * 8 * 729**3
        print(f"lindblad 729 B={B} N={N}: {t1 - t0:.3f} s -> {flop / (t1 - t0) / 1e12:.1f} TFLOP/s (7 products per slice)", flush=True)
    for D, B, N in [(128, 64, 100), (256, 32, 50)]:
        h0 = rand_herm(rng, D, 0.02); hks = np.stack([rand_herm(rng, D, 0.01) for _ in range(2)])
        sig = rng.uniform(-1, 1, size=(B, 2, N))
        a3 = [torch.as_tensor(x, device=dev) for x in (h0, hks, sig)]
        for env in ("", "1"):
            if env: _lib.set_option("no_tiled", "1")
            else: _lib.set_option("no_tiled", None)
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                propagation.propagate_batch(a3[0], a3[1], a3[2], 1.0)
                torch.cuda.synchronize(); t1 = time.perf_counter()
            print(f"unitary D={D} B={B} N={N} {'generic' if env else 'tiled'}: {t1 - t0:.3f} s", flush=True)
        _lib.set_option("no_tiled", None)
sys.exit(0 if worst < 1e-10 else 1)
