"""Time propagate_batch for random Hermitian systems of a given dimension (kernel coverage check)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import propagation as P, _lib
from oracle import c3_oracle as o

def run(D, B, N, K=2, lind=False, check=True, generic=False):
    rng = np.random.default_rng(D)
    h0 = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)); h0 = (h0 + h0.conj().T) * (3e10 / D)
    hks = rng.normal(size=(K, D, D)) + 1j * rng.normal(size=(K, D, D)); hks = hks + np.conj(np.swapaxes(hks, -1, -2))
    sig = rng.normal(size=(B, K, N)) * 1e9 / D
    col = None
    if lind:
        col = (rng.normal(size=(2, D, D)) + 1j * rng.normal(size=(2, D, D))) * 3e3
    dt = 1e-11
    dev = torch.device("cuda:0")
    bp = P.BatchPropagator(torch.as_tensor(h0, device=dev), torch.as_tensor(hks, device=dev), torch.as_tensor(sig, device=dev), dt,
                           col_ops=None if col is None else torch.as_tensor(col, device=dev), force_generic=generic)
    bp.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): bp.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    err = float("nan")
    if check:
        ref = o.propagate_batch(h0, hks, sig[:2], dt, col_ops=col, lindbladian=lind)
        U = bp.U[:2].cpu().numpy()
        err = max(np.linalg.norm(U[b] - ref[b]) for b in range(2))
    Dm = D * D if lind else D
    print(f"D={D:3d} Dm={Dm:3d} B={B} N={N} kernel={_lib.last_kernel():14s} {ms:9.3f} ms  {B/ms*1e3:12.1f} props/s  err={err:.2e}", flush=True)

if __name__ == "__main__":
    for spec in sys.argv[1:]:
        parts = spec.split(",")
        D, B, N = int(parts[0]), int(parts[1]), int(parts[2])
        lind = "L" in parts[3:]
        generic = "G" in parts[3:]
        run(D, B, N, lind=lind, generic=generic)
