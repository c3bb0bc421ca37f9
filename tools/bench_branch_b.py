"""Branch B (per-slice Hamiltonians handed over whole) and c3p_expm: MFMA kernels vs the generic kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
out = {}
for cfg, B, N in ((2, 256, 1000), (3, 64, 500)):
    w = make_workload(cfg, B=B, N=N)
    dev = "cuda:0"
    h0, hks, sig = (torch.as_tensor(x, device=dev) for x in (w.h0, w.hks, w.signals))
    H = h0[None, None] + torch.einsum("bkn,kij->bnij", sig.to(torch.complex128), hks)  # [B,N,D,D]
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    a = timed(lambda: prop.propagate_batch(h0, hks, sig, w.dt))
    b = timed(lambda: prop.propagate_batch(H, None, None, w.dt))
    c = timed(lambda: prop.propagate_batch(H, None, None, w.dt, force_generic=True))
    X = (-1j * w.dt) * H.reshape(-1, w.D, w.D)
    e = timed(lambda: prop.expm(X))
    f = timed(lambda: prop.expm(X, force_generic=True))
    Ua = prop.propagate_batch(h0, hks, sig, w.dt)["U"]; Ub = prop.propagate_batch(H, None, None, w.dt)["U"]
    out[w.name] = {"B": B, "N": N, "branchA_ms": a, "branchB_mfma_ms": b, "branchB_generic_ms": c, "expm_mfma_ms": e, "expm_generic_ms": f,
                   "expm_matrices_per_s": B * N / e * 1e3, "A_vs_B_max_abs": float((Ua - Ub).abs().max())}
print(json.dumps(out))
