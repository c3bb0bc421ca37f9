import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from c3_amd import propagation as prop
from c3_amd import _lib
from c3_amd.workloads import make_workload
dev = "cuda:0"; t = lambda x: torch.as_tensor(x, device=dev)
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for B in (16, 64, 256):
    w = make_workload(4, B=B, N=400); Dm = w.D * w.D
    h0, hks, sig, col = t(w.h0), t(w.hks), t(w.signals), t(w.col_ops)
    Ubar = torch.randn(B, Dm, Dm, dtype=torch.complex128, device=dev)
    r = {}
    for tag, env in (("batched", None), ("per_product", "C3P_TILED_NO_BATCH")):
        if env: os.environ[env] = "1"
        g = prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)
        r[tag] = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)) * 1e3
        r[tag + "_g"] = g
        if env: os.environ.pop(env)
    print("cfg4 N=400 B=%d: batched %.1f ms, per product %.1f ms, x%.2f, rel diff %.1e" % (B, r["batched"], r["per_product"], r["per_product"] / r["batched"], float((r["batched_g"] - r["per_product_g"]).abs().max() / r["per_product_g"].abs().max())), flush=True)
_lib.set_option("tiled_grad", "1")
for D, B in ((48, 64), (64, 64), (64, 512)):
    rng = np.random.default_rng(D)
    herm = lambda sc: (lambda m: sc * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    N, K = 300, 2
    h0, hks = t(herm(0.12)), t(np.stack([herm(0.08) for _ in range(K)])); sig = t(rng.uniform(-1, 1, size=(B, K, N)))
    Ubar = torch.randn(B, D, D, dtype=torch.complex128, device=dev)
    r = {}
    for tag, env in (("batched", None), ("per_product", "C3P_TILED_NO_BATCH")):
        if env: os.environ[env] = "1"
        r[tag] = timed(lambda: prop.propagate_batch_vjp(h0, hks, sig, 1.0, Ubar)) * 1e3
        if env: os.environ.pop(env)
    print("unitary D=%d N=300 B=%d: batched %.1f ms, per product %.1f ms, x%.2f" % (D, B, r["batched"], r["per_product"], r["per_product"] / r["batched"]), flush=True)
