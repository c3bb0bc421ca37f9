import os, sys, time
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
for B in (4, 16, 64, 256):
    w = make_workload(4, B=B, N=300); Dm = w.D * w.D
    h0, hks, sig, col = t(w.h0), t(w.hks), t(w.signals), t(w.col_ops)
    Ubar = torch.randn(B, Dm, Dm, dtype=torch.complex128, device=dev)
    r = {}
    for v in ("0", "1"):
        _lib.set_option("tiled_tile32", v)
        g = prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)
        r[v] = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)) * 1e3; r["g" + v] = g
    print("cfg4 N=300 B=%d: tile64 %.1f ms, tile32 %.1f ms, x%.2f, rel diff %.1e" % (B, r["0"], r["1"], r["0"] / r["1"], float((r["g0"] - r["g1"]).abs().max() / r["g0"].abs().max())), flush=True)
