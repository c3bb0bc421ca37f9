"""Real-Hamiltonian mid-D classes, forward + gradient timing per D: python tools/time_midd_classes.py <library>"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
rng = np.random.default_rng(3)
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(20): x @ x
torch.cuda.synchronize()
out = []
for D in (13, 16, 20, 24, 32, 36, 40):
    sym = lambda m: (m + m.T) / 2
    h0 = t((1e10 * sym(rng.normal(size=(D, D)))).astype(complex)); hks = t(np.stack([sym(rng.normal(size=(D, D))).astype(complex) for _ in range(2)]))
    sig = t(rng.normal(size=(256, 2, 400)) * 4e9)
    f = lambda: prop.propagate_batch(h0, hks, sig, 1e-11)
    Ubar = torch.randn(256, D, D, dtype=torch.complex128, device="cuda:0")
    g = lambda: prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar)
    res = []
    for fn in (f, g):
        fn(); fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.append(1e3 * min(ts))
    out.append(f"D={D}: fwd {res[0]:.3f} ms grad {res[1]:.3f} ms")
print(os.path.basename(_lib.LIB_PATH), " | ".join(out))
