import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
rng = np.random.default_rng(1)
for D in (33, 34, 36):
    a = rng.normal(size=(D, D)); h0 = (5e10 * (a + a.T) / 2).astype(complex)
    hks = np.stack([((lambda m: (m + m.T) / 2)(rng.normal(size=(D, D)))).astype(complex) for _ in range(2)])
    sig = rng.normal(size=(3, 2, 23)) * 2e9
    Ubar = rng.normal(size=(3, D, D)) + 1j * rng.normal(size=(3, D, D))
    ph = rng.uniform(0, 6, size=(3, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    with _lib.options(no_real_grad=1):
        g2 = np.asarray(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph))
    print("D", D, "real sweep vs general sweep rel diff", np.abs(g - g2).max() / np.abs(g2).max())
for D in (33, 35, 36):
    w5 = make_workload(5, B=3, N=29)
    # cfg5's (real) operators cut to the leading D x D block: Hermitian, real, the drive strength of the configuration
    h0, hks = np.ascontiguousarray(w5.h0[:D, :D]), np.ascontiguousarray(w5.hks[:, :D, :D])
    Ubar = rng.normal(size=(3, D, D)) + 1j * rng.normal(size=(3, D, D))
    ph = rng.uniform(0, 6, size=(3, D))
    g = np.asarray(prop.propagate_batch_vjp(h0, hks, w5.signals, w5.dt, Ubar, fr_phase=ph))
    with _lib.options(no_real_grad=1):
        g2 = np.asarray(prop.propagate_batch_vjp(h0, hks, w5.signals, w5.dt, Ubar, fr_phase=ph))
    print("cfg5 operators, D", D, "real sweep vs general sweep rel diff", np.abs(g - g2).max() / np.abs(g2).max())
w = make_workload(5, B=256)
t = lambda x: torch.as_tensor(x, device="cuda:0")
h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
Ubar = torch.randn(256, w.D, w.D, dtype=torch.complex128, device="cuda:0")
f = lambda: prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar, fr_phase=ph)
f(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(os.path.basename(_lib.LIB_PATH), "cfg5 gradient B=256 ms", 1e3 * min(ts))
