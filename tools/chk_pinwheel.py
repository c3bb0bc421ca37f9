"""Pinwheel deal of the D = 25..28 real class (cfg3): forward against the oracle (all three polynomial variants, dUs),
real backward sweep against the general one, and the cfg3 rate.  python tools/chk_pinwheel.py [library]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
from oracle import c3_oracle
rng = np.random.default_rng(7)
sym = lambda m: (m + m.T) / 2
for D in (25, 26, 27, 28):
    for amp in (0.3, 1.0, 1.4, 3.0, 9.0):  # scaled norms across the degree-16 / 18 / 20 thresholds and 1-3 squarings
        h0 = (amp * 1e10 * sym(rng.normal(size=(D, D)))).astype(complex)
        hks = np.stack([sym(rng.normal(size=(D, D))).astype(complex) for _ in range(2)])
        sig = rng.normal(size=(3, 2, 37)) * amp * 4e9
        ph = rng.uniform(0, 6, size=(3, D))
        r = prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph, want_dUs=True)
        U = np.asarray(r["U"]); dUs = np.asarray(r["dUs"])
        ref = c3_oracle.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)
        refd = np.stack([c3_oracle.pwc_arrays(h0, hks, sig[b], 1e-11)["dUs"] for b in range(3)])
        eU = max(np.linalg.norm(U[b] - ref[b]) for b in range(3))
        ed = np.abs(dUs - refd).max()
        r2 = prop.propagate_batch(h0, hks, sig, 1e-11, fr_phase=ph)
        eU = max(eU, max(np.linalg.norm(np.asarray(r2["U"])[b] - ref[b]) for b in range(3)))
        Ubar = rng.normal(size=(3, D, D)) + 1j * rng.normal(size=(3, D, D))
        g = np.asarray(torch.as_tensor(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph)).cpu())
        with _lib.options(no_real_grad=1):
            g2 = np.asarray(torch.as_tensor(prop.propagate_batch_vjp(h0, hks, sig, 1e-11, Ubar, fr_phase=ph)).cpu())
        print(f"D {D} amp {amp}: kernel {_lib.last_kernel()} |U-ref|_F {eU:.2e} max|dU-ref| {ed:.2e} grad real vs general {np.abs(g - g2).max() / np.abs(g2).max():.2e}", flush=True)
t = lambda x: torch.as_tensor(x, device="cuda:0")
for B in (256, 512):
    w = make_workload(3, B=B)
    h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
    f = lambda: prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(os.path.basename(_lib.LIB_PATH), f"cfg3 B={B} ms {1e3 * min(ts):.3f} propagators/s {B * w.N / min(ts):.4e}", flush=True)
w = make_workload(3, B=256)
h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
Ubar = torch.randn(256, w.D, w.D, dtype=torch.complex128, device="cuda:0")
f = lambda: prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar, fr_phase=ph)
f(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(os.path.basename(_lib.LIB_PATH), "cfg3 gradient B=256 ms", 1e3 * min(ts))
