"""Lindblad ODE steps at cfg5's dimension (D = 36, two collapse operators): matrix-core kernel against the workgroup kernel."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
rng = np.random.default_rng(1)
t = lambda a: torch.as_tensor(a, device="cuda:0")
for D, C in ((36, 2), (48, 1), (33, 3)):
    herm = lambda s: (lambda a: (s * (a + a.conj().T) / 2).astype(complex))(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0, hks = herm(1.0), np.stack([herm(0.4) for _ in range(2)])
    col = np.stack([0.2 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))) for _ in range(C)])
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)); rho = a @ a.conj().T; rho /= np.trace(rho)
    for B in (16, 256):
        sig = rng.normal(size=(B, 2, 400))
        args = (t(h0), t(hks), t(sig), 0.01, t(rho))
        for solver in ("rk4", "tsit5"):
            res = []
            for opt in ({}, {"ode_lind_wg": 1}):
                with _lib.options(**opt):
                    f = lambda: prop.ode_solve_batch(*args, solver, "lindblad", col_ops=t(col), final_only=True)
                    x = f(); torch.cuda.synchronize(); ts = []
                    for _ in range(3):
                        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                res.append((1e3 * min(ts), _lib.last_kernel(), x))
            print(f"D={D} C={C} B={B} {solver}: {res[0][1]} {res[0][0]:.2f} ms, {res[1][1]} {res[1][0]:.2f} ms, x{res[1][0] / res[0][0]:.2f}, diff {float((res[0][2] - res[1][2]).abs().max()):.1e}", flush=True)
