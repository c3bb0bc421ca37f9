import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import propagation as prop, _lib
rng = np.random.default_rng(1)
def run(D, B, N=1000, real=True):
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + (0 if real else 1j) * rng.normal(size=(D, D)))
    h0 = np.diag(rng.uniform(0, 1, D)).astype(complex) + herm(0.02); hks = np.stack([herm(0.05) for _ in range(2)])
    sig = rng.uniform(-1, 1, size=(B, 2, N)); dt = 0.08
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    Ub = t(rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D)))
    a = (t(h0.astype(complex)), t(hks.astype(complex)), t(sig), dt, Ub)
    out = {}
    for name, opt in (("valu", None), ("tiled", "tiled_grad")):
        if opt: _lib.set_option(opt, 1)
        prop.propagate_batch_vjp(*a); torch.cuda.synchronize()
        t0 = time.perf_counter(); prop.propagate_batch_vjp(*a); torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) * 1e3, 1); out[name + "_k"] = _lib.last_kernel_detail()[:60]
        if opt: _lib.set_option(opt, None)
    t0 = time.perf_counter(); prop.propagate_batch(a[0], a[1], a[2], dt); torch.cuda.synchronize(); out["fwd"] = round((time.perf_counter() - t0) * 1e3, 1)
    print(D, B, real, out, flush=True)
for D in (48, 64):
    for B in (64, 256):
        run(D, B)
run(48, 256, real=False)
