"""A/B of the evaluation schemes for normal generators on the complex loops: default (four-product scheme where it saves a product),
no_t18n = 2 (T18 with the economised parameters only), no_t18n = 1 (published T18 parameters).
    python tools/ab_e4n.py            # cfg2 with a complex control operator, then D = 12, 20, 27, 36 at B = 256, N = 400"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib, propagation as prop
from c3_amd.workloads import make_workload
from oracle import c3_oracle
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()


def timed(f, reps=30, rounds=5):
    for _ in range(10): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / reps)
    return min(ts) * 1e3


def case(name, h0, hks, sig, dt, ph=None, reps=30):
    a = (t(h0), t(hks), t(sig), dt)
    kw = {} if ph is None else {"fr_phase": t(ph)}
    ref = c3_oracle.propagate_batch(h0, hks, sig[:2], dt, fr_phase=None if ph is None else ph[:2])
    out = []
    for label, opt in (("four-product", None), ("T18N", 2), ("T18", 1), ("four-product", None), ("T18N", 2)):
        _lib.set_option("no_t18n", opt)
        ms = timed(lambda: prop.propagate_batch(*a, **kw), reps)
        U = prop.propagate_batch(*a, **kw)["U"][:2].cpu().numpy()
        out.append(f"{label} {ms:.4f} ms (err {max(np.linalg.norm(U[b] - ref[b]) for b in range(2)):.1e})")
    _lib.set_option("no_t18n", None)
    print(name, "|", " | ".join(out), flush=True)


w = make_workload(2, B=256)
k = min(1, w.K - 1)
up = np.triu(w.hks[k].real, 1)
hks = w.hks.copy(); hks[k] = hks[k] + 0.3j * (up - up.T)
case("cfg2 complex operator (B 256, N 1000)", w.h0, hks, w.signals, w.dt, w.fr_phase, reps=50)
rng = np.random.default_rng(3)
for D in (9, 12, 20, 27, 36):
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0 = np.diag(rng.uniform(0, 1, D)).astype(complex) + herm(0.05); hk = np.stack([herm(0.3) for _ in range(2)])
    B, N = 256, 400
    sig = rng.uniform(-1, 1, size=(B, 2, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    bound = one(h0) + sum(np.abs(sig[:, j, :]).max() * one(hk[j]) for j in range(2))
    for target in (1.0, 1.3, 2.5):
        case(f"D={D} norm bound {target}", h0, hk, sig, target / bound, reps=10)
