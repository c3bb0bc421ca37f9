import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import propagation as prop, _lib
from oracle import c3_oracle
rng = np.random.default_rng(3)
t = lambda a: torch.as_tensor(a, device="cuda:0")
for D in (tuple(int(x) for x in sys.argv[1:]) or (13, 20, 27, 36)):
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0 = np.diag(rng.uniform(0, 1, D)).astype(complex) + herm(0.05); hks = np.stack([herm(0.3) for _ in range(2)])
    B, N = 64, 200
    sig = rng.uniform(-1, 1, size=(B, 2, N))
    one = lambda h: np.abs(h - np.trace(h) / D * np.eye(D)).sum(axis=0).max()
    bound = one(h0) + sum(np.abs(sig[:, k, :]).max() * one(hks[k]) for k in range(2))
    for target in (1.0, 1.5, 1.95, 2.05, 3.5):
        dt = target / bound
        res = {}
        for name, opt in (("t18n", None), ("taylor", 1)):
            _lib.set_option("no_t18n", opt)
            a = (t(h0), t(hks), t(sig), dt)
            U = prop.propagate_batch(*a)["U"]; torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): prop.propagate_batch(*a)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
            ref = c3_oracle.propagate_batch(h0, hks, sig[:3], dt)
            err = max(np.linalg.norm(U[b].cpu().numpy() - ref[b]) for b in range(3))
            res[name] = (round(ms, 3), float("%.1e" % err))
        _lib.set_option("no_t18n", None)
        print(D, target, res, flush=True)
