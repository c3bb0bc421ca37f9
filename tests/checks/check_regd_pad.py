"""Timing of the zero-padded classes of the register-resident kernel against the arena kernel (C3P_REGD_PAD=0) per matrix
dimension: which dimensions are worth padding.   python tests/checks/check_regd_pad.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from c3_amd import propagation as prop
from c3_amd import _lib

dev = "cuda:0"
B, N, K = 256, 200, 2
for D in (41, 43, 45, 47, 48, 50, 53, 56, 58, 60, 64, 66, 68, 70, 72, 76, 80):
    rng = np.random.default_rng(D)
    herm = lambda s: (lambda a: s * (a + a.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    h0, hks = torch.as_tensor(herm(0.08), device=dev), torch.as_tensor(np.stack([herm(0.05) for _ in range(K)]), device=dev)
    sig = torch.as_tensor(rng.uniform(-1, 1, size=(B, K, N)), device=dev)
    out = {}
    for mode in ("all", "0"):
        _lib.set_option("regd_pad", mode)
        prop.propagate_batch(h0, hks, sig, 1.0); torch.cuda.synchronize()
        t0 = time.perf_counter(); U = prop.propagate_batch(h0, hks, sig, 1.0)["U"]; torch.cuda.synchronize()
        out[mode] = (time.perf_counter() - t0, U)
    _lib.set_option("regd_pad", None)
    d = float((out["all"][1] - out["0"][1]).abs().max())
    print(f"D={D}: padded regd {out['all'][0]*1e3:8.2f} ms   arena {out['0'][0]*1e3:8.2f} ms   speedup {out['0'][0]/out['all'][0]:.2f}   |diff| {d:.1e}", flush=True)
