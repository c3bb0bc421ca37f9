"""Mid-D kernel: real-Hamiltonian instance vs the complex instance (C3P_NO_REAL=1) across the geometry classes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import propagation as p
from c3_amd import _lib

dev = torch.device("cuda:0")
out = {}
for D in (13, 16, 20, 24, 27, 32, 36, 40):
    rng = np.random.default_rng(D)
    B, K, N = 768, 2, 400
    a = rng.normal(size=(D, D)); h0 = (a + a.T) * 1e10 / D
    hk = rng.normal(size=(K, D, D)); hks = hk + np.swapaxes(hk, -1, -2)
    sig = rng.normal(size=(B, K, N)) * 2e8
    t = lambda x, dt_: torch.as_tensor(np.asarray(x, dtype=dt_), device=dev)
    H0, HK, S = t(h0, np.complex128), t(hks, np.complex128), t(sig, np.float64)
    res = {}
    for mode in ("real", "complex"):
        if mode == "complex":
            _lib.set_option("no_real", "1")
        else:
            _lib.set_option("no_real", None)
        for _ in range(3):
            U = p.propagate_batch(H0, HK, S, 1e-11)["U"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            U = p.propagate_batch(H0, HK, S, 1e-11)["U"]
        torch.cuda.synchronize()
        res[mode + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        res[mode + "_U"] = U.cpu().numpy()
    res["max_abs_diff"] = float(np.abs(res.pop("real_U") - res.pop("complex_U")).max())
    res["speedup"] = res["complex_ms"] / res["real_ms"]
    out[D] = res
_lib.set_option("no_real", None)
print(json.dumps(out))
