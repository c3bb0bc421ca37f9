"""Branch-B gradients (per-slice Hamiltonians in, their cotangents out): the on-chip general-generator sweeps against the tiled
sweep (C3P_TILED_GRAD=1) and the forward pass.    python tools/bench_grad_per_slice.py --out gpurun_out/grad_per_slice.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import _lib, propagation as prop

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="3:16:1000,9:16:1000,27:1:10000,27:16:2000,36:8:1000")
ap.add_argument("--out", default=None)
a = ap.parse_args()
rows = []
for case in a.cases.split(","):
    D, B, N = (int(x) for x in case.split(":"))
    rng = np.random.default_rng(D)
    h = rng.normal(size=(B, N, D, D)) + 1j * rng.normal(size=(B, N, D, D))
    Hs = torch.as_tensor(0.4 / np.sqrt(D) * (h + h.conj().transpose(0, 1, 3, 2)) / 2, device="cuda:0")
    Ubar = torch.as_tensor(rng.normal(size=(B, D, D)) + 1j * rng.normal(size=(B, D, D)), device="cuda:0")

    def timed(fn, reps=2):
        fn(); torch.cuda.synchronize(); best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best

    fwd = timed(lambda: prop.propagate_batch(Hs, None, None, 1.0))
    g = prop.propagate_per_slice_vjp(Hs, 1.0, Ubar)
    kern = _lib.last_kernel()
    vjp = timed(lambda: prop.propagate_per_slice_vjp(Hs, 1.0, Ubar))
    _lib.set_option("tiled_grad", "1")
    try:
        gt = prop.propagate_per_slice_vjp(Hs, 1.0, Ubar)
        tiled = timed(lambda: prop.propagate_per_slice_vjp(Hs, 1.0, Ubar), reps=1)
    finally:
        _lib.set_option("tiled_grad", None)
    row = {"D": D, "B": B, "N": N, "kernel": kern, "forward_ms": fwd * 1e3, "vjp_ms": vjp * 1e3, "vjp_over_forward": vjp / fwd,
           "tiled_vjp_ms": tiled * 1e3, "speedup_vs_tiled": tiled / vjp, "max_rel_diff_vs_tiled": float((g - gt).abs().max() / gt.abs().max())}
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.out, "w"), indent=1)
