"""Lindblad gradients at small superoperator dimensions (D^2 <= 36): c3p_pwc_lindblad_vjp on the three-kernel general sweep
(c3p_grad.hip) against the tiled sweep (C3P_TILED_GRAD=1) and the forward pass, same inputs.
    python tools/bench_grad_lindblad.py --out gpurun_out/grad_lindblad_small.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from c3_amd import _lib, propagation as prop

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="2:256:1000,3:256:1000,3:16:1000,4:256:1000,5:64:500,6:64:500")
ap.add_argument("--tiled-max-n", type=int, default=1000)
ap.add_argument("--no-tiled", action="store_true", help="skip the tiled-sweep comparison (profiling runs)")
ap.add_argument("--out", default=None)
a = ap.parse_args()
rows = []
dev = "cuda:0"
for case in a.cases.split(","):
    D, B, N = (int(x) for x in case.split(":"))
    rng = np.random.default_rng(D)
    herm = lambda s: (lambda m: s * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    K, Dm = 2, D * D
    t = lambda x: torch.as_tensor(x, device=dev)
    h0, hks = t(herm(0.8)), t(np.stack([herm(0.5) for _ in range(K)]))
    col = t(np.stack([0.2 * (rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))]))
    sig = t(rng.uniform(-1, 1, size=(B, K, N)))
    Ubar = t(rng.normal(size=(B, Dm, Dm)) + 1j * rng.normal(size=(B, Dm, Dm)))
    dt = 0.05

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best

    fwd = timed(lambda: prop.propagate_batch(h0, hks, sig, dt, col_ops=col, lindbladian=True))
    g = prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar)
    kern = _lib.last_kernel()
    vjp = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar))
    gt, tiled = g, float("nan")
    if not a.no_tiled:
        _lib.set_option("tiled_grad", "1")
        try:
            gt = prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar)
            tiled = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, dt, col, Ubar), reps=1)
        finally:
            _lib.set_option("tiled_grad", None)
    row = {"D": D, "Dm": Dm, "B": B, "N": N, "kernel": kern, "forward_ms": fwd * 1e3, "vjp_ms": vjp * 1e3, "vjp_over_forward": vjp / fwd,
           "tiled_vjp_ms": tiled * 1e3, "speedup_vs_tiled": tiled / vjp, "gradients_per_s": B / vjp,
           "max_rel_diff_vs_tiled": float((g - gt).abs().max() / gt.abs().max())}
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.out, "w"), indent=1)
