"""Lindblad control gradient at cfg4's operators (81 x 81 superoperators, N = 1000): the on-chip backward sweep in the
Hermitian basis (c3p_regrg.hip) next to the forward pass and, for small batches, the tiled sweep it replaces; every Taylor
degree of the pair evaluation forced in turn (A/B of the Horner depth against the squaring count).
    python tools/bench_grad_lindblad_hb.py --out gpurun_out/final/grad_lindblad_hb.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import propagation as prop
from c3_amd import _lib
from c3_amd.workloads import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--batches", default="16,64,256,512")
ap.add_argument("--tiled-up-to", type=int, default=64)
ap.add_argument("--degrees", default="8,12,16,20")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--N", type=int, default=1000)
a = ap.parse_args()
dev = "cuda:0"
t = lambda x: torch.as_tensor(x, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


rows = []
for B in (int(x) for x in a.batches.split(",")):
    w = make_workload(4, B=B, N=a.N)
    Dm = w.D * w.D
    h0, hks, sig, col = t(w.h0), t(w.hks), t(w.signals), t(w.col_ops)
    Ubar = torch.randn(B, Dm, Dm, dtype=torch.complex128, device=dev)
    tf = timed(lambda: prop.propagate_batch(h0, hks, sig, w.dt, col_ops=col, lindbladian=True))
    g = prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)
    tg = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar))
    row = {"case": "cfg4 Lindblad 81x81", "B": B, "N": w.N, "forward_ms": tf * 1e3, "vjp_ms": tg * 1e3, "vjp_over_forward": tg / tf, "gradients_per_s": B / tg}
    if a.degrees:
        row["vjp_ms_by_degree"] = {}
        for deg in (int(x) for x in a.degrees.split(",")):
            with _lib.options(regr_grad_degree=deg):
                row["vjp_ms_by_degree"][str(deg)] = 1e3 * timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar))
    if B <= a.tiled_up_to:
        with _lib.options(tiled_grad=1):
            gt = prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar)
            tt = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar))
        row["tiled_sweep_ms"] = tt * 1e3
        row["speedup_over_tiled"] = tt / tg
        row["max_rel_diff_vs_tiled"] = float((g - gt).abs().max() / gt.abs().max())
    rows.append(row)
    print(json.dumps(row), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.out, "w"), indent=1)
