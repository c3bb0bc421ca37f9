"""Timing of the tiled backward sweep (round 3): the Lindblad gradient (c3p_pwc_lindblad_vjp) at cfg4's operators and the
unitary gradient above D = 40, next to the forward pass of the same shape.
    python tools/bench_grad_tiled.py --out gpurun_out/r03/grad_tiled.json"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from c3_amd import propagation as prop
from c3_amd import _lib
from c3_amd.workloads import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = "cuda:0"
t = lambda x: torch.as_tensor(x, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps


rows = []
# cfg4: 81 x 81 Lindblad superoperators, N = 1000
for B in (16, 64):
    w = make_workload(4, B=B)
    Dm = w.D * w.D
    h0, hks, sig, col = t(w.h0), t(w.hks), t(w.signals), t(w.col_ops)
    Ubar = torch.randn(B, Dm, Dm, dtype=torch.complex128, device=dev)
    tf = timed(lambda: prop.propagate_batch(h0, hks, sig, w.dt, col_ops=col, lindbladian=True))
    tg = timed(lambda: prop.propagate_batch_lindblad_vjp(h0, hks, sig, w.dt, col, Ubar))
    # executed by the sweep per slice (s squarings): forward 6 + s + 1 products, backward 15 + 3 s + 2
    s = 1
    flop = B * w.N * (7 + s + 17 + 3 * s) * 8 * Dm**3
    rows.append({"case": "cfg4 Lindblad 81x81", "B": B, "N": w.N, "Dm": Dm, "forward_ms": tf * 1e3, "vjp_ms": tg * 1e3, "vjp_over_forward": tg / tf,
                 "gradients_per_s": B / tg, "vjp_executed_TFLOPs": flop / tg / 1e12})
    print(json.dumps(rows[-1]), flush=True)
# unitary, D = 48 and 64 (complex Hermitian random operators), N = 1000
for D, B in ((48, 64), (64, 64)):
    rng = np.random.default_rng(D)
    herm = lambda sc: (lambda m: sc * (m + m.conj().T) / 2)(rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D)))
    N, K = 1000, 2
    h0, hks = t(herm(0.12)), t(np.stack([herm(0.08) for _ in range(K)]))
    sig = t(rng.uniform(-1, 1, size=(B, K, N)))
    Ubar = torch.randn(B, D, D, dtype=torch.complex128, device=dev)
    tf = timed(lambda: prop.propagate_batch(h0, hks, sig, 1.0))
    tg = timed(lambda: prop.propagate_batch_vjp(h0, hks, sig, 1.0, Ubar))
    rows.append({"case": f"unitary D={D}", "B": B, "N": N, "Dm": D, "forward_ms": tf * 1e3, "vjp_ms": tg * 1e3, "vjp_over_forward": tg / tf,
                 "gradients_per_s": B / tg})
    if D <= 64:
        tv = timed(lambda: prop.propagate_batch_vjp(h0, hks, sig, 1.0, Ubar, force_generic=True))
        rows[-1]["valu_sweep_ms (round 2 path)"] = tv * 1e3
    print(json.dumps(rows[-1]), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"rows": rows}, open(a.out, "w"), indent=1)
