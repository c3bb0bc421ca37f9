"""rk4_unitary at a BASELINE config's operators: propagators per second of c3p_rk4_unitary (final
propagator only) for several batch sizes, on the kernel the library picks and, with C3P_ODE_PROP_ROWS=1, on the lane-row
column kernel; error of two samples against the oracle.

    python tests/perf/bench_rk4_unitary.py --config 3 --batches 64,256,1024 --out gpurun_out/rk4_unitary_cfg3.json
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from c3_amd import _lib
from c3_amd.workloads import make_workload
from oracle import c3_oracle as o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--batches", default="64,256,1024")
    ap.add_argument("--samples", type=int, default=401, help="Hamiltonian samples Ns (RK steps = (Ns - 1) / 2)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = _lib.load()
    rows = []
    for B in (int(x) for x in a.batches.split(",")):
        wl = make_workload(a.config, B=B, N=a.samples)
        D, K, Ns = wl.D, wl.K, a.samples
        h0, hks, sig = (torch.as_tensor(np.ascontiguousarray(x), device="cuda:0") for x in (wl.h0, wl.hks, wl.signals))
        U = torch.zeros((B, D, D), dtype=torch.complex128, device="cuda:0")
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        for env in (None, "C3P_ODE_PROP_ROWS"):
            if env:
                _lib.set_option(env[4:].lower(), "1")
            try:
                fn = lambda: lib.c3p_rk4_unitary(p(h0), p(hks), p(sig), None, 0, wl.dt, B, K, Ns, D, 0, p(U), None, None)
                assert fn() == 0
                torch.cuda.synchronize()
                kern = _lib.last_kernel()
                best = 1e30
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
            finally:
                if env:
                    _lib.set_option(env[4:].lower(), None)
            got = U[:2].cpu().numpy()
            err = 0.0
            for b in range(2):
                Hs = np.asarray(wl.h0)[None] + np.einsum("kn,kij->nij", np.asarray(wl.signals[b]), np.asarray(wl.hks))
                err = max(err, float(np.abs(got[b] - o.rk4_unitary_arrays(Hs, wl.dt, D)["U"]).max()))
            steps = (Ns - 1) // 2
            row = {"B": B, "D": D, "K": K, "Ns": Ns, "rk4_steps": steps, "kernel": kern, "ms": best * 1e3, "propagators_per_s": B / best,
                   "propagator_steps_per_s": B * steps / best, "max_err_vs_oracle": err}
            rows.append(row)
            print(json.dumps(row), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"config": a.config, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
