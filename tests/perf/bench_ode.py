"""ODE-solver variant: RK steps/s of c3p_ode_solve at a BASELINE config's operators, for
several batch sizes, with the final-state error against the oracle's solver (same tableau) and, for the Schroedinger
step, against the PWC propagator.

    python tests/perf/bench_ode.py --config 2 --batches 256,2048,16384,131072 --out gpurun_out/ode_cfg2.json

Flop accounting per RK step and sample (complex MAC = 8 flop, real x complex = 4):
  schrodinger: stages * 8 D^2 (matrix-vector) + nodes * 4 K D^2 (H assembly)      -- SURVEY 8d's figure, rk4: 4 / 3
  von_neumann: stages * 2 * 8 D^3 + nodes * 4 K D^2;  lindblad adds per collapse operator 2 * 8 D^3 (C rho C^+) and
               2 * 8 D^3 for the anticommutator (the reference forms C^+ C rho and rho C^+ C per stage)
`frac` = algorithmic flops / time / 78.6 TFLOP/s (dense fp64 vector peak).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from c3_amd import _lib, propagation as prop

if os.environ.get("C3P_LIB"):  # A/B builds of the library (e.g. the -DC3P_RHOQ_TIMING probes)
    _lib.LIB_PATH = os.path.abspath(os.environ["C3P_LIB"])
from c3_amd.workloads import make_workload
from oracle import c3_oracle as o

PEAK = 78.6e12
STAGES = {"rk4": (4, 3), "rk38": (4, 4), "rk5": (7, 6), "tsit5": (7, 6)}  # (stages, distinct nodes)


def flops_per_step(D, K, C, solver, step):
    st, nodes = STAGES[solver]
    asm = nodes * 4 * K * D * D
    if step == "schrodinger":
        return st * 8 * D * D + asm
    f = st * 2 * 8 * D**3 + asm
    if step == "lindblad":
        f += st * C * (2 * 8 * D**3 + 2 * 8 * D**3 + 8 * D**3)  # C rho C^+, {C^+ C, rho}, C^+ C itself
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batches", default="256,2048,16384,131072")
    ap.add_argument("--solvers", default="rk4,rk38,rk5,tsit5")
    ap.add_argument("--steps", default="schrodinger,von_neumann,lindblad")
    ap.add_argument("--rho-batches", default="256,2048,16384")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--complex-ops", action="store_true", help="add an imaginary part to the control operators (complex instance)")
    ap.add_argument("--synth-col", action="store_true", help="synthetic collapse operators where the config has none (lindblad step)")
    ap.add_argument("--trajectory", action="store_true", help="ode_solver (every state of the trajectory) instead of ode_solver_final_state")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = "cuda:0"
    rows = []
    wl0 = make_workload(a.config, B=4)
    D, K, N = wl0.D, wl0.K, wl0.N
    hks_np = np.array(wl0.hks)
    if a.complex_ops:
        # a Y-type drive: i (a - a^+)-like Hermitian imaginary part on the first control operator
        im = np.triu(np.abs(hks_np[0].real), 1)
        hks_np[0] = hks_np[0] + 1j * (im - im.T)
    col_np = getattr(wl0, "col_ops", None)
    if col_np is None:
        wl4 = make_workload(4, B=1, N=4)
        col_np = wl4.col_ops if wl4.D == D else None
    if col_np is None and a.synth_col:
        # two real band operators (a relaxation-like superdiagonal, a dephasing-like diagonal) for configs without their own
        lower = np.diag(np.sqrt(np.arange(1, D) % 3 + 1.0), 1)
        col_np = np.stack([0.05 * lower, 0.03 * np.diag(np.arange(D) % 3).astype(float)]).astype(complex)
    psi0 = np.zeros((D, 1), complex)
    psi0[0, 0] = 1.0
    rho0 = psi0 @ psi0.conj().T
    for step in a.steps.split(","):
        if step == "lindblad" and col_np is None:
            continue
        batches = a.batches if step == "schrodinger" else a.rho_batches
        for B in (int(x) for x in batches.split(",")):
            wl = make_workload(a.config, B=B)
            h0, sig = (torch.as_tensor(x, device=dev) for x in (wl.h0, wl.signals))
            hks = torch.as_tensor(hks_np, device=dev)
            init = torch.as_tensor(psi0 if step == "schrodinger" else rho0, device=dev)
            col = torch.as_tensor(col_np, device=dev) if step == "lindblad" else None
            C = int(col_np.shape[0]) if step == "lindblad" else 0
            for solver in a.solvers.split(","):
                fn = lambda: prop.ode_solve_batch(h0, hks, sig, wl.dt, init, solver, step, col_ops=col, final_only=not a.trajectory)
                res = fn()
                torch.cuda.synchronize()
                kern = _lib.last_kernel()
                best = 1e30
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    res = fn()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                got = (res[:2, -1] if a.trajectory else res[:2]).cpu().numpy()
                ref = np.stack([
                    o.ode_solver_arrays(wl.h0, hks_np, wl.signals[b], wl.ts, psi0 if step == "schrodinger" else rho0, solver, step,
                                        col=col_np if step == "lindblad" else None, final_only=True)["states"] for b in range(2)])
                fl = flops_per_step(D, K, C, solver, step)
                steps = B * N
                row = {"step": step, "solver": solver, "trajectory": bool(a.trajectory), "B": B, "N": N, "D": D, "K": K, "C": C, "kernel": kern, "complex_ops": bool(a.complex_ops),
                       "ms": best * 1e3, "rk_steps_per_s": steps / best, "final_states_per_s": B / best,
                       "algorithmic_flop_per_step": fl, "algorithmic_tflops": steps * fl / best * 1e-12, "frac": steps * fl / best / PEAK,
                       "err_vs_oracle_rk": float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())), "max_abs_state": float(np.abs(ref).max())}
                if step == "schrodinger" and not a.complex_ops:
                    U = o.propagate_batch(wl.h0, wl.hks, wl.signals[:2], wl.dt)
                    row["err_vs_pwc"] = float(np.abs(got - np.stack([U[b] @ psi0 for b in range(2)])).max())
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump({"config": wl0.name, "peak_flops": PEAK, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
