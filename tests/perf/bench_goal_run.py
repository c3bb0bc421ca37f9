#!/usr/bin/env python3
"""One optimiser evaluation, timed whole: envelope rows -> signals -> propagators -> goal -> d goal / d rows
(c3_amd.optimal_control.goal_run_with_grad; the reference: Optimizer.goal_run_with_grad, optimizers/optimizer.py:206-216, over
OptimalControl.goal_run, optimalcontrol.py:200-228, and the noise-instance loop of optimalcontrol_robust.py:49-70).

Per case: the forward pass alone (propagate_batch on resident signals), the three-call evaluation (forward, cotangent,
vector-Jacobian product: the forward segment products are computed twice), the fused evaluation (c3p_pwc_unitary_goal_vjp:
once; open systems at D = 7, 8, 9: one taped forward pass + the vjp from the tape), each eagerly and replayed from ONE captured hipGraph (torch.cuda.graph around the whole evaluation).

    python tests/perf/bench_goal_run.py --out gpurun_out/goal_run.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(fn, reps, torch):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return 1e3 * float(np.median(ts))


def graph_timed(fn, reps, torch, dev):
    """capture fn once (after an eager call on the capture stream) and time replays; None if the capture is refused"""
    s = torch.cuda.Stream(device=dev)
    try:
        with torch.cuda.stream(s):
            fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = fn()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        return None, str(e)[:200], None
    ms = timed(g.replay, reps, torch)
    return ms, None, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="1:256:200,2:256:1000,2:1024:1000,3:256:2000,5:64:5000,L3:64:1000,L4:64:1000,L4:256:1000,L9:16:1000,L9:64:1000")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    from c3_amd import _lib, optimal_control as oc, propagation, signals as sg, workloads

    _lib.require_gpu()
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(a, device=dev)
    TWO_PI = 2 * np.pi
    rows = []
    for case in args.cases.split(","):
        tag, B, N = case.split(":")
        B, N = int(B), int(N)
        lind = tag.startswith("L")
        if lind:
            D1 = int(tag[1:])  # L3: one qutrit (cfg1's operators); L4: two coupled qubits; L9: cfg4's two qutrits
            wl = workloads.make_workload(4 if D1 == 9 else 1, B=1, N=8)
            if D1 == 9:
                col = wl.col_ops
            elif D1 == 4:
                sz, sx, sm, id2 = np.diag([0.0, 1.0]), np.array([[0, 1], [1, 0]], dtype=float), np.array([[0, 1], [0, 0]], dtype=float), np.eye(2)
                w1, w2, g = 5.05e9 * TWO_PI, 5.65e9 * TWO_PI, 20e6 * TWO_PI

                class _W:
                    pass

                wl = _W()
                wl.D, wl.K = 4, 2
                wl.h0 = (w1 * np.kron(sz, id2) + w2 * np.kron(id2, sz) + g * np.kron(sx, sx)).astype(complex)
                wl.hks = np.stack([np.kron(sx, id2), np.kron(id2, sx)]).astype(complex)
                col = np.stack([np.sqrt(1 / 27e-6) * np.kron(sm, id2), np.sqrt(1 / 23e-6) * np.kron(id2, sm)]).astype(complex)
            else:
                a = workloads.annihilator(3).astype(complex)
                col = workloads.qubit_collapse_op(a, 27e-6, 39e-6)[None]
            dims = {3: [3], 4: [2, 2], 9: [3, 3]}[D1]
        else:
            wl = workloads.make_workload(int(tag), B=1, N=8)
            col = None
            dims = {1: [3], 2: [3, 3], 3: [3, 3, 3], 5: [3, 3, 4]}[int(tag)]
        D, K = wl.D, wl.K
        sim_res, awg_res = 100e9, 2e9
        T = N / sim_res
        rng = np.random.default_rng(3)
        chans = [[dict(shape="gaussian_nonorm", amp=rng.uniform(0.2, 0.5, size=B), xy_angle=0.1 * k, freq_offset=-50e6 * TWO_PI, delta=-0.5, t_final=T, sigma=T / 4, drag=True)] for k in range(K)]
        env, shapes = sg.pack_components(chans, B=B)
        f0 = [5.05e9, 5.65e9, 6.25e9]
        carrier = np.tile(np.array([[f0[k] * TWO_PI, 1e9 * TWO_PI] for k in range(K)]), (B, 1, 1))
        index = list(range(len(dims)))
        L = 2 ** len(index)
        ideal = np.eye(L, dtype=complex)
        Dm = D * D if lind else D
        ph = rng.uniform(0, 6, size=(B, Dm))
        env_d, car_d, shp_d, h0_d, hk_d, id_d, ph_d = t(env), t(carrier), t(shapes.astype(np.int32)), t(wl.h0), t(wl.hks), t(ideal), t(ph)
        col_d = None if col is None else t(col)
        fid = "lindbladian_unitary_infid" if lind else "unitary_infid"

        def run(fused):
            return oc.goal_run_with_grad(h0_d, hk_d, env_d, shp_d, car_d, 0.0, T, awg_res, sim_res, id_d, index, dims, fr_phase=ph_d, fid_func=fid,
                                         col_ops=col_d, device=dev, fused=fused)

        sig = sg.synthesize_signals(env_d, shp_d, car_d, 0.0, T, awg_res, sim_res)
        dt = 1.0 / sim_res
        if lind:
            fwd = lambda: propagation.propagate_batch(h0_d, hk_d, sig, dt, col_ops=col_d, lindbladian=True, fr_phase=ph_d)
        else:
            fwd = lambda: propagation.propagate_batch(h0_d, hk_d, sig, dt, fr_phase=ph_d)
        reps = max(2, min(50, int(2e5 / (B * N * (Dm / 9.0) ** 3)) + 2))
        row = {"case": f"{'Lindblad ' if lind else ''}D={D}" + (f" ({Dm}x{Dm})" if lind else ""), "B": B, "N": N, "K": K, "reps": reps}
        row["forward_ms"] = timed(fwd, reps, torch)
        # the forward pass alone, replayed from a captured hipGraph: what launch gaps cost the 12 - 130 us kernels
        row["forward_graph_ms"], ferr, _ = graph_timed(fwd, reps, torch, dev)
        if ferr:
            row["forward_graph_error"] = ferr
        a = run(False)
        row["three_call_ms"] = timed(lambda: run(False), reps, torch)
        gms, err, _ = graph_timed(lambda: run(False), reps, torch, dev)
        row["three_call_graph_ms"] = gms
        if err:
            row["three_call_graph_error"] = err
        if lind and propagation.lindblad_tape_supported(B, K, N, D):
            # open systems at D = 7, 8, 9: one forward pass + a tape (run(None)) against forward + vjp recomputing the chain
            b = run(None)
            for key in ("goal", "grad_env", "grad_carrier"):
                x, y = a[key], b[key]
                assert float((x - y).abs().max()) <= 1e-9 * max(float(y.abs().max()), 1e-30), (case, key)
            row["fused_ms"] = timed(lambda: run(None), reps, torch)
        if not lind and propagation.goal_vjp_is_fused(B, D):
            b = run(True)
            for key in ("goal", "grad_env", "grad_carrier"):
                x, y = a[key], b[key]
                assert float((x - y).abs().max()) <= 1e-10 * max(float(y.abs().max()), 1e-30), (case, key)
            row["fused_ms"] = timed(lambda: run(True), reps, torch)
            gms, err, _ = graph_timed(lambda: run(True), reps, torch, dev)
            row["fused_graph_ms"] = gms
            if err:
                row["fused_graph_error"] = err
        best = min(v for k, v in row.items() if k.endswith("_ms") and not k.startswith("forward") and v is not None)
        row["best_ms"] = best
        row["iterations_per_s"] = 1e3 / best
        row["gradients_per_s"] = 1e3 * B / best
        row["best_over_forward"] = best / row["forward_ms"]
        row["three_call_over_forward"] = row["three_call_ms"] / row["forward_ms"]
        rows.append(row)
        print(json.dumps(row), flush=True)
    out = {"what": "one optimiser evaluation (signal synthesis + propagation + goal + gradient w.r.t. the envelope rows) per batch of B "
                   "parameter / noise instances; ms per evaluation, median of 3 x reps; forward_ms = propagate_batch alone on resident signals",
           "rows": rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
