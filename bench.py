#!/usr/bin/env python
"""Headline benchmark: full-gate propagators/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic pulse-parameter
samples: U[b] = FR_b * prod_n exp(-i H_b[n] dt) for all b (config cfg2 of BASELINE.json:
two-qubit CR gate, D=9, 1000 PWC slices, B=256 samples per GPU; weak scaling: every rank
propagates its own 256 samples, then one RCCL all-gather of the U slabs).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6  # MI355X dense fp64 (vector = matrix; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


# thresholds of the reference's expm (tf.linalg.expm, Higham 2005 Pade 3/5/7/9/13 chosen from ||A||_1)
_PADE_THETA = (1.495585217958292e-2, 2.539398330063230e-1, 9.504178996162932e-1, 2.097847961257068, 5.371920351148152)
_PADE_PRODUCTS = (2, 3, 4, 5, 6)


def algorithmic_flops_per_prop(wl):
    """sum over slices of F_slice(D,K,m,s) = 8 D^3 (pi_m + s + 1) + (32/3) D^3 + 4 K D^2 with the
    Pade order m / squarings s the REFERENCE's expm would pick per slice from ||A_n||_1; one chain product dropped
    per propagator (+ 4 D^3 per slice of superoperator fill for Lindblad).  Self-contained: the oracle is not
    consulted for the reported roofline."""
    import numpy as np

    H = wl.h0[None] + np.einsum("kn,kij->nij", wl.signals[0], wl.hks)
    if wl.lindblad:
        D = wl.D
        I = np.eye(D)
        clp = np.zeros((D * D, D * D), dtype=np.complex128)
        for c in wl.col_ops:
            spre, spost = np.kron(c, I), np.kron(I, c.T)
            clp += spre @ spost.conj().T - 0.5 * (spre.conj().T @ spre) - 0.5 * (spost @ spost.conj().T)
        norms = np.empty(H.shape[0])
        for n in range(H.shape[0]):  # ||dt L(H_n)||_1 (propagation.py:551-582)
            L = -1j * (np.kron(H[n], I) - np.kron(I, H[n].T)) + clp
            norms[n] = np.abs(L * wl.dt).sum(axis=0).max()
        Dm = D * D
    else:
        norms = np.abs(-1j * H * wl.dt).sum(axis=-2).max(axis=-1)
        Dm = wl.D
    tot = 0.0
    for a in norms:
        idx = int(np.searchsorted(_PADE_THETA, a))
        sq = 0
        if idx >= len(_PADE_THETA):
            idx = len(_PADE_THETA) - 1
            sq = max(0, int(np.floor(np.log2(a / _PADE_THETA[-1]))))  # TF's squaring count
        tot += 8.0 * Dm**3 * (_PADE_PRODUCTS[idx] + sq + 1) + (32.0 / 3.0) * Dm**3 + 4.0 * wl.K * Dm**2
    tot -= 8.0 * Dm**3
    if wl.lindblad:
        tot += 4.0 * wl.D**3 * wl.N
    return float(tot)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config index (2 = headline)")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: config's B, capped at 256/GPU for cfg3/5)")
    ap.add_argument("--slices", type=int, default=None)
    ap.add_argument("--gather-every", type=int, default=32, help="batches exchanged per all-gather (multi-GPU)")
    ap.add_argument("--ramp-ms", type=float, default=60.0, help="untimed device clock ramp before the W warmup steps (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic LDS kernel")
    ap.add_argument("--check", action="store_true", help="verify a few samples against the oracle")
    args = ap.parse_args()

    import numpy as np
    import torch

    from c3_amd import _lib, propagation, workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = workloads.CONFIGS[args.config]
    B = args.batch if args.batch is not None else min(cfg["B"], 256)
    wl = workloads.make_workload(args.config, B=B, N=args.slices, b_offset=rank * B)
    Dm = wl.D * wl.D if wl.lindblad else wl.D
    fr = wl.fr_phase
    if wl.lindblad:
        fr = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
    bp = propagation.BatchPropagator(
        torch.as_tensor(wl.h0, device=dev),
        torch.as_tensor(wl.hks, device=dev),
        torch.as_tensor(wl.signals, device=dev),
        wl.dt,
        col_ops=torch.as_tensor(wl.col_ops, device=dev) if wl.lindblad else None,
        fr_phase=torch.as_tensor(fr, device=dev),
        force_generic=args.generic,
    )
    # The only data-path collective is the all-gather of the U slabs (RCCL over xGMI), in stream order.
    # xGMI all-gathers of 0.33 MB per rank are latency-bound (tens of microseconds against a 0.25 ms
    # batch), so the slabs of `--gather-every` consecutive batches (default 32) are exchanged by ONE
    # collective: fewer, larger messages; every batch's propagators still reach every rank inside the
    # timed region.  (An asynchronous gather beside the chain kernel was measured SLOWER: the RCCL kernel
    # takes CUs away from a grid sized to fill the chip exactly and creates a partial second round.)
    G = max(1, int(args.gather_every))
    Ubuf = torch.empty((G, B, Dm, Dm), dtype=torch.complex128, device=dev)
    gathered = torch.empty((world * G * B * Dm * Dm,), dtype=torch.complex128, device=dev) if use_dist else None
    counter = [0]
    pending = [0]

    def flush():
        g = pending[0]
        if use_dist and g > 0:
            n = g * B * Dm * Dm
            dist.all_gather_into_tensor(torch.view_as_real(gathered[: world * n]), torch.view_as_real(Ubuf[:g].reshape(-1)))
        pending[0] = 0

    def step():
        i = counter[0] % G
        counter[0] += 1
        U = bp.run(out=Ubuf[i])
        pending[0] += 1
        if pending[0] == G:
            flush()
        return U

    def drain():
        flush()
        counter[0] = 0

    lib = _lib.load()
    if use_dist:
        # the communicator and every message size of the run are set up before anything is timed
        for g in sorted({min(G, max(1, args.steps)), args.steps % G, args.warmup % G, min(G, max(1, args.warmup))} - {0}):
            pending[0] = g
            flush()
        torch.cuda.synchronize()
    # Untimed clock ramp: the MI355X needs tens of milliseconds of sustained work to reach its steady clocks
    # (measured: 0.210 ms per batch after 5 warmup batches, 0.194 ms after 300).  The metric is sustained
    # throughput, so the device is brought to steady state before the W warmup steps and the K timed steps.
    if args.ramp_ms > 0:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for _ in range(16):
                bp.run(out=Ubuf[0])
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    # The K steps end here on this rank (the last all-gather inside drain() has already waited for every rank's
    # slabs); the closing barrier follows and the MAX over ranks of the per-rank times is reported, so the
    # barrier's own latency (~0.3 ms, 6 % of a 30-step run) is not booked as step time.
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_name = _lib.last_kernel()

    # ---- dominant-kernel duration: HIP events on the launch stream, outside the timed region ----
    lib.c3p_set_profiling(1)
    kms = []
    for _ in range(min(10, max(3, args.steps))):
        bp.run()
        torch.cuda.synchronize()
        kms.append(lib.c3p_last_kernel_ms())
    lib.c3p_set_profiling(0)
    kernel_ms = float(np.mean(kms))

    err = None
    if args.check and rank == 0:
        from oracle import c3_oracle

        nchk = min(4, B)
        U = bp.run()
        torch.cuda.synchronize()
        ref = c3_oracle.propagate_batch(
            wl.h0, wl.hks, wl.signals[:nchk], wl.dt, col_ops=wl.col_ops, lindbladian=wl.lindblad, fr_phase=wl.fr_phase[:nchk]
        )
        Uh = U[:nchk].cpu().numpy()
        err = float(max(np.linalg.norm(Uh[b] - ref[b]) for b in range(nchk)))

    if rank == 0:
        total_props = world * B * args.steps
        value = total_props / elapsed
        f_prop = algorithmic_flops_per_prop(wl)
        achieved = f_prop * B / (kernel_ms * 1e-3) / 1e12
        traffic = None
        issued = None
        tfile = os.path.join(ROOT, "profiles", "r01", "traffic.json")
        if os.path.exists(tfile) and args.batch is None and args.slices is None and not args.generic:
            try:
                prof = json.load(open(tfile)).get(f"cfg{args.config}", {})
                traffic = prof.get("bytes_per_launch")
                issued = prof.get("issued_mfma_flop_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "full-gate propagators/s",
            "value": value,
            "unit": "propagators/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c128",
            "data": "synthetic",
            "config": {
                "workload": wl.name,
                "D": wl.D,
                "matrix_dim": Dm,
                "slices": wl.N,
                "controls": wl.K,
                "batch_per_gpu": B,
                "global_batch": world * B,
                "clock_ramp_ms": args.ramp_ms,
                "parallelism": f"dp{world} (batch sharded; one RCCL all-gather of U per {G} batches)" if world > 1 else "single GPU",
                "kernel": kernel_name,
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_TFLOPS,
                "traffic": traffic,
                "issued_flop_per_launch": issued,
                "algorithmic_flop_per_launch": f_prop * B,
                "kernel": f"chain kernel ({kernel_name})",
                "kernel_ms": kernel_ms,
                "algorithmic_flop_per_propagator": f_prop,
                "note": "fp64 compute-bound path: algorithmic flops (SURVEY 8d, reference Pade order per slice) / hipEvent kernel time; peak = dense fp64 (MFMA = vector on MI355X); traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01/traffic.json), null when not profiled for this configuration",
            },
        }
        if err is not None:
            out["max_fro_err_vs_oracle"] = err
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c3_oracle  # the CPU restatement, timed beside the GPU path (checker only)

            out["cpu_baseline"] = cpu_baseline(wl, c3_oracle)
        print(json.dumps(out), flush=True)
    if use_dist:
        if args.check and gathered is not None:
            # the gathered slab of this rank must equal its own result
            last = (counter[0] - 1) % nbuf
            assert torch.equal(gathered[last][rank * B : (rank + 1) * B], Ubuf[last]), "all-gather mismatch"
        dist.destroy_process_group()


def cpu_baseline(wl, oracle):
    """The numpy oracle (reference algorithm: batched per-slice Pade expm + pairwise tree
    product) timed single-threaded on a bounded sample of the same workload."""
    import numpy as np

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None

    def run(nb):
        t0 = time.perf_counter()
        oracle.propagate_batch(
            wl.h0, wl.hks, wl.signals[:nb], wl.dt, col_ops=wl.col_ops, lindbladian=wl.lindblad, fr_phase=wl.fr_phase[:nb]
        )
        return time.perf_counter() - t0

    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        if ctx is not None:
            ctx.__enter__()
        t1 = run(1)
        nb = int(max(1, min(wl.B, round(12.0 / max(t1, 1e-3)))))
        t = run(nb)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return {
        "value": nb / t,
        "unit": "propagators/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{nb} of {wl.B} samples of {wl.name}, numpy oracle (Higham Pade expm per slice + tf_matmul_n tree), 1 thread, {os.cpu_count()} host cores present",
    }


if __name__ == "__main__":
    main()
