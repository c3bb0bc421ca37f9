"""GPU check + timing of the register-resident kernel (c3p_regd.hip) against the oracle and the arena kernel.

    python tests/checks/check_regd.py [--time]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from c3_amd import propagation, workloads, _lib  # noqa: E402
from oracle import c3_oracle  # noqa: E402


def rand_herm(rng, D, scale):
    a = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    return scale * (a + a.conj().T) / 2


def unitary_case(D, B, N, K, seed, want_dUs=False):
    rng = np.random.default_rng(seed)
    h0 = rand_herm(rng, D, 0.02)
    hks = np.stack([rand_herm(rng, D, 0.01) for _ in range(K)])
    sig = rng.uniform(-1, 1, size=(B, K, N))
    dt = 1.0
    r = propagation.propagate_batch(h0, hks, sig, dt, want_dUs=want_dUs)
    ref = c3_oracle.propagate_batch(h0, hks, sig, dt)
    err = max(np.linalg.norm(r["U"][b] - ref[b]) for b in range(B))
    msg = f"unitary D={D} B={B} N={N} K={K}: kernel={_lib.last_kernel()} err={err:.2e}"
    if want_dUs:
        d = c3_oracle.tf_propagation_vectorized(h0, hks, sig[0], dt)
        e2 = np.abs(r["dUs"][0] - d).max()
        msg += f" dUs err={e2:.2e}"
        err = max(err, e2)
    print(msg, flush=True)
    return err


def lindblad_case(cfg_dims, B, N, seed, want_dUs=False):
    wl = workloads.make_workload(4, B=B, N=N)
    ph = np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase])
    r = propagation.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True,
                                    fr_phase=ph, want_dUs=want_dUs)
    ref = c3_oracle.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, col_ops=wl.col_ops, lindbladian=True, fr_phase=wl.fr_phase)
    err = max(np.linalg.norm(r["U"][b] - ref[b]) for b in range(B))
    print(f"lindblad 81 B={B} N={N}: kernel={_lib.last_kernel()} err={err:.2e}", flush=True)
    return err


def main():
    worst = 0.0
    for (B, N) in [(1, 1), (2, 2), (3, 5), (2, 40), (5, 70), (300, 9)]:
        worst = max(worst, lindblad_case(None, B, N, 1))
    for D in (49, 65, 81):
        for (B, N, K) in [(2, 3, 1), (3, 37, 2), (1, 8, 0)]:
            if K == 0:
                continue
            worst = max(worst, unitary_case(D, B, N, K, 7 + D, want_dUs=(N == 3)))
    print("WORST", worst)
    if "--time" in sys.argv:
        dev = torch.device("cuda:0")
        for B in (256, 512):
            wl = workloads.make_workload(4, B=B, N=1000)
            args = [torch.as_tensor(x, device=dev) for x in (wl.h0, wl.hks, wl.signals)]
            col = torch.as_tensor(wl.col_ops, device=dev)
            ph = torch.as_tensor(np.stack([(p[:, None] - p[None, :]).ravel() for p in wl.fr_phase]), device=dev)
            for env in ("", "1"):
                if env:
                    _lib.set_option("no_regd", "1")
                else:
                    _lib.set_option("no_regd", None)
                for rep in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = propagation.propagate_batch(args[0], args[1], args[2], wl.dt, col_ops=col, lindbladian=True, fr_phase=ph)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    print(f"B={B} N=1000 {'bigd' if env else 'regd'} rep{rep}: {1e3 * (t1 - t0):.1f} ms -> {B / (t1 - t0):.0f} props/s", flush=True)
                if not env:
                    Ureg = r["U"].cpu().numpy()
                else:
                    Ubig = r["U"].cpu().numpy()
            print("regd vs bigd max |dU|_F:", max(np.linalg.norm(Ureg[b] - Ubig[b]) for b in range(B)))
        _lib.set_option("no_regd", None)
    return 0 if worst < 1e-10 else 1


if __name__ == "__main__":
    sys.exit(main())
