import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3_amd import propagation as P, _lib, workloads
from oracle import c3_oracle as o
np.set_printoptions(precision=4, linewidth=200, suppress=True)
rng = np.random.default_rng(0)
for D in (3, 4, 9):
    for N in (1, 2, 3, 9):
        M = rng.normal(size=(2, N, D, D)) + 1j * rng.normal(size=(2, N, D, D))
        out = np.asarray(P.tf_matmul_left(M))
        ref = np.stack([o.tf_matmul_left(M[b]) for b in range(2)])
        err = np.abs(out - ref).max() / np.abs(ref).max()
        print(f"chain D={D} N={N} kernel={_lib.last_kernel()} relerr={err:.2e}")
        if err > 1e-12 and N <= 2 and D <= 4:
            print(out[0]); print(ref[0])
for cfg, N in ((1, 1), (1, 2), (2, 1), (2, 3)):
    wl = workloads.make_workload(cfg, B=2, N=N)
    r = P.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt, want_dUs=True)
    ref = o.propagate_batch(wl.h0, wl.hks, wl.signals, wl.dt)
    print(f"pwc cfg={cfg} N={N} kernel={_lib.last_kernel()} err={np.abs(np.asarray(r['U'])-ref).max():.2e}")
    if cfg == 1 and N == 1:
        print(np.asarray(r['U'])[0]); print(ref[0])
