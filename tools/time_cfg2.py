"""cfg2 (B = 256) real and complex-operator instances with one library: python tools/time_cfg2.py <library>"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
from oracle import c3_oracle
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()
out = []
for cplx in (False, True):
    w = make_workload(2, B=256)
    if cplx:  # as bench.py --complex: the second control operator gets an imaginary (Hermitian) part
        k = min(1, w.K - 1)
        up = np.triu(w.hks[k].real, 1)
        w.hks = w.hks.copy()
        w.hks[k] = w.hks[k] + 0.3j * (up - up.T)
    h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
    f = lambda: prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
    for _ in range(20): r = f()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 50)
    U = r["U"][:1].cpu().numpy()
    ref = c3_oracle.propagate_batch(w.h0, w.hks, w.signals[:1], w.dt, fr_phase=w.fr_phase[:1])
    out.append(f"{'complex' if cplx else 'real'}: {1e3 * min(ts):.4f} ms {256 / min(ts):.4e}/s err {np.linalg.norm(U[0] - ref[0]):.1e}")
print(os.path.basename(_lib.LIB_PATH), " | ".join(out), flush=True)
