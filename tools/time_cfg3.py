"""cfg3 forward (and gradient) timing of one library: python tools/time_cfg3.py <library> [grad]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
from oracle import c3_oracle
t = lambda x: torch.as_tensor(x, device="cuda:0")
# clock ramp
x = torch.randn(4096, 4096, device="cuda:0"); 
for _ in range(20): x @ x
torch.cuda.synchronize()
for B in (256, 512):
    w = make_workload(3, B=B)
    h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
    f = lambda: prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
    for _ in range(5): r = f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    U = r["U"][:2].cpu().numpy()
    ref = c3_oracle.propagate_batch(w.h0, w.hks, w.signals[:2, :, :], w.dt, fr_phase=w.fr_phase[:2]) if B == 256 else None
    err = max(np.linalg.norm(U[b] - ref[b]) for b in range(2)) if ref is not None else -1
    print(os.path.basename(_lib.LIB_PATH), f"cfg3 B={B} ms {1e3 * min(ts):.3f} propagators/s {B / min(ts):.4e} err {err:.2e}", flush=True)
if len(sys.argv) > 2:
    w = make_workload(3, B=256)
    h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
    Ubar = torch.randn(256, w.D, w.D, dtype=torch.complex128, device="cuda:0")
    f = lambda: prop.propagate_batch_vjp(h0, hks, sig, w.dt, Ubar, fr_phase=ph)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(os.path.basename(_lib.LIB_PATH), "cfg3 gradient B=256 ms %.3f" % (1e3 * min(ts)), flush=True)
