"""cfg3 (B = 512) against the number of time segments: python tools/sweep_cfg3_seg.py <library>"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from c3_amd import propagation as prop
from c3_amd.workloads import make_workload
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(20): x @ x
w = make_workload(3, B=512)
h0, hks, sig, ph = t(w.h0), t(w.hks), t(w.signals), t(w.fr_phase)
f = lambda: prop.propagate_batch(h0, hks, sig, w.dt, fr_phase=ph)
out = []
for S in (0, 3, 6, 9, 12, 15, 18, 24):
    with _lib.options(segments=S):
        for _ in range(3): f()
        torch.cuda.synchronize(); ts = []
        for _ in range(6):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out.append(f"S={S}: {512 / min(ts):.4e}")
print(os.path.basename(_lib.LIB_PATH), " ".join(out))
