import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from c3_amd import _lib, propagation as prop
from c3_amd.workloads import make_workload
t = lambda x: torch.as_tensor(x, device="cuda:0")
x = torch.randn(4096, 4096, device="cuda:0")
for _ in range(30): x @ x
torch.cuda.synchronize()
w = make_workload(2, B=256)
k = min(1, w.K - 1)
up = np.triu(w.hks[k].real, 1)
hks = w.hks.copy(); hks[k] = hks[k] + 0.3j * (up - up.T)
a = (t(w.h0), t(hks), t(w.signals), w.dt)
ph = t(w.fr_phase)
def timed(f, reps=50, rounds=5):
    for _ in range(10): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / reps)
    return min(ts) * 1e3
for seg in (None, 16, 32, 64):
    for skew in (500, 560, 600, 640, 680, 700, 740):
        _lib.set_option("mw_skew", skew); _lib.set_option("segments", seg)
        try:
            print("segments", seg, "skew", skew, "%.4f ms" % timed(lambda: prop.propagate_batch(*a, fr_phase=ph)), flush=True)
        except Exception as e:
            print("segments", seg, "skew", skew, "failed", e)
